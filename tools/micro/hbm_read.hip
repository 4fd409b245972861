// Ceiling probe: how fast can ANY kernel stream 542 MB of f32 through a 256-CU MI355X, back-to-back (Infinity-Cache warm) and
// right after another kernel has written 1 GB (cold)?  Variants: grid-stride 16-byte loads (rows ignored), and one workgroup per
// 16 960-byte row (the CTC pass's access pattern) with 1 or 2 rows in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void stream_kernel(const f32x4* __restrict__ x, size_t n4, float* out) {
    f32x4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f32x4 a = x[i], b = x[i + stride], c = x[i + 2 * stride], d = x[i + 3 * stride];
        acc += (a + b) + (c + d);
    }
    for (; i < n4; i += stride) acc += x[i];
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 123.456f) out[0] = s;
}
template <int DEPTH>
__global__ __launch_bounds__(256, 8) void rows_kernel(const float* __restrict__ x, int rows, int ld4, int nv4, float* out) {
    f32x4 acc = {0, 0, 0, 0};
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    for (int r = blockIdx.x * DEPTH; r < rows; r += gridDim.x * DEPTH) {
        f32x4 v[DEPTH][5];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                int i = threadIdx.x + 256 * j;
                v[d][j] = (i < nv4 && r + d < rows) ? x4[(size_t)(r + d) * ld4 + i] : f32x4{0, 0, 0, 0};
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc += v[d][j];
    }
    float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 123.456f) out[0] = s;
}
__global__ void fill_kernel(f32x4* y, size_t n4) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    for (; i < n4; i += stride) y[i] = f32x4{1, 2, 3, 4};
}
int main() {
    const int rows = 32000, V = 4234, ld = 4240, ld4 = ld / 4, nv4 = (V + 3) / 4;
    const size_t n4 = (size_t)rows * ld4;
    float *x, *out, *scratch;
    const size_t scr4 = (size_t)1 << 26;   // 1 GiB
    CK(hipMalloc(&x, n4 * 16)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&scratch, scr4 * 16));
    fill_kernel<<<4096, 256>>>((f32x4*)x, n4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](int variant, int grid) {
        if (variant == 0) stream_kernel<<<grid, 256>>>((const f32x4*)x, n4, out);
        if (variant == 1) rows_kernel<1><<<grid, 256>>>(x, rows, ld4, nv4, out);
        if (variant == 2) rows_kernel<2><<<grid, 256>>>(x, rows, ld4, nv4, out);
    };
    const char* names[] = {"stream", "rows x1", "rows x2"};
    for (int variant = 0; variant < 3; ++variant)
        for (int grid : {2048, 4096, 32000}) {
            if (variant == 2 && grid == 32000) grid = 16000;
            for (int cold = 0; cold < 2; ++cold) {
                std::vector<float> ts;
                for (int it = 0; it < 25; ++it) {
                    if (cold) fill_kernel<<<4096, 256>>>((f32x4*)scratch, scr4);
                    hipEventRecord(a); run(variant, grid); hipEventRecord(b); hipEventSynchronize(b);
                    float ms; hipEventElapsedTime(&ms, a, b);
                    if (it >= 5) ts.push_back(ms);
                }
                std::sort(ts.begin(), ts.end());
                float med = ts[ts.size() / 2];
                printf("{\"kernel\": \"%s\", \"grid\": %d, \"cold\": %d, \"ms\": %.4f, \"TBps\": %.2f}\n", names[variant], grid, cold, med,
                       (double)rows * V * 4 / med / 1e9);
            }
        }
    // the in-step situation exactly: the SAME 542 MB are written by one kernel and read by the next
    for (int variant = 0; variant < 3; ++variant) {
        std::vector<float> ts;
        for (int it = 0; it < 25; ++it) {
            fill_kernel<<<4096, 256>>>((f32x4*)x, n4);
            hipEventRecord(a); run(variant, variant == 0 ? 32000 : 4096); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (it >= 5) ts.push_back(ms);
        }
        std::sort(ts.begin(), ts.end());
        float med = ts[ts.size() / 2];
        printf("{\"kernel\": \"%s\", \"after_writing_the_same_buffer\": 1, \"ms\": %.4f, \"TBps\": %.2f}\n", names[variant], med, (double)rows * V * 4 / med / 1e9);
    }
    // back-to-back x20 (what a microbenchmark loop sees)
    for (int variant = 0; variant < 3; ++variant) {
        hipEventRecord(a);
        for (int it = 0; it < 20; ++it) run(variant, 2048);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("{\"kernel\": \"%s\", \"grid\": 2048, \"back_to_back\": 20, \"ms\": %.4f, \"TBps\": %.2f}\n", names[variant], ms / 20, (double)rows * V * 4 / (ms / 20) / 1e9);
    }
    return 0;
}
