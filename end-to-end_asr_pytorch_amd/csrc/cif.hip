// CIF - continuous integrate-and-fire (cif_model.py:57-106) for gfx950.
//
// The firing decision `integrate > threshold` is a discrete function of an fp32 running sum that is decremented by
// exactly 1.0 on fire; reproducing the reference's boundaries requires the reference's operation ORDER, so the
// recurrence itself is not re-associated.  One wavefront per utterance walks the T frames: alpha is loaded 64
// frames at a time (coalesced), each frame's value is broadcast with v_readlane and every lane runs the same scalar
// recurrence (a few dependent VALU ops per frame); the lane that owns the frame records its weights.  The
// H-wide weighted sums - the actual bandwidth - run in a second, fully parallel kernel: one workgroup per output
// token, threads over channels, frames of the token's segment accumulated in the reference's order with separate
// (non-fused) multiply and add so the fp32 results match the CPU loop bit-for-bit.
#include "asr_common.h"

// The reference rounds the product and the sum separately (torch: cur[:,None]*hidden, then +=); hipcc contracts
// a*b+c into v_fma_f32 by default, which changes the last bit.  Contraction is disabled for this whole file (pragma
// here + -ffp-contract=off in the build; the __fmul_rn/__fadd_rn header inlines keep their own contract flag, so the
// arithmetic below is written with plain operators).
#pragma clang fp contract(off)

namespace {

__global__ __launch_bounds__(64) void cif_scan_kernel(const float* __restrict__ alpha, int L, float thr, float* __restrict__ cur_out,
                                                      float* __restrict__ rem_out, int32_t* __restrict__ fire_idx,
                                                      int32_t* __restrict__ n_fire, int32_t* __restrict__ n_label) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* a = alpha + (int64_t)b * L;
    float integrate = 0.f;
    int n = 0;
    float psum = 0.f;
    for (int t0 = 0; t0 < L; t0 += 64) {
        const int t = t0 + lane;
        const float av = (t < L) ? a[t] : 0.f;
        psum += av;
        float my_cur = 0.f, my_rem = 0.f;
        const int cnt = min(64, L - t0);
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            if (i < cnt) {  // wave-uniform
                const float al = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, av), i));
                const float dc = 1.0f - integrate;
                integrate = integrate + al;
                const bool fire = integrate > thr;
                if (fire) {
                    integrate = integrate - 1.0f;
                    if (lane == i) fire_idx[(int64_t)b * L + n] = t0 + i;
                    ++n;
                }
                const float c = fire ? dc : al;
                if (lane == i) { my_cur = c; my_rem = al - c; }
            }
        }
        if (t < L) {
            cur_out[(int64_t)b * L + t] = my_cur;
            rem_out[(int64_t)b * L + t] = my_rem;
        }
    }
    psum = wave_sum(psum);
    if (lane == 0) {
        n_fire[b] = n;
        n_label[b] = (int32_t)rintf(psum);  // torch.round: half to even
    }
}

__global__ __launch_bounds__(256) void cif_gather_kernel(const float* __restrict__ hidden, const float* __restrict__ cur,
                                                         const float* __restrict__ rem, const int32_t* __restrict__ fire_idx,
                                                         const int32_t* __restrict__ n_fire, int L, int H, int Umax,
                                                         float* __restrict__ out) {
    const int u = blockIdx.x % Umax, b = blockIdx.x / Umax;
    float* o = out + ((int64_t)b * Umax + u) * H;
    if (u >= n_fire[b]) {
        for (int c = threadIdx.x; c < H; c += 256) o[c] = 0.f;
        return;
    }
    const int t_end = fire_idx[(int64_t)b * L + u];
    const int t_start = (u == 0) ? 0 : fire_idx[(int64_t)b * L + u - 1];
    const float* hb = hidden + (int64_t)b * L * H;
    const float* cb = cur + (int64_t)b * L;
    const float w0 = (u == 0) ? cb[0] : rem[(int64_t)b * L + t_start];
    for (int c = threadIdx.x; c < H; c += 256) {
        float frame = w0 * hb[(int64_t)t_start * H + c];
        for (int t = t_start + 1; t <= t_end; ++t) {
            const float prod = cb[t] * hb[(int64_t)t * H + c];  // rounded product, then rounded sum (no FMA)
            frame = frame + prod;
        }
        o[c] = frame;
    }
}

}  // namespace

extern "C" int asr_cif_scan_fwd(void* stream, const float* alpha, int B, int L, float threshold, float* cur, float* rem,
                                int32_t* fire_idx, int32_t* n_fire, int32_t* n_label) {
    ASR_REQUIRE(alpha && cur && rem && fire_idx && n_fire && n_label && B > 0 && L > 0, ASR_ERR_ARG, "cif_scan: bad args");
    hipLaunchKernelGGL(cif_scan_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), alpha, L, threshold, cur, rem, fire_idx,
                       n_fire, n_label);
    ASR_LAUNCH_CHECK("cif_scan_fwd");
    return 0;
}

extern "C" int asr_cif_gather_fwd(void* stream, const float* hidden, const float* cur, const float* rem, const int32_t* fire_idx,
                                  const int32_t* n_fire, int B, int L, int H, int Umax, float* out) {
    ASR_REQUIRE(hidden && cur && rem && fire_idx && n_fire && out && B > 0 && L > 0 && H > 0 && Umax > 0, ASR_ERR_ARG,
                "cif_gather: bad args");
    hipLaunchKernelGGL(cif_gather_kernel, dim3(B * Umax), dim3(256), 0, static_cast<hipStream_t>(stream), hidden, cur, rem, fire_idx,
                       n_fire, L, H, Umax, out);
    ASR_LAUNCH_CHECK("cif_gather_fwd");
    return 0;
}
