// Diagnostic (not part of the product): how fast can EVERY CU stream the same 2 MiB (a layer's feed-forward weights) out of its XCD's
// L2?  The row-block kernels (ffn2.hip, ffn.hip, vocab.hip, dgrad_rows.hip) do exactly that, 64 KiB per chunk, and
// profiles/r6/ffn2_stamps.txt measures 32 B/clk/CU for it whatever the kernel does beside (= the guide's 66-73 GB/s per CU for rows
// shared by every workgroup).  Is that a limit of the path or of the way the stream is issued?
//   depth  : 16-byte loads in flight per lane (x waves x 1 KiB per CU)
//   waves  : wavefronts per CU streaming (4 or 8)
//   order  : 0 every workgroup walks the buffer in the same order (lockstep: at any time the XCD's 32 CUs want the same lines)
//            1 the walk of workgroup i starts (i >> 3) * 1/32 of the way in (each CU of an XCD at a different place)
//            2 every 64-KiB chunk in the common order, its sixteen 4-KiB pieces rotated by (i >> 3) (what a kernel with a barrier per
//              chunk could do without changing its results)
//   form   : 0 global_load_dwordx4 into registers, 1 the same bytes by LDS-DMA (buffer_load ... lds) into a 64-KiB ring
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int DEPTH>
__global__ __launch_bounds__(512, 1) void stream_regs(const u32x4* __restrict__ w, int n16, int passes, int order, unsigned* sink) {
    __shared__ unsigned char hold[96 * 1024];      // one workgroup per CU
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int slot = blockIdx.x >> 3;
    const int pieces = n16 / 64;                   // 1-KiB pieces (one wave instruction each)
    const int per_chunk = 64;                      // 64 pieces = 64 KiB
    u32x4 r[DEPTH];
    u32x4 acc = {0, 0, 0, 0};
    const int total = passes * (pieces / nw);
    auto piece_of = [&](int it) {
        int p = (it % (pieces / nw)) * nw + wave;
        if (order == 1) p = (p + slot * (pieces / 32)) % pieces;
        else if (order == 2) {
            const int c = p / per_chunk, q = p % per_chunk;
            p = c * per_chunk + ((q + 4 * slot) % per_chunk);
        }
        return p;
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = w[(size_t)piece_of(d) * 64 + lane];
    for (int it = 0; it < total; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc ^= r[d];
            const int nx = it + DEPTH + d;
            r[d] = w[(size_t)piece_of(nx < total ? nx : d) * 64 + lane];
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc ^= r[d];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) sink[0] = (unsigned)hold[lane];
}

__device__ __forceinline__ void dma16(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory");
}

// LDS-DMA: each wave keeps DEPTH 1-KiB requests in flight into its own slice of the ring (nothing reads the ring)
template <int DEPTH>
__global__ __launch_bounds__(512, 1) void stream_dma(const u32x4* __restrict__ w, int n16, int passes, int order, unsigned* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char ring[128 * 1024];
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = blockIdx.x >> 3;
    const int pieces = n16 / 64, per_chunk = 64;
    const uint64_t b = (uint64_t)w;
    const u32x4 rs = {(unsigned)b, (unsigned)(b >> 32) & 0xffffu, (unsigned)n16 * 16u, 0x00020000u};
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char*)ring) + wave * (DEPTH * 1024);
    const int total = passes * (pieces / nw);
    auto piece_of = [&](int it) {
        int p = (it % (pieces / nw)) * nw + wave;
        if (order == 1) p = (p + slot * (pieces / 32)) % pieces;
        else if (order == 2) {
            const int c = p / per_chunk, q = p % per_chunk;
            p = c * per_chunk + ((q + 4 * slot) % per_chunk);
        }
        return p;
    };
    for (int it = 0; it < total; it += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            // (the request that last used this slot was issued DEPTH requests ago)
            if (d == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH > 1 ? DEPTH - 1 : 0) : "memory");
            dma16(rs, (unsigned)lane * 16u, (unsigned)piece_of(it + d) * 1024u, lds0 + d * 1024);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ring[threadIdx.x] == 0x5a && ring[threadIdx.x + 4096] == 0x5b) sink[0] = 1;
}

int main() {
    const int bytes = 2 << 20, n16 = bytes / 16, passes = 16;
    u32x4* w; unsigned* sink;
    hipMalloc(&w, bytes); hipMalloc(&sink, 64);
    hipMemset(w, 1, bytes);
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%d CUs, %d MHz\n", cus, prop.clockRate / 1000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int form = 0; form < 2; ++form)
    for (int waves = 4; waves <= 8; waves += 4)
    for (int depth = 4; depth <= 16; depth *= 2)
    for (int order = 0; order < 3; ++order)
    for (int wgs = cus; wgs >= 32; wgs = (wgs == cus ? 32 : 0)) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
#define GO(K, D) hipLaunchKernelGGL(K<D>, dim3(wgs), dim3(waves * 64), 0, 0, w, n16, passes, order, sink)
            if (form == 0) { if (depth == 4) GO(stream_regs, 4); else if (depth == 8) GO(stream_regs, 8); else GO(stream_regs, 16); }
            else           { if (depth == 4) GO(stream_dma, 4);  else if (depth == 8) GO(stream_dma, 8);  else GO(stream_dma, 16); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double per_cu = (double)bytes * passes / (best * 1e-3) / 1e9;
        printf("%s waves %d depth %2d (%3d KiB in flight per CU) order %d wgs %3d: %7.1f us  %6.1f GB/s per CU  %5.1f TB/s chip\n",
               form ? "lds-dma" : "regs   ", waves, depth, waves * depth, order, wgs, best * 1e3, per_cu, per_cu * wgs / 1e3);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("error\n"); return 1; }
    return 0;
}
