// Diagnostic: what buffer_load_dwordx4 ... lds does to the LDS destination when the buffer offset is out of range.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ u32x4 rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    return u32x4{(unsigned)b, (unsigned)(b >> 32) & 0xffffu, bytes, 0x00020000u};
}
__global__ void probe(const unsigned* src, unsigned* out, unsigned bytes) {
    __shared__ unsigned lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const u32x4 rs = rsrc_words(src, bytes);
    const unsigned la = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned*)lds);
    unsigned keep;
    // lanes 0..31 in range, lanes 32..63 beyond `bytes`
    const unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(la) : "memory");
    // whole wave far out of range
    const unsigned la2 = la + 1024;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(voff), "s"(rs), "s"(la2), "s"(0x7ff00000u) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    unsigned *src, *out, h[512];
    hipMalloc(&src, 4096); hipMalloc(&out, 2048);
    unsigned hs[1024];
    for (int i = 0; i < 1024; ++i) hs[i] = 0x1000 + i;
    hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, src, out, 512u);     // 512 bytes in range = lanes 0..31
    hipMemcpy(h, out, 2048, hipMemcpyDeviceToHost);
    printf("partial: lane 31 dwords %x %x | lane 32 dwords %x %x | lane 63 %x\n", h[31 * 4], h[31 * 4 + 3], h[32 * 4], h[32 * 4 + 3], h[63 * 4]);
    printf("far:     lane 0 dwords %x %x | lane 63 %x\n", h[256], h[259], h[256 + 63 * 4]);
    return 0;
}
