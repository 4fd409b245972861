// Hardware probes (test infrastructure for kernel development; not on the product path).
#include "asr_common.h"

typedef __attribute__((ext_vector_type(4))) short s16x4;

namespace {
// LDS tile T[64][64] of 16-bit values row*256+col.  Lane l (group g = l>>4, i = l&15, q = i>>2, p = i&3) supplies the address of
// T[m0 + 8*g + q][n0 + 4*p]; the 4 values each lane receives from ds_read_b64_tr_b16 are written out.
__global__ void probe_tr_kernel(short* out, int m0, int n0) {
    __shared__ __attribute__((aligned(16))) short T[64][64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) T[i >> 6][i & 63] = (short)((i >> 6) * 256 + (i & 63));
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
    const short* addr = &T[m0 + 8 * g + q][n0 + 4 * p];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
}  // namespace

extern "C" int asr_debug_probe_tr(void* stream, void* out, int m0, int n0) {
    hipLaunchKernelGGL(probe_tr_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (short*)out, m0, n0);
    ASR_LAUNCH_CHECK("probe_tr");
    return 0;
}
