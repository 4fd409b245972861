// Diagnostic (not part of the product): what the shader clock really is under load.  Each wave runs a loop of MFMAs (mode 0), of
// dependent VALU adds (mode 1) or of exp2 (mode 2) and brackets it with s_memtime (shader clock) and s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void probe(int mode, int iters, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) unsigned char lds_probe[16384];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(lds_probe)[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc0 = {}, acc1 = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f); b[i] = (__bf16)(i * 0.01f); }
    float x = threadIdx.x * 1e-3f, y = 0.f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (mode == 0) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
        }
    } else if (mode == 1) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
        }
    } else if (mode == 2) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_exp_f32 %0, %1" : "=v"(y) : "v"(x));
        }
    } else if (mode == 4) {
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
        }
    } else if (mode == 5) {
        f32x16 acc2 = {}, acc3 = {};
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc2, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc3, 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) acc0[i] += acc2[i] + acc3[i];
    } else if (mode == 6) {
        typedef __attribute__((ext_vector_type(4))) float f32x4;
        f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
        for (int i = 0; i < iters; ++i) {
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) acc0[i] += c0[i] + c1[i] + c2[i] + c3[i];
    } else if (mode == 7) {      // mfma + 6 v_exp (independent destinations)
        float e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %6\n\tv_exp_f32 %1, %6\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %6\n\tv_exp_f32 %4, %6\n\tv_exp_f32 %5, %6" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3), "=v"(e4), "=v"(e5) : "v"(x));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            asm volatile("v_exp_f32 %0, %6\n\tv_exp_f32 %1, %6\n\tv_exp_f32 %2, %6\n\tv_exp_f32 %3, %6\n\tv_exp_f32 %4, %6\n\tv_exp_f32 %5, %6" : "=v"(e0), "=v"(e1), "=v"(e2), "=v"(e3), "=v"(e4), "=v"(e5) : "v"(x));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 8) {      // mfma + 12 independent v_add (6 chains of 2)
        float e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            asm volatile("v_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            asm volatile("v_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 9) {      // 24 independent v_add only (no mfma): the VALU-only cost of mode 8
        float e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x));
            asm volatile("v_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 10) {      // phase-separated: 8 mfma back to back, then 48 independent v_add (attention v2's shape)
        float e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
        for (int i = 0; i < iters / 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                asm volatile("v_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %6\n\tv_add_f32 %2, %2, %6\n\tv_add_f32 %3, %3, %6\n\tv_add_f32 %4, %4, %6\n\tv_add_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 11) {
        float e0 = 1, e1 = 2, e2 = 3, e3 = 4, e4 = 5, e5 = 6;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %6, %7\n\tv_max3_f32 %3, %3, %6, %7\n\tv_max3_f32 %4, %4, %6, %7\n\tv_max3_f32 %5, %5, %6, %7" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
            asm volatile("v_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %1, %1, %6, %7\n\tv_max3_f32 %2, %2, %6, %7\n\tv_max3_f32 %3, %3, %6, %7\n\tv_max3_f32 %4, %4, %6, %7\n\tv_max3_f32 %5, %5, %6, %7" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 12) {
        float e0 = 1, e1 = 2, e2 = 3, e3 = 4, e4 = 5, e5 = 6;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_bfe_i32 %0, %0, 3, 1\n\tv_bfe_i32 %1, %1, 3, 1\n\tv_bfe_i32 %2, %2, 3, 1\n\tv_bfe_i32 %3, %3, 3, 1\n\tv_bfe_i32 %4, %4, 3, 1\n\tv_bfe_i32 %5, %5, 3, 1" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
            asm volatile("v_bfe_i32 %0, %0, 3, 1\n\tv_bfe_i32 %1, %1, 3, 1\n\tv_bfe_i32 %2, %2, 3, 1\n\tv_bfe_i32 %3, %3, 3, 1\n\tv_bfe_i32 %4, %4, 3, 1\n\tv_bfe_i32 %5, %5, 3, 1" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 13) {
        float e0 = 1, e1 = 2, e2 = 3, e3 = 4, e4 = 5, e5 = 6;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_and_b32 %0, %0, %6\n\tv_and_b32 %1, %1, %6\n\tv_and_b32 %2, %2, %6\n\tv_and_b32 %3, %3, %6\n\tv_and_b32 %4, %4, %6\n\tv_and_b32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
            asm volatile("v_and_b32 %0, %0, %6\n\tv_and_b32 %1, %1, %6\n\tv_and_b32 %2, %2, %6\n\tv_and_b32 %3, %3, %6\n\tv_and_b32 %4, %4, %6\n\tv_and_b32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 14) {
        float e0 = 1, e1 = 2, e2 = 3, e3 = 4, e4 = 5, e5 = 6;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_cvt_pk_bf16_f32 %0, %0, %6\n\tv_cvt_pk_bf16_f32 %1, %1, %6\n\tv_cvt_pk_bf16_f32 %2, %2, %6\n\tv_cvt_pk_bf16_f32 %3, %3, %6\n\tv_cvt_pk_bf16_f32 %4, %4, %6\n\tv_cvt_pk_bf16_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
            asm volatile("v_cvt_pk_bf16_f32 %0, %0, %6\n\tv_cvt_pk_bf16_f32 %1, %1, %6\n\tv_cvt_pk_bf16_f32 %2, %2, %6\n\tv_cvt_pk_bf16_f32 %3, %3, %6\n\tv_cvt_pk_bf16_f32 %4, %4, %6\n\tv_cvt_pk_bf16_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 15) {
        float e0 = 1, e1 = 2, e2 = 3, e3 = 4, e4 = 5, e5 = 6;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_mov_b32 %0, %6\n\tv_mov_b32 %1, %6\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %6\n\tv_mov_b32 %4, %6\n\tv_mov_b32 %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
            asm volatile("v_mov_b32 %0, %6\n\tv_mov_b32 %1, %6\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %6\n\tv_mov_b32 %4, %6\n\tv_mov_b32 %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 16) {
        float e0 = 1, e1 = 2, e2 = 3, e3 = 4, e4 = 5, e5 = 6;
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_fma_f32 %0, %0, %6, %7\n\tv_fma_f32 %1, %1, %6, %7\n\tv_fma_f32 %2, %2, %6, %7\n\tv_fma_f32 %3, %3, %6, %7\n\tv_fma_f32 %4, %4, %6, %7\n\tv_fma_f32 %5, %5, %6, %7" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
            asm volatile("v_fma_f32 %0, %0, %6, %7\n\tv_fma_f32 %1, %1, %6, %7\n\tv_fma_f32 %2, %2, %6, %7\n\tv_fma_f32 %3, %3, %6, %7\n\tv_fma_f32 %4, %4, %6, %7\n\tv_fma_f32 %5, %5, %6, %7" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(x), "v"(y));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 17) {
        typedef __attribute__((ext_vector_type(2))) float f2;
        f2 e0 = {1, 1}, e1 = {2, 2}, e2 = {3, 3}, e3 = {4, 4}, e4 = {5, 5}, e5 = {6, 6}, xx = {x, x};
        for (int i = 0; i < iters; ++i) {
            asm volatile("v_pk_add_f32 %0, %0, %6\n\tv_pk_add_f32 %1, %1, %6\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %6\n\tv_pk_add_f32 %4, %4, %6\n\tv_pk_add_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(xx));
            asm volatile("v_pk_add_f32 %0, %0, %6\n\tv_pk_add_f32 %1, %1, %6\n\tv_pk_add_f32 %2, %2, %6\n\tv_pk_add_f32 %3, %3, %6\n\tv_pk_add_f32 %4, %4, %6\n\tv_pk_add_f32 %5, %5, %6" : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(xx));
        }
        y += e0[0] + e1[0] + e2[1] + e3[0] + e4[1] + e5[0];
    } else if (mode == 18) {      // mfma, then 12 VALU that READ the other (finished) accumulator - the attention loop's situation
        float e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0;
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            asm volatile("v_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %7\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %9\n\tv_add_f32 %4, %4, %10\n\tv_add_f32 %5, %5, %11\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %7\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %9\n\tv_add_f32 %4, %4, %10\n\tv_add_f32 %5, %5, %11"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(acc1[0]), "v"(acc1[1]), "v"(acc1[2]), "v"(acc1[3]), "v"(acc1[4]), "v"(acc1[5]));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            asm volatile("v_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %7\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %9\n\tv_add_f32 %4, %4, %10\n\tv_add_f32 %5, %5, %11\n\tv_add_f32 %0, %0, %6\n\tv_add_f32 %1, %1, %7\n\tv_add_f32 %2, %2, %8\n\tv_add_f32 %3, %3, %9\n\tv_add_f32 %4, %4, %10\n\tv_add_f32 %5, %5, %11"
                         : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5) : "v"(acc0[0]), "v"(acc0[1]), "v"(acc0[2]), "v"(acc0[3]), "v"(acc0[4]), "v"(acc0[5]));
        }
        y += e0 + e1 + e2 + e3 + e4 + e5;
    } else if (mode == 19) {      // the same with the 12 VALU WRITING the B operand of the next mfma (P -> PV)
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        u4 pb = {1, 2, 3, 4};
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, pb), acc0, 0, 0, 0);
            asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %7\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %7\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %7"
                         : "+v"(pb[0]), "+v"(pb[1]), "+v"(pb[2]), "+v"(pb[3]) : "v"(acc1[0]), "v"(acc1[1]), "v"(acc1[2]), "v"(acc1[3]));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, pb), acc1, 0, 0, 0);
            asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %7\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %7\n\tv_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %5\n\tv_add_u32 %2, %2, %6\n\tv_add_u32 %3, %3, %7"
                         : "+v"(pb[0]), "+v"(pb[1]), "+v"(pb[2]), "+v"(pb[3]) : "v"(acc0[0]), "v"(acc0[1]), "v"(acc0[2]), "v"(acc0[3]));
        }
        y += (float)pb[0];
    } else if (mode == 20) {     // the row-block kernels' step: mfma, one ds_read_b128 refilling the operand ring 8 steps ahead, counted wait
        bf16x8 ring[8];
        for (int j = 0; j < 8; ++j) ring[j] = a;
        const unsigned la = (threadIdx.x & 63) * 16;
        for (int i = 0; i < iters / 4; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[k], b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[k], b, acc0, 0, 0, 0);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[k]) : "v"(la), "n"(k * 1024));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (mode == 21) {     // mfma + a pack-like dependent chain per step: 2 fma on values of the OTHER accumulator, 1 cvt_pk
        float e0 = 0, e1 = 0;
        unsigned pk = 0;
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %3, %5, %5\n\tv_fma_f32 %1, %4, %5, %5\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "=&v"(e0), "=&v"(e1), "=v"(pk) : "v"(acc1[2]), "v"(acc1[3]), "v"(x));
            y += __builtin_bit_cast(float, pk);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %3, %5, %5\n\tv_fma_f32 %1, %4, %5, %5\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "=&v"(e0), "=&v"(e1), "=v"(pk) : "v"(acc0[2]), "v"(acc0[3]), "v"(x));
            y += __builtin_bit_cast(float, pk);
        }
    } else if (mode == 22) {     // mfma + 3 independent VALU on plain registers per step (the same count as mode 21, no accumulator reads)
        float e0 = 0, e1 = 0, e2 = 0;
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %3, %3, %0\n\tv_fma_f32 %1, %3, %3, %1\n\tv_fma_f32 %2, %3, %3, %2" : "+v"(e0), "+v"(e1), "+v"(e2) : "v"(x));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %3, %3, %0\n\tv_fma_f32 %1, %3, %3, %1\n\tv_fma_f32 %2, %3, %3, %2" : "+v"(e0), "+v"(e1), "+v"(e2) : "v"(x));
        }
        y += e0 + e1 + e2;
    } else if (mode == 23) {     // mode 20 + mode 22 in one step: mfma, ring refill, 3 independent VALU
        bf16x8 ring[8];
        for (int j = 0; j < 8; ++j) ring[j] = a;
        const unsigned la = (threadIdx.x & 63) * 16;
        float e0 = 0, e1 = 0, e2 = 0;
        for (int i = 0; i < iters / 4; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
                if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[k], b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[k], b, acc0, 0, 0, 0);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ring[k]) : "v"(la), "n"(k * 1024));
                asm volatile("v_fma_f32 %0, %3, %3, %0\n\tv_fma_f32 %1, %3, %3, %1\n\tv_fma_f32 %2, %3, %3, %2" : "+v"(e0), "+v"(e1), "+v"(e2) : "v"(x));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        y += e0 + e1 + e2;
    } else if (mode == 24) {     // mode 20 without the counted wait in front of every mfma (the compiler's own waits only)
        bf16x8 ring[8];
        for (int j = 0; j < 8; ++j) ring[j] = a;
        const bf16x8* lp = reinterpret_cast<const bf16x8*>(lds_probe) + (threadIdx.x & 63);
        for (int i = 0; i < iters / 4; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[k], b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[k], b, acc0, 0, 0, 0);
                ring[k] = lp[k * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (mode == 25) {     // mode 21 on a THIRD accumulator no mfma in flight writes
        float e0 = 0, e1 = 0;
        unsigned pk = 0;
        f32x16 acc2;
        for (int j = 0; j < 16; ++j) acc2[j] = x * j;
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %3, %5, %5\n\tv_fma_f32 %1, %4, %5, %5\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "=&v"(e0), "=&v"(e1), "=v"(pk) : "v"(acc2[2]), "v"(acc2[3]), "v"(x));
            y += __builtin_bit_cast(float, pk);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %3, %5, %5\n\tv_fma_f32 %1, %4, %5, %5\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "=&v"(e0), "=&v"(e1), "=v"(pk) : "v"(acc2[6]), "v"(acc2[7]), "v"(x));
            y += __builtin_bit_cast(float, pk);
        }
    } else if (mode == 26) {     // mode 25 with the third accumulator parked in AGPRs: 2 v_accvgpr_read in front of the chain
        float e0 = 0, e1 = 0, r0v = 0, r1v = 0;
        unsigned pk = 0;
        float g0 = x, g1 = x * 2, g2 = x * 3, g3 = x * 4;
        asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3" :: "v"(g0), "v"(g1), "v"(g2), "v"(g3) : "a0", "a1", "a2", "a3");
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
            asm volatile("v_accvgpr_read_b32 %3, a0\n\tv_accvgpr_read_b32 %4, a1\n\tv_fma_f32 %0, %3, %5, %5\n\tv_fma_f32 %1, %4, %5, %5\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "=&v"(e0), "=&v"(e1), "=v"(pk), "=&v"(r0v), "=&v"(r1v) : "v"(x));
            y += __builtin_bit_cast(float, pk);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
            asm volatile("v_accvgpr_read_b32 %3, a2\n\tv_accvgpr_read_b32 %4, a3\n\tv_fma_f32 %0, %3, %5, %5\n\tv_fma_f32 %1, %4, %5, %5\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "=&v"(e0), "=&v"(e1), "=v"(pk), "=&v"(r0v), "=&v"(r1v) : "v"(x));
            y += __builtin_bit_cast(float, pk);
        }
    } else if (mode == 27) {     // mode 25 + an LDS store of the packed words every 4th step and a global store every 8th
        float e0 = 0, e1 = 0;
        unsigned pk = 0;
        f32x16 acc2;
        for (int j = 0; j < 16; ++j) acc2[j] = x * j;
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
        u32x4 hn = {0, 0, 0, 0};
        const unsigned la = (threadIdx.x & 63) * 16;
        for (int i = 0; i < iters / 2; ++i) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (k & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
                asm volatile("v_fma_f32 %0, %3, %5, %5\n\tv_fma_f32 %1, %4, %5, %5\n\tv_cvt_pk_bf16_f32 %2, %0, %1" : "=&v"(e0), "=&v"(e1), "=v"(pk) : "v"(acc2[2 * k]), "v"(acc2[2 * k + 1]), "v"(x));
                hn[k] = pk;
                if (k == 3) asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(hn) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {   // mode 3: MFMA with 6 independent VALU in its shadow
        for (int i = 0; i < iters; ++i) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 6; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = r1 - r0; }
    float s = x + y;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    if (s == 12345.678f) sink[0] = s;
}
int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 512;
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * 16); hipMalloc(&sink, 4);
    unsigned long long* h = (unsigned long long*)malloc(blocks * 16);
    const char* names[] = {"mfma x2 (2 chains) / iter", "16 dependent v_add / iter", "16 v_exp / iter", "2 x (mfma + 6 v_add) / iter", "mfma x2 (1 chain) / iter", "mfma x4 (4 chains) / iter", "mfma16x16x32 x4 (4 chains)", "2 x (mfma + 6 v_exp)", "2 x (mfma + 12 indep v_add)", "24 indep v_add", "[8 mfma, then 96 indep v_add] / 4 iters", "12 x v_max3_f32 (6 chains)", "12 x v_bfe_i32 (6 chains)", "12 x v_and_b32 (6 chains)", "12 x v_cvt_pk_bf16_f32 (6 chains)", "12 x v_mov_b32 (6 regs)", "12 x v_fma_f32 (6 chains, VOP3)", "12 x v_pk_add_f32 (6 chains)", "2 x (mfma + 12 v_add reading the other accumulator)", "2 x (mfma + 12 VALU writing the next mfma B operand)", "8 x (wait, mfma, ds_read_b128 ring refill) / 4 iters", "2 x (mfma + fma,fma,cvt_pk on the other accumulator)", "2 x (mfma + 3 indep fma)", "8 x (wait, mfma, ds_read ring, 3 fma) / 4 iters", "8 x (mfma, ds_read ring; compiler waits) / 4 iters", "2 x (mfma + fma,fma,cvt_pk on an idle third accumulator)", "2 x (mfma + 2 accvgpr_read,fma,fma,cvt_pk from idle AGPRs)", "4 x (mfma + fma,fma,cvt_pk idle acc; ds_write_b128 per 4) / 2 iters"};
    const int threads = argc > 2 ? atoi(argv[2]) : 256;
    for (int waves = 1; waves <= 2; ++waves)
    for (int mode = (argc > 3 ? atoi(argv[3]) : 0); mode < 28; ++mode) {
        const int iters = 20000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(probe, dim3(blocks * waves), dim3(threads), 0, 0, mode, iters, out, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, out, blocks * 16, hipMemcpyDeviceToHost);
        double c = 0, r = 0;
        for (int i = 0; i < blocks; ++i) { c += h[2 * i]; r += h[2 * i + 1]; }
        c /= blocks; r /= blocks;
        printf("%-30s blocks=%d: kernel %.3f ms; per wave: %.0f shader cycles, %.0f refclk ticks (100 MHz) -> %.3f GHz; cycles/iter %.1f\n",
               names[mode], blocks * waves, ms, c, r, c / (r * 10.0), c / iters);
    }
    return 0;
}
