// Diagnostic (not part of the product): what the shader clock really is under load.  Each wave runs a loop of MFMAs (mode 0), of
// dependent VALU adds (mode 1) or of exp2 (mode 2) and brackets it with s_memtime (shader clock) and s_memrealtime (100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(256) void probe(int mode, int iters, unsigned long long* out, float* sink) {
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
    const char* names[] = {"mfma x2 (2 chains) / iter", "16 dependent v_add / iter", "16 v_exp / iter", "2 x (mfma + 6 v_add) / iter", "mfma x2 (1 chain) / iter", "mfma x4 (4 chains) / iter", "mfma16x16x32 x4 (4 chains)", "2 x (mfma + 6 v_exp)", "2 x (mfma + 12 indep v_add)", "24 indep v_add", "[8 mfma, then 96 indep v_add] / 4 iters", "12 x v_max3_f32 (6 chains)", "12 x v_bfe_i32 (6 chains)", "12 x v_and_b32 (6 chains)", "12 x v_cvt_pk_bf16_f32 (6 chains)", "12 x v_mov_b32 (6 regs)", "12 x v_fma_f32 (6 chains, VOP3)", "12 x v_pk_add_f32 (6 chains)", "2 x (mfma + 12 v_add reading the other accumulator)", "2 x (mfma + 12 VALU writing the next mfma B operand)"};
    const int threads = argc > 2 ? atoi(argv[2]) : 256;
    for (int waves = 1; waves <= 2; ++waves)
    for (int mode = (argc > 3 ? atoi(argv[3]) : 0); mode < 20; ++mode) {
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
