// Shared device/host helpers for libasr_hip.so (gfx950 / CDNA4 only: wave = 64, MFMA, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/asr_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define ASR_WAVE 64

// ---- error plumbing (thread-local message, no exceptions across the ABI) ------------------------------------
void asr_set_error(const char* fmt, ...);
// asr_proj_heads for encoder-sized bf16 rows (ffn.hip; called from gemm.hip): 0 = launched, -2 = not its shape
int asr_proj_heads_rows(hipStream_t stream, const void* X, const void* W, const float* bias, void* out, int64_t proj_stride, int n_proj, int B,
                        int L, int h, float scale_first);
// the encoder's self-attention forward (attention_fwd4.hip; called from attention.hip): 0 = launched, -2 = not its case
int asr_attention_fwd_v4(hipStream_t s, const void* q, const void* k, const void* v, void* ctx, float* lse, int B, int h, int Lq, int Lk,
                         const int32_t* k_len, asr_dropout_t drop, const uint32_t* drop_bits);
int asr_attention_bwd_dkv_v4(hipStream_t s, const void* q, const void* k, const void* v, const void* d_o, const float* nscal, void* dk,
                             void* dv, int64_t ldkv, int B, int h, int Lq, int Lk, const int32_t* k_len, asr_dropout_t drop,
                             const uint32_t* drop_bits);     // attention_bwd4.hip: 0 = launched, -2 = not its case
int asr_attention_bwd_dq_v4(hipStream_t s, const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                            float* nscal, void* dq, int64_t ldq, int B, int h, int Lq, int Lk, const int32_t* k_len, float scale,
                            asr_dropout_t drop, const uint32_t* drop_bits);
// collective.hip: the function of the bucket-ready marker node (graph_exec.hip turns such a node into an all-reduce call) and the call
const void* asr_collective_marker_func();
// asr_ffn_fwd's launch (ffn2.hip; arguments checked by ffn.hip)
struct asr_ffn2_pre_t {      // asr_attn_ffn_fwd: the attention sub-layer's tail run in front of the feed-forward one (ffn2.hip, PRE)
    const void* ctx16; const float* res32; const void* w; const float* bias; const float* gamma; const float* beta;
    float* s_out; float* mean_out; float* rstd_out; float eps; asr_dropout_t drop_x;
};
int asr_ffn_fwd2_launch(hipStream_t stream, const void* x16, const float* x32, const void* w1, const float* b1, const void* w2, const float* b2,
                        const float* gamma, const float* beta, const int32_t* row_len, void* hid_out, void* bits_out, float* s_out, float* y32,
                        void* y16, float* mean_out, float* rstd_out, int M, int L, int d_ff, float eps, asr_dropout_t drop_x,
                        const asr_ffn2_pre_t* pre = nullptr);
int asr_launch_budget_current();     // common.hip: asr_launch_budget (0 = none)
int asr_deterministic();     // common.hip: ASR_AMD_DETERMINISTIC / asr_set_deterministic
#define ASR_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            asr_set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)
#define ASR_LAUNCH_CHECK(name)                                          \
    do {                                                                \
        hipError_t e__ = hipGetLastError();                             \
        if (e__ != hipSuccess) {                                        \
            asr_set_error("%s: %s", name, hipGetErrorString(e__));      \
            return (int)e__;                                            \
        }                                                               \
    } while (0)

static inline bool asr_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// ---- device helpers ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- dropout (asr_hip.h: asr_dropout_t) ---------------------------------------------------------------------
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
// asr_dropout_t.salt (asr_hip.h): resolve the per-replay keys once at kernel entry (one scalar load)
__device__ __forceinline__ asr_dropout_t drop_resolve(asr_dropout_t d) {
    if (d.thr16 != 0 && d.salt != nullptr) {
        const uint32_t s = *d.salt;
        d.key0 ^= lowbias32(s ^ 0x5bd1e995u);
        d.key1 ^= lowbias32(s + 0x27d4eb2fu);
    }
    return d;
}
__device__ __forceinline__ uint32_t drop_subkey(const asr_dropout_t& d, uint32_t n0) { return lowbias32(n0 * 0x9E3779B9u + d.key0); }
// random word of element pair `pair` = n1 * ceil(N2/2) + (n2 >> 1): low half decides even n2, high half odd n2
__device__ __forceinline__ uint32_t drop_word(const asr_dropout_t& d, uint32_t sub, uint32_t pair) { return lowbias32(pair ^ sub) ^ d.key1; }
__device__ __forceinline__ bool drop_keep_lo(const asr_dropout_t& d, uint32_t w) { return (w & 0xFFFFu) >= d.thr16; }
__device__ __forceinline__ bool drop_keep_hi(const asr_dropout_t& d, uint32_t w) { return (w >> 16) >= d.thr16; }
__device__ __forceinline__ float drop_scale(const asr_dropout_t& d) { return 65536.f / (float)(65536u - d.thr16); }
// ---- attention dropout mask as bit images (attention.hip: attn_dropmask_kernel) ----
// One hash pass per layer call writes the keep bits twice, the attention kernels (forward, dQ, dK/dV) then read two 32-bit words
// per lane and 64-key (64-query) tile instead of hashing 16 words each: the hash is 2 quarter-rate multiplies + 3 xor-shifts per
// 2 elements, ~1.8x the cost of the exp2 of the same 2 elements, and was paid three times.
//   Mk[bh][kw][q]  (kw = key / 32):  bit (key & 31)  = keep(q, key)     q-stationary kernels: lane = query, coalesced along q
//   Mq[bh][qw][key] (qw = q / 32):   bit (q & 31)    = keep(q, key)     k-stationary kernel:  lane = key,   coalesced along key
// with Lq / Lk rounded up to 128 (words of all-padding tiles are unspecified: every consumer masks those positions itself).
// (Tried: images of 64-bit wave lane masks, scalar-loaded and applied with one v_cndmask per element via inverse ballot - 1 VALU
// instead of 2 per element.  -6 us on the forward kernel alone, but +8 / +27 us on dQ / dK,dV: scalar loads share lgkmcnt with
// the LDS reads those kernels keep in flight, and a wait on an out-of-order SMEM return drains them all.)
__host__ __device__ __forceinline__ int drop_pad128(int n) { return (n + 127) & ~127; }
__host__ __device__ __forceinline__ int64_t drop_mk_words(int BH, int Lq, int Lk) { return (int64_t)BH * (drop_pad128(Lk) / 32) * drop_pad128(Lq); }
// all-ones / all-zeros from bit `pos` of w: the value is AND-ed onto the f32 bit pattern it keeps or drops
__device__ __forceinline__ float drop_and(float v, uint32_t w, int pos) {
    // (through asm: written with __builtin_amdgcn_sbfe the compiler rewrites bfe + and as and + cmp + cndmask - three VALU
    // instructions per probability in kernels that are VALU-bound)
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "n"(pos));
    return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & m);
}
// 4 consecutive elements n2 = c .. c+3 (c even) of row n1: multiply by the keep mask * scale
__device__ __forceinline__ f32x4 drop4(const asr_dropout_t& d, uint32_t sub, uint32_t n1, uint32_t n2h, uint32_t c, f32x4 v, float sc) {
    const uint32_t pair = n1 * n2h + (c >> 1);
    const uint32_t w0 = drop_word(d, sub, pair), w1 = drop_word(d, sub, pair + 1);
    v[0] = drop_keep_lo(d, w0) ? v[0] * sc : 0.f;
    v[1] = drop_keep_hi(d, w0) ? v[1] * sc : 0.f;
    v[2] = drop_keep_lo(d, w1) ? v[2] * sc : 0.f;
    v[3] = drop_keep_hi(d, w1) ? v[3] * sc : 0.f;
    return v;
}

// online (max, sum-exp) pair combine
__device__ __forceinline__ void lse_combine(float& m, float& s, float m2, float s2) {
#pragma clang fp contract(off)      // (one rounding per operation wherever this is inlined: callers compare forms of one op bit for bit)
    float mn = fmaxf(m, m2);
    if (mn == -INFINITY) { s = 0.f; m = mn; return; }
    s = s * __expf(m - mn) + s2 * __expf(m2 - mn);
    m = mn;
}

template <typename T> struct DType;
template <> struct DType<float> { static constexpr int code = ASR_F32; };
template <> struct DType<bf16_t> { static constexpr int code = ASR_BF16; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }

// ---- MFMA 16x16 step over one 4-chunk group of K (a 64-byte slice of K per row) -----------------------------
// Each lane supplies the 16-byte chunk (lane>>4) of row (lane&15) for both operands.  bf16: one
// v_mfma_f32_16x16x32_bf16;  f32: four exact v_mfma_f32_16x16x4_f32 (element j of each lane's float4 pairs up).
// D[row][col]: row index comes from operand `a`, col from operand `b`; lane holds col = lane&15, rows (lane>>4)*4+reg.
template <typename CT> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const u32x4& a, const u32x4& b, f32x4& c) {
        f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j], fb[j], c, 0, 0, 0);
    }
};


// ---- 4x4 bf16 register transpose on v_perm_b32 ---------------------------------------------------------------
// rows r[0..3] hold 4 bf16 each (two dwords); column d of the result is { r0[d], r1[d], r2[d], r3[d] } packed the same way.
// 8 v_perm_b32 instead of ~30 shift/and/or operations; used by every loader that builds a K-contiguous LDS image from
// row-major data whose reduction index is the slow dimension (V^T, dgrad / wgrad GEMM operands, attention backward tiles).
__device__ __forceinline__ void transpose4x4_bf16(const u32x2 (&r)[4], u32x2 (&c)[4]) {
    c[0] = u32x2{__builtin_amdgcn_perm(r[1][0], r[0][0], 0x05040100u), __builtin_amdgcn_perm(r[3][0], r[2][0], 0x05040100u)};
    c[1] = u32x2{__builtin_amdgcn_perm(r[1][0], r[0][0], 0x07060302u), __builtin_amdgcn_perm(r[3][0], r[2][0], 0x07060302u)};
    c[2] = u32x2{__builtin_amdgcn_perm(r[1][1], r[0][1], 0x05040100u), __builtin_amdgcn_perm(r[3][1], r[2][1], 0x05040100u)};
    c[3] = u32x2{__builtin_amdgcn_perm(r[1][1], r[0][1], 0x07060302u), __builtin_amdgcn_perm(r[3][1], r[2][1], 0x07060302u)};
}

// ---- LDS-DMA / small loads as hand-counted inline asm (ffn.hip backward, vocab.hip) ---------------------------------------------
// The compiler orders every LDS read it cannot disambiguate behind
// ALL pending LDS-DMA: with the builtin form each ds_read_b64_tr_b16 of the loop got an s_waitcnt vmcnt(0) in front (the fused
// backward ran 200 us instead of 90).  An asm statement's memory operations are invisible to that pass; the waits are the kernel's
// own counted ones (vmcnt retires in order: DMA first, then the stores, exactly as in the forward).
__device__ __forceinline__ u32x4 rsrc_words(const void* base, unsigned bytes) {      // raw buffer descriptor: stride 0, `bytes` records
    const uint64_t b = (uint64_t)base;
    return u32x4{(unsigned)b, (unsigned)(b >> 32) & 0xffffu, bytes, 0x00020000u};
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)((__attribute__((address_space(3))) const unsigned char*)p);
}
__device__ __forceinline__ void dma16_asm(u32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x4 load128_asm(u32x4 rsrc, unsigned voff, unsigned soff) {
    u32x4 v;
    asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return v;
}
// two 16-bit loads (same lane offset, two scalar offsets), zero-extended (no d16_hi form: with SRAM ECC a d16 load clears the other half)
__device__ __forceinline__ u32x2 load16x2_asm(u32x4 rsrc, unsigned voff, unsigned soff_lo, unsigned soff_hi) {
    uint32_t lo, hi;
    asm volatile("s_nop 4\n\tbuffer_load_ushort %0, %2, %3, %4 offen\n\tbuffer_load_ushort %1, %2, %3, %5 offen"
                 : "=&v"(lo), "=&v"(hi) : "v"(voff), "s"(rsrc), "s"(soff_lo), "s"(soff_hi) : "memory");
    return u32x2{lo, hi};
}
__device__ __forceinline__ uint32_t load32_asm(u32x4 rsrc, unsigned voff, unsigned soff) {
    uint32_t v;
    asm volatile("s_nop 4\n\tbuffer_load_dword %0, %1, %2, %3 offen" : "=v"(v) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    return v;
}


