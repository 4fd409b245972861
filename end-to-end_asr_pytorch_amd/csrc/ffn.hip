// Position-wise feed-forward sub-layer of an encoder layer as ONE forward launch and ONE data-gradient launch (d_model = 256):
//   forward   y = LayerNorm(dropout(relu(x W1^T + b1) W2^T + b2) + x) [* non_pad_mask]      src/transformer/module.py:48-53, encoder.py:77
//   backward  dH = (ds W2) * relu'(H),  dX = dH W1 + ds                                       (autograd of the same lines)
// The [M, d_ff] hidden activation never makes a round trip through HBM inside either launch: a workgroup owns 128 tokens, keeps
// their 256-wide rows in REGISTERS as MFMA B operands, and streams W1 / W2 in 64-unit chunks of the hidden dimension through a
// double-buffered LDS image (LDS-DMA, every workgroup reads the same 2 MB of weights from its XCD's L2).  Per chunk and wave
// (32 tokens, one wave per SIMD, the whole 512-register file):
//   S^T[64 hidden x 32 tok]  = W1c . X^T             (v_mfma_f32_32x32x16_bf16, K = 256; accumulator starts at b1)
//   H^T = relu(S^T) -> bf16 in the accumulator's own registers = the B operand of the second product (no LDS, no lane movement:
//         rows of a 32x32 accumulator are the k index of the next MFMA; the hidden units are dealt to MFMA rows with bits 2 and 3
//         of the row index swapped so that the k order the second product sees is the natural one)
//   Y^T[256 x 32 tok]       += W2c . H^T             (K = 64)
// The first product of chunk i and the second of chunk i - 1 share a loop iteration, so the MFMA pipe never waits for the ReLU.
// Training also writes H once (bf16, for the weight gradient dW2 = ds^T H) and the ReLU mask as 1 bit per unit; the epilogue is
// bias + dropout + residual + LayerNorm over the complete 256-wide row each lane pair holds (outputs: pre-norm sum, y32, y16, mean,
// rstd - exactly what asr_gemm_nt x 2 + asr_add_layernorm_fwd leave).
#include "asr_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int FBM = 128;      // tokens per workgroup (4 waves x 32)
constexpr int FHC = 64;       // hidden units per chunk
constexpr int FD = 256;       // d_model
constexpr int W1BUF = FHC * FD * 2;   // 32 KiB: [64 hidden][256 k] bf16, 512-byte rows
constexpr int W2BUF = FD * FHC * 2;   // 32 KiB: [256 d][64 hidden] bf16, 128-byte rows
constexpr int FFN_MAX_DFF = 2048;
constexpr int HST_BYTES = 4 * 4096;   // per wave: a chunk's H^T tile (32 tokens x 128 B) on its way to full-line stores
constexpr int SMEM_BYTES = 2 * W1BUF + 2 * W2BUF + HST_BYTES + FFN_MAX_DFF * 4;

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// mask word of one (token, lane half, chunk): bit P (P = 8 t + p) = unit 2P positive, bit 16 + P = unit 2P + 1 positive, where a lane's
// 32 units of a chunk are numbered e = 16 t + j (t: 32-row tile, j: accumulator register)
__device__ __forceinline__ int64_t bits_index(int chunk, int h, int Mp, int m) { return ((int64_t)(chunk * 2 + h)) * Mp + m; }

struct FfnFwdArgs {
    const bf16_t* x16;
    const float* x32;
    const bf16_t* w1;
    const float* b1;
    const bf16_t* w2;
    const float* b2;
    const float* gamma;
    const float* beta;
    const int32_t* row_len;
    bf16_t* hid;
    uint32_t* bits;
    float* s_out;
    float* y32;
    bf16_t* y16;
    float* mean;
    float* rstd;
    int M, L, dff, Mp;
    float eps;
    asr_dropout_t drop;
    int dbg;      // ASR_AMD_FFN_DBG (timing breakdowns only): 1 = no epilogue, 2 = no LDS-DMA in the loop
};

template <bool TRAIN>
__global__ __launch_bounds__(256, 1) void ffn_fwd_kernel(const FfnFwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    unsigned char* const w1s = smem;
    unsigned char* const w2s = smem + 2 * W1BUF;
    float* const b1s = reinterpret_cast<float*>(smem + 2 * W1BUF + 2 * W2BUF + HST_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const hst = smem + 2 * W1BUF + 2 * W2BUF + wave * 4096;
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * FBM + wave * 32 + r;
    const bool valid = m < a.M;
    const int mc = valid ? m : a.M - 1;
    const int dff = a.dff, NC = dff / FHC;

    // ---- weight staging: buffer descriptors over the whole matrices, per-lane byte offsets fixed for the kernel's life --------------
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w1), 0, dff * FD * 2, 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w2), 0, dff * FD * 2, 0x00020000);
    unsigned off1[8], off2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = wave * 8 + k;
        {   // W1 piece p: chunk rows 2p, 2p + 1 (512 B each); LDS slot pc of row u holds 16-byte chunk (pc & 16) | ((pc ^ u) & 15)
            const int u = 2 * p + (lane >> 5), pc = lane & 31;
            const int c = (pc & 16) | ((pc ^ u) & 15);
            off1[k] = (unsigned)(u * FD * 2 + c * 16);
        }
        {   // W2 piece p: rows 8p .. 8p + 7 of the [256][64] chunk image (128 B each); slot pc of row d holds chunk pc ^ ((d >> 1) & 7)
            const int d = 8 * p + (lane >> 3), pc = lane & 7;
            const int c = pc ^ ((d >> 1) & 7);
            off2[k] = (unsigned)(d * dff * 2 + c * 16);
        }
    }
    auto stage_w1 = [&](int buf, int chunk) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_void*)(w1s + buf * W1BUF + (wave * 8 + k) * 1024), 16, off1[k],
                                                     chunk * (FHC * FD * 2), 0, 0);
    };
    auto stage_w2 = [&](int buf, int chunk) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_void*)(w2s + buf * W2BUF + (wave * 8 + k) * 1024), 16, off2[k],
                                                     chunk * (FHC * 2), 0, 0);
    };

    // ---- fragment read addresses -----------------------------------------------------------------------------------------------
    const int u15 = swap23(r) & 15;
    unsigned a1[8], a2[4];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a1[kk] = (unsigned)(swap23(r) * 512 + (((2 * kk + h) ^ u15) << 4));
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) a2[sg] = (unsigned)(r * 128 + (((2 * sg + h) ^ ((r >> 1) & 7)) << 4));

    // ---- this lane's token as the first product's B operand: X[m][16 ks + 8 h .. + 8] --------------------------------------------
    bf16x8 xb[16];
    {
        const bf16_t* xr = a.x16 + (int64_t)mc * FD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
    }
    for (int i = tid * 4; i < dff; i += 1024) *reinterpret_cast<f32x4*>(b1s + i) = *reinterpret_cast<const f32x4*>(a.b1 + i);

    stage_w1(0, 0);
    f32x16 Y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) Y[t][j] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    u32x4 Hf[4];        // H^T of the previous chunk as four k-steps of the second product's B operand
    const auto rsh = __builtin_amdgcn_make_buffer_rsrc(a.hid, 0, TRAIN ? (int)((int64_t)a.M * dff * 2) : 0, 0x00020000);
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc(a.bits, 0, TRAIN ? (int)((int64_t)NC * 2 * a.Mp * 4) : 0, 0x00020000);
    const unsigned boff = valid ? ((unsigned)h * a.Mp + m) * 4u : 0x80000000u;     // rows past M: an offset no later add brings back in range - the range check drops the store
    // H leaves through LDS: a lane holds 16-byte pieces of ITS token's row (a store instruction would touch 32 rows, 32 bytes each);
    // written to the wave's [32 tokens][128 B] tile (16-byte slot ^ (token & 7)) and read back 8 lanes per token, a store instruction
    // covers 8 tokens x one full 128-byte line.  The tile of chunk i is flushed during iteration i + 1.
    const unsigned hwr = (unsigned)(r * 128), hrd = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
    unsigned hoff[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int mt = blockIdx.x * FBM + wave * 32 + 8 * ps + (lane >> 3);
        hoff[ps] = mt < a.M ? (unsigned)mt * (unsigned)dff * 2u + 16u * (lane & 7) : 0x80000000u;
    }

    // The loop body is pinned step by step (one MFMA per step, __builtin_amdgcn_sched_barrier(0) between steps - a wave issues in
    // order, so whatever should run in an MFMA's shadow has to sit right behind it in the instruction stream):
    //   steps  0..31  first product of chunk i:     MFMA k, then the LDS read of the fragment MFMA k + 8 will take (ring of 16)
    //   steps 32..63  second product of chunk i-1:  MFMA, LDS read 8 ahead, and one slice of chunk i's ReLU / pack / mask work
    bf16x8 A[16];
    typedef __attribute__((ext_vector_type(2))) short s16x2_t;
    auto frag1 = [&](const unsigned char* w1, int k) {      // first product, MFMA k: k-step k >> 1, row tile k & 1
        const int ks = k >> 1, t = k & 1;
        return *reinterpret_cast<const bf16x8*>(w1 + a1[ks & 7] + t * 16384 + (ks >> 3) * 256);
    };
    auto frag2 = [&](const unsigned char* w2, int k) {      // second product, MFMA k: k-step k >> 3, row tile k & 7 of Y^T
        return *reinterpret_cast<const bf16x8*>(w2 + a2[k >> 3] + (k & 7) * 4096);
    };
    auto init_s = [&](int chunk, f32x16 (&S)[2]) {
        const float* bb = b1s + chunk * FHC + 8 * h;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(bb + 32 * t), q1 = *reinterpret_cast<const f32x4*>(bb + 32 * t + 4);
            const f32x4 q2 = *reinterpret_cast<const f32x4*>(bb + 32 * t + 16), q3 = *reinterpret_cast<const f32x4*>(bb + 32 * t + 20);
#pragma unroll
            for (int j = 0; j < 4; ++j) { S[t][j] = q0[j]; S[t][4 + j] = q1[j]; S[t][8 + j] = q2[j]; S[t][12 + j] = q3[j]; }
        }
    };
    // ReLU + bf16 of one register pair (P = 8 t + p) on the packed pair as a signed 16-bit max (a negative bf16 is a negative int16):
    // one instruction per pair and no canonicalising v_max in front of an fmaxf of MFMA results
    auto relu_pair = [&](const f32x16 (&S)[2], u32x4 (&Hn)[4], int P) {
        const int t = P >> 3, p = P & 7;
        uint32_t pk;       // (no builtin for the two-source form; written as two casts the compiler emits two converts and a v_perm)
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(S[t][2 * p]), "v"(S[t][2 * p + 1]));
        const s16x2_t rl = __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, pk), s16x2_t{0, 0});
        Hn[2 * t + (p >> 2)][p & 3] = __builtin_bit_cast(uint32_t, rl);
    };
    auto mask_pair = [&](const u32x4 (&Hn)[4], uint32_t& word, int P) {      // bit 15 / 31 of w + 0x7fff7fff: that half of w is not zero
        const int t = P >> 3, p = P & 7;
        word = (word >> 1) | ((Hn[2 * t + (p >> 2)][p & 3] + 0x7fff7fffu) & 0x80008000u);
        asm volatile("" : "+v"(word));      // keeps the three instructions in this step (pure arithmetic otherwise sinks to its one use)
    };
#define FFN_STEP() __builtin_amdgcn_sched_barrier(0)
#define FFN_WAIT_STAGE(NST)                                                                                        \
    do {                                                                                                           \
        /* the LDS-DMA of this iteration is older than its NST stores (vmcnt retires in order): wait for it only */ \
        /* (and every LDS read of the buffers the next iteration's DMA overwrites has returned) */                 \
        if (TRAIN) asm volatile("s_waitcnt vmcnt(" #NST ") lgkmcnt(0)" ::: "memory");                              \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                           \
        __builtin_amdgcn_s_barrier();                                                                              \
        asm volatile("" ::: "memory");                                                                             \
    } while (0)

    // one iteration: FIRST = no second product yet (chunk 0), LAST = no first product any more (after the last chunk)
    auto body = [&](int i, auto first_c, auto last_c) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        const unsigned char* w1 = w1s + (i & 1) * W1BUF;
        const unsigned char* w2 = w2s + ((i - 1) & 1) * W2BUF;
        f32x16 S[2];
        u32x4 Hn[4], Hout[4];
        uint32_t word = 0;
        if constexpr (!LAST) {
            init_s(i, S);
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = frag1(w1, k);
            FFN_STEP();
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                S[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k & 15], xb[k >> 1], S[k & 1], 0, 0, 0);
                if (k + 8 < 32) A[(k + 8) & 15] = frag1(w1, k + 8);
                else if (!FIRST) A[(k + 8) & 15] = frag2(w2, k + 8 - 32);
                if (TRAIN && !FIRST) {      // chunk i - 1's tile: four row-wise reads, four full-line stores
                    if (k >= 4 && k < 8) Hout[k - 4] = *reinterpret_cast<const u32x4*>(hst + hrd + (k - 4) * 1024);
                    if (k >= 16 && k < 20) __builtin_amdgcn_raw_buffer_store_b128(Hout[k - 16], rsh, hoff[k - 16], (i - 1) * (FHC * 2), 0);
                }
                FFN_STEP();
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = frag2(w2, k);
            FFN_STEP();
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if constexpr (!FIRST) {
                Y[k & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k & 15], __builtin_bit_cast(bf16x8, Hf[k >> 3]), Y[k & 7], 0, 0, 0);
                if (k + 8 < 32) A[(k + 8) & 15] = frag2(w2, k + 8);
            }
            if constexpr (!LAST) {
                if ((k & 1) == 0) relu_pair(S, Hn, k >> 1);
                else if (TRAIN) mask_pair(Hn, word, k >> 1);
                if (TRAIN && (k & 7) == 7)
                    *reinterpret_cast<u32x4*>(hst + hwr + ((((k >> 3) * 2 + h) ^ (r & 7)) << 4)) = Hn[k >> 3];
            } else if (TRAIN) {
                if (k >= 4 && k < 8) Hout[k - 4] = *reinterpret_cast<const u32x4*>(hst + hrd + (k - 4) * 1024);
                if (k >= 16 && k < 20) __builtin_amdgcn_raw_buffer_store_b128(Hout[k - 16], rsh, hoff[k - 16], (i - 1) * (FHC * 2), 0);
            }
            FFN_STEP();
        }
        if constexpr (!LAST) {
            if (TRAIN) __builtin_amdgcn_raw_buffer_store_b32(word, rsb, boff, i * (2 * a.Mp * 4), 0);
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) Hf[sg] = Hn[sg];
        }
    };
    {
        if (NC > 1) stage_w1(1, 1);
        stage_w2(0, 0);
        asm volatile("" ::: "memory");
        body(0, std::true_type{}, std::false_type{});
        FFN_WAIT_STAGE(1);
    }
    for (int i = 1; i < NC; ++i) {
        if (!(a.dbg & 2)) {
            if (i + 1 < NC) stage_w1((i + 1) & 1, i + 1);
            stage_w2(i & 1, i);
        }
        asm volatile("" ::: "memory");
        body(i, std::false_type{}, std::false_type{});
        FFN_WAIT_STAGE(5);
    }
    body(NC, std::false_type{}, std::true_type{});
#undef FFN_WAIT_STAGE
#undef FFN_STEP

    if (a.dbg & 1) {
        float acc = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc += Y[t][j];
        if (acc == 123.456f) a.y32[0] = acc;
        return;
    }
    // ---- epilogue: v = dropout(Y + b2) + x; LayerNorm over the row that lanes (r, 0) and (r, 1) hold together ---------------------------
    const asr_dropout_t drop = drop_resolve(a.drop);
    const int b = mc / a.L, tpos = mc - b * a.L;
    const uint32_t sub = drop.thr16 ? drop_subkey(drop, (uint32_t)b) : 0u;
    const float sc = drop_scale(drop);
    const float* xres = a.x32 + (int64_t)mc * FD + 4 * h;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * t + 8 * g;          // + 4 h
            const f32x4 bias = *reinterpret_cast<const f32x4*>(a.b2 + d0 + 4 * h);
            f32x4 v = {Y[t][4 * g] + bias[0], Y[t][4 * g + 1] + bias[1], Y[t][4 * g + 2] + bias[2], Y[t][4 * g + 3] + bias[3]};
            if (drop.thr16) v = drop4(drop, sub, (uint32_t)tpos, FD >> 1, (uint32_t)(d0 + 4 * h), v, sc);
            v += *reinterpret_cast<const f32x4*>(xres + d0);
            if (a.s_out && valid) *reinterpret_cast<f32x4*>(a.s_out + (int64_t)m * FD + d0 + 4 * h) = v;
            sum += (v[0] + v[1]) + (v[2] + v[3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) Y[t][4 * g + j] = v[j];
        }
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum * (1.f / FD);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float dlt = Y[t][j] - mean;
            q += dlt * dlt;
        }
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.f / FD) + a.eps);
    if (valid && h == 0) {
        if (a.mean) a.mean[m] = mean;
        if (a.rstd) a.rstd[m] = rstd;
    }
    const bool keep = a.row_len ? (tpos < a.row_len[b]) : true;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * t + 8 * g + 4 * h;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + d0), bt = *reinterpret_cast<const f32x4*>(a.beta + d0);
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = keep ? (Y[t][4 * g + j] - mean) * rstd * gm[j] + bt[j] : 0.f;
            if (valid) {
                *reinterpret_cast<f32x4*>(a.y32 + (int64_t)m * FD + d0) = o;
                if (a.y16) {
                    const bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                    *reinterpret_cast<bf16x4*>(a.y16 + (int64_t)m * FD + d0) = ob;
                }
            }
        }
    }
}


// ---- data gradient: dH^T = (W2c^T . ds^T) * mask, dX^T += W1c^T . dH^T -----------------------------------------------------------
// Same decomposition, same (lane, register) <-> (token, hidden unit) map as the forward, so a lane reads back exactly the mask words it
// wrote.  Both products now sum over the index the weights are NOT contiguous in (W2 [256][d_ff] over its rows, W1 [d_ff][256] over its
// rows), so the weight chunks are staged row-major as stored and every A fragment is two ds_read_b64_tr_b16 (4 k-rows x 16 columns per
// 16-lane group, delivered column-major).  The column quads a group's four address lanes point at are free: quads 1 and 2 are swapped
// for the first product, which deals the hidden units to MFMA rows in the forward's bit-2/3-swapped order.  Image swizzles (16-byte
// slot of row r holds chunk slot ^ f(r)) are chosen so that the 32 lanes of a half read 32 distinct 8-byte bank pairs:
//   W2 chunk [256 d][64 hid], 128-byte rows:  f(d) = ((d >> 1) & 1) << 2         W1 chunk [64 hid][256 d], 512-byte rows:  f(u) = (u & 3) << 2
struct FfnBwdArgs {
    const bf16_t* ds16;
    const float* ds32;
    const bf16_t* w1;
    const bf16_t* w2;
    const uint32_t* bits;
    bf16_t* dhid;
    float* dx;
    int M, dff, Mp;
};

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p, int second_off) {
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + second_off));
    const u32x2 a = __builtin_bit_cast(u32x2, v0), b = __builtin_bit_cast(u32x2, v1);
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}

__global__ __launch_bounds__(256, 1) void ffn_bwd_kernel(const FfnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * W1BUF + 2 * W2BUF + HST_BYTES];
    unsigned char* const w2s = smem;                    // first product's weights here
    unsigned char* const w1s = smem + 2 * W2BUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const hst = smem + 2 * W1BUF + 2 * W2BUF + wave * 4096;
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * FBM + wave * 32 + r;
    const bool valid = m < a.M;
    const int mc = valid ? m : a.M - 1;
    const int dff = a.dff, NC = dff / FHC;

    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w1), 0, dff * FD * 2, 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w2), 0, dff * FD * 2, 0x00020000);
    unsigned off1[8], off2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = wave * 8 + k;
        {
            const int u = 2 * p + (lane >> 5), pc = lane & 31;
            off1[k] = (unsigned)(u * FD * 2 + ((pc ^ ((u & 3) << 2)) << 4));
        }
        {
            const int d = 8 * p + (lane >> 3), pc = lane & 7;
            off2[k] = (unsigned)(d * dff * 2 + ((pc ^ (((d >> 1) & 1) << 2)) << 4));
        }
    }
    auto stage_w1 = [&](int buf, int chunk) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_void*)(w1s + buf * W1BUF + (wave * 8 + k) * 1024), 16, off1[k],
                                                     chunk * (FHC * FD * 2), 0, 0);
    };
    auto stage_w2 = [&](int buf, int chunk) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (lds_void*)(w2s + buf * W2BUF + (wave * 8 + k) * 1024), 16, off2[k],
                                                     chunk * (FHC * 2), 0, 0);
    };

    // transposed-read addresses: within a 16-lane group, lane 4q + p supplies block row q, column quad p (first product: quad p')
    const int q = (lane & 15) >> 2, pq = lane & 3, g1 = (lane >> 4) & 1;
    const int pp = ((pq & 1) << 1) | (pq >> 1);
    unsigned a1[2], a2[4];
    {
        const unsigned l0 = (unsigned)((8 * h + q) * 128 + (2 * g1 + (pp >> 1)) * 16 + (pp & 1) * 8);
        a1[0] = l0 + (unsigned)((q >> 1) * 64);
        a1[1] = l0 + (unsigned)((1 - (q >> 1)) * 64);
#pragma unroll
        for (int v = 0; v < 4; ++v) a2[v] = (unsigned)((8 * h + q) * 512 + (4 * (v ^ q) + 2 * g1 + (pq >> 1)) * 16 + (pq & 1) * 8);
    }

    bf16x8 db[16];
    {
        const bf16_t* xr = a.ds16 + (int64_t)mc * FD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) db[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
    }
    stage_w2(0, 0);
    f32x16 Y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) Y[t][j] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    u32x4 Hf[4];
    const auto rsh = __builtin_amdgcn_make_buffer_rsrc(a.dhid, 0, (int)((int64_t)a.M * dff * 2), 0x00020000);
    const auto rsb = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(a.bits), 0, (int)((int64_t)NC * 2 * a.Mp * 4), 0x00020000);
    const unsigned boff = ((unsigned)h * a.Mp + mc) * 4u;
    const unsigned hwr = (unsigned)(r * 128), hrd = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));      // as in the forward
    unsigned hoff[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int mt = blockIdx.x * FBM + wave * 32 + 8 * ps + (lane >> 3);
        hoff[ps] = mt < a.M ? (unsigned)mt * (unsigned)dff * 2u + 16u * (lane & 7) : 0x80000000u;
    }

    bf16x8 A[16];
    auto frag1 = [&](const unsigned char* w2, int k) {      // first product, MFMA k: k-step k >> 1 (16 rows of the image), row tile k & 1
        return tr_pair(w2 + a1[k & 1] + (k >> 1) * 2048, 512);
    };
    auto frag2 = [&](const unsigned char* w1, int k) {      // second product, MFMA k: k-step k >> 3 (16 rows), row tile k & 7 of dX^T
        return tr_pair(w1 + a2[k & 3] + (k >> 3) * 8192 + ((k & 7) >> 2) * 256, 2048);
    };
    // element e = 16 t + j of a lane's chunk: mask bit e >> 1 (e even) / 16 + (e >> 1) (e odd)
    auto mask_pair = [&](const f32x16 (&S)[2], u32x4 (&Hn)[4], uint32_t word, int P) {
        const int t = P >> 3, p = P & 7;
        const float lo = drop_and(S[t][2 * p], word, P), hi = drop_and(S[t][2 * p + 1], word, 16 + P);
        uint32_t pk;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(lo), "v"(hi));
        Hn[2 * t + (p >> 2)][p & 3] = pk;
    };
#define FFN_STEP() __builtin_amdgcn_sched_barrier(0)
#define FFN_WAIT_STAGE(NST)                                                           \
    do {                                                                              \
        asm volatile("s_waitcnt vmcnt(" #NST ") lgkmcnt(0)" ::: "memory");            \
        __builtin_amdgcn_s_barrier();                                                 \
        asm volatile("" ::: "memory");                                                \
    } while (0)

    auto body = [&](int i, uint32_t word, auto first_c, auto last_c) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        const unsigned char* w2 = w2s + (i & 1) * W2BUF;
        const unsigned char* w1 = w1s + ((i - 1) & 1) * W1BUF;
        f32x16 S[2];
        u32x4 Hn[4], Hout[4];
        if constexpr (!LAST) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) S[t][j] = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = frag1(w2, k);
            FFN_STEP();
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                S[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k & 15], db[k >> 1], S[k & 1], 0, 0, 0);
                if (k + 8 < 32) A[(k + 8) & 15] = frag1(w2, k + 8);
                else if (!FIRST) A[(k + 8) & 15] = frag2(w1, k + 8 - 32);
                if (!FIRST) {
                    if (k >= 4 && k < 8) Hout[k - 4] = *reinterpret_cast<const u32x4*>(hst + hrd + (k - 4) * 1024);
                    if (k >= 16 && k < 20) __builtin_amdgcn_raw_buffer_store_b128(Hout[k - 16], rsh, hoff[k - 16], (i - 1) * (FHC * 2), 0);
                }
                FFN_STEP();
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = frag2(w1, k);
            FFN_STEP();
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if constexpr (!FIRST) {
                Y[k & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k & 15], __builtin_bit_cast(bf16x8, Hf[k >> 3]), Y[k & 7], 0, 0, 0);
                if (k + 8 < 32) A[(k + 8) & 15] = frag2(w1, k + 8);
            }
            if constexpr (!LAST) {
                if ((k & 1) == 0) mask_pair(S, Hn, word, k >> 1);
                if ((k & 7) == 7) *reinterpret_cast<u32x4*>(hst + hwr + ((((k >> 3) * 2 + h) ^ (r & 7)) << 4)) = Hn[k >> 3];
            } else {
                if (k >= 4 && k < 8) Hout[k - 4] = *reinterpret_cast<const u32x4*>(hst + hrd + (k - 4) * 1024);
                if (k >= 16 && k < 20) __builtin_amdgcn_raw_buffer_store_b128(Hout[k - 16], rsh, hoff[k - 16], (i - 1) * (FHC * 2), 0);
            }
            FFN_STEP();
        }
        if constexpr (!LAST) {
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) Hf[sg] = Hn[sg];
        }
    };
    {
        const uint32_t word = __builtin_amdgcn_raw_buffer_load_b32(rsb, boff, 0, 0);
        if (NC > 1) stage_w2(1, 1);
        stage_w1(0, 0);
        asm volatile("" ::: "memory");
        body(0, word, std::true_type{}, std::false_type{});
        FFN_WAIT_STAGE(0);
    }
    for (int i = 1; i < NC; ++i) {
        // this chunk's mask word: issued BEFORE the LDS-DMA so that its wait (at the first use, half an iteration later) is a
        // counted one that leaves the DMA in flight
        const uint32_t word = __builtin_amdgcn_raw_buffer_load_b32(rsb, boff, i * (2 * a.Mp * 4), 0);
        if (i + 1 < NC) stage_w2((i + 1) & 1, i + 1);
        stage_w1(i & 1, i);
        asm volatile("" ::: "memory");
        body(i, word, std::false_type{}, std::false_type{});
        FFN_WAIT_STAGE(4);
    }
    body(NC, 0u, std::false_type{}, std::true_type{});
#undef FFN_WAIT_STAGE
#undef FFN_STEP

    // ---- epilogue: dx = dX + ds32 ------------------------------------------------------------------------------------------------
    const float* res = a.ds32 + (int64_t)mc * FD + 4 * h;
    const auto rsx = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, (int)((int64_t)a.M * FD * 4), 0x00020000);
    const unsigned xoff = valid ? (unsigned)m * (FD * 4u) + 16u * h : 0x80000000u;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d0 = 32 * t + 8 * g;
            const f32x4 rv = *reinterpret_cast<const f32x4*>(res + d0);
            const f32x4 v = {Y[t][4 * g] + rv[0], Y[t][4 * g + 1] + rv[1], Y[t][4 * g + 2] + rv[2], Y[t][4 * g + 3] + rv[3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsx, xoff + d0 * 4, 0, 0);
        }
    }
}

}  // namespace

extern "C" int64_t asr_ffn_bits_words(int M, int d_ff) { return (int64_t)(d_ff / FHC) * 2 * ((M + FBM - 1) / FBM * FBM); }

extern "C" int asr_ffn_fwd(void* stream, const void* x16, const float* x32, const void* w1, const float* b1, const void* w2, const float* b2,
                           const float* gamma, const float* beta, const int32_t* row_len, void* hid_out, void* bits_out, float* s_out,
                           float* y32, void* y16, float* mean_out, float* rstd_out, int B, int L, int d_model, int d_ff, float eps,
                           asr_dropout_t drop_x) {
    const int64_t M64 = (int64_t)B * L;
    ASR_REQUIRE(d_model == FD, -1, "asr_ffn_fwd: d_model = %d (the fused sub-layer is built for 256)", d_model);
    ASR_REQUIRE(d_ff >= FHC && d_ff % FHC == 0 && d_ff <= FFN_MAX_DFF, -1, "asr_ffn_fwd: d_ff = %d (a multiple of 64 up to %d)", d_ff, FFN_MAX_DFF);
    ASR_REQUIRE(M64 > 0 && M64 * d_ff * 2 < (1ll << 31), -1, "asr_ffn_fwd: B * L out of range");
    ASR_REQUIRE(x16 && x32 && w1 && b1 && w2 && b2 && gamma && beta && y32, -1, "asr_ffn_fwd: null argument");
    ASR_REQUIRE((hid_out == nullptr) == (bits_out == nullptr), -1, "asr_ffn_fwd: hid_out and bits_out come together (training) or not at all");
    ASR_REQUIRE(asr_aligned(x16, 16) && asr_aligned(x32, 16) && asr_aligned(w1, 16) && asr_aligned(w2, 16) && asr_aligned(y32, 16) &&
                    asr_aligned(hid_out, 16) && asr_aligned(s_out, 16) && asr_aligned(y16, 8) && asr_aligned(b1, 16) && asr_aligned(b2, 16) &&
                    asr_aligned(gamma, 16) && asr_aligned(beta, 16), -1, "asr_ffn_fwd: 16-byte aligned buffers required");
    const int M = (int)M64;
    FfnFwdArgs a{(const bf16_t*)x16, x32, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, gamma, beta, row_len, (bf16_t*)hid_out,
                 (uint32_t*)bits_out, s_out, y32, (bf16_t*)y16, mean_out, rstd_out, M, L, d_ff, (M + FBM - 1) / FBM * FBM, eps, drop_x,
                 getenv("ASR_AMD_FFN_DBG") ? atoi(getenv("ASR_AMD_FFN_DBG")) : 0};
    const dim3 grid((M + FBM - 1) / FBM), block(256);
    if (hid_out)
        hipLaunchKernelGGL(ffn_fwd_kernel<true>, grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(ffn_fwd_kernel<false>, grid, block, 0, (hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_ffn_fwd");
    return 0;
}

extern "C" int asr_ffn_bwd(void* stream, const void* ds16, const float* ds32, const void* w1, const void* w2, const void* bits,
                           void* dhid_out, float* dx_out, int M, int d_model, int d_ff) {
    ASR_REQUIRE(d_model == FD, -1, "asr_ffn_bwd: d_model = %d (the fused sub-layer is built for 256)", d_model);
    ASR_REQUIRE(d_ff >= FHC && d_ff % FHC == 0 && d_ff <= FFN_MAX_DFF, -1, "asr_ffn_bwd: d_ff = %d (a multiple of 64 up to %d)", d_ff, FFN_MAX_DFF);
    ASR_REQUIRE(M > 0 && (int64_t)M * d_ff * 2 < (1ll << 31), -1, "asr_ffn_bwd: M out of range");
    ASR_REQUIRE(ds16 && ds32 && w1 && w2 && bits && dhid_out && dx_out, -1, "asr_ffn_bwd: null argument");
    ASR_REQUIRE(asr_aligned(ds16, 16) && asr_aligned(ds32, 16) && asr_aligned(w1, 16) && asr_aligned(w2, 16) && asr_aligned(dhid_out, 16) &&
                    asr_aligned(dx_out, 16), -1, "asr_ffn_bwd: 16-byte aligned buffers required");
    FfnBwdArgs a{(const bf16_t*)ds16, ds32, (const bf16_t*)w1, (const bf16_t*)w2, (const uint32_t*)bits, (bf16_t*)dhid_out, dx_out, M, d_ff,
                 (M + FBM - 1) / FBM * FBM};
    hipLaunchKernelGGL(ffn_bwd_kernel, dim3((M + FBM - 1) / FBM), dim3(256), 0, (hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_ffn_bwd");
    return 0;
}
