// Vocabulary projection of the training step's CTC branch (src/transformer/transformer.py:119,148 `ctc_fc`, a bias-free Linear
// d_model -> V, + the `F.log_softmax` and the label gather of loss.py:41-43) - everything the CTC forward needs from the logits, taken
// while they are still fp32 accumulators on their way out:
//   logits[m, v] = x[m, :] . W[v, :]                  stored once, as IEEE fp16 (the CTC gradient pass reads them)
//   lse[m]       = log sum_v exp(logits[m, v])
//   lp_ext[m, s] = (logits[m, label_s] - lse[m]) log2 e   for the 2 U + 1 states of the frame's utterance (ctc.hip's table rows)
// The CTC forward then never touches the 542 MB of logits (B 32 x L 1000 x V 4234) at all: it is the alpha / beta recursion on the table
// (asr_ctc_loss_fwd_table).
//
// Same decomposition as the fused feed-forward's first product (ffn.hip): a workgroup owns 128 frames, each wave 32 of them as the
// MFMA's B operand held in registers (x^T, K = d_model = 256); the weight streams through LDS in chunks of 64 vocabulary rows
// (LDS-DMA, double-buffered, every workgroup reads the same 2 MB from its XCD's L2); S^T[64 vocab x 32 frames] per chunk and wave on
// v_mfma_f32_32x32x16_bf16 with the frame on the LANE - so a frame's running (max, sum exp) is in-register work on the lane's own
// 32 values of the chunk, the two lane halves meet once at the very end.  The chunk's logits leave through a per-wave LDS tile
// ([32 frames][64 vocab] f32) read back 16 lanes per frame: a store instruction covers 4 frames x 256 contiguous bytes.
// The statistics / store work of chunk i - 1 sits between the MFMAs of chunk i (pinned step by step, one wave per SIMD).
#include <type_traits>

#include "asr_common.h"

namespace {

constexpr int VBM = 128;             // frames per workgroup
constexpr int VC = 64;               // vocabulary rows per chunk
constexpr int VD = 256;              // d_model
constexpr int VWBUF = VC * VD * 2;   // 32 KiB: [64 rows][256 k] bf16
constexpr int VTILE = 32 * VC * 4;   // 8 KiB per wave: [32 frames][64 vocab] f32

__device__ __forceinline__ int vswap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

struct VocabArgs {
    const bf16_t* x16;
    const bf16_t* w;
    float* lse;
    int M, V;
    int64_t ldl;
    _Float16* logits16;  // [M, ldl] IEEE half (ldl % 8 == 0)
    float* lp_ext;       // [M, 128]: (x[label] - lse) log2 e of the extended label sequence, -inf beyond 2 U_b + 1 (ctc.hip's table)
    const int64_t* targets;      // [B, Umax]
    int B, L, Umax, blank;
};
constexpr int GLC = 64;              // gathered logits per frame in LDS - columns 0 .. Umax - 1 the labels, column 63 the blank
constexpr int VGLAB = VBM * GLC * 4; // 32 KiB

// The logits leave as IEEE fp16 (their only later reader is the CTC gradient pass, which forms
// exp(logit - lse): 11 significand bits keep that within 0.2 % at |logit| <= 8; bf16's 8 bits would put 1.6 % on the largest probabilities) and the CTC forward's table rows are produced HERE - every logit passes through the per-wave LDS tile in fp32 on
// its way out, so the ~52 an utterance's extended label sequence needs are picked up there instead of being gathered from the 542 MB
// in a launch of its own: per chunk a lane reads the tile entries of the labels that fall into the chunk (the utterance's labels
// sorted by vocabulary index at kernel entry, one running pointer per lane) into a [128 frames][64] LDS table, and once the row's
// log-sum-exp is known the workgroup writes its 128 table rows, 512 contiguous bytes each.  Same table bits as ctc.hip's gather.
__global__ __launch_bounds__(256, 1) void vocab_proj_ctc_kernel(const VocabArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * VWBUF + 4 * VTILE + VGLAB + 2 * 64 * 4 + 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const tile = smem + 2 * VWBUF + wave * VTILE;
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * VBM + wave * 32 + r;
    const int mc = m < a.M ? m : a.M - 1;
    const int V = a.V, NCH = (V + VC - 1) / VC;
    // ---- the (at most two) utterances this block's frames belong to, their labels sorted by vocabulary index -------------
    float* const glab = reinterpret_cast<float*>(smem + 2 * VWBUF + 4 * VTILE);                    // [128][GLC]
    int* const skey = reinterpret_cast<int*>(smem + 2 * VWBUF + 4 * VTILE + VGLAB);               // [2][64]: vocab id << 8 | column
    int* const snum = skey + 128;                                                                   // [2]: entries (U_b + 1)
    int ub = 0, ptr = 0, nent = 0;
    {
        const int b0 = (blockIdx.x * VBM) / a.L;
        if (tid < 128) {
            const int u = tid >> 6, t = tid & 63, b = b0 + u;
            // entry t < U_b: label t (the first U_b = #nonzero entries of the row, loss.py:40); entry U_b: the blank
            int nl = 0, lab = 0;
            if (b < a.B) {
                const int64_t* tgr = a.targets + (int64_t)b * a.Umax;
                const int mine = t < a.Umax ? (tgr[t] != 0 ? 1 : 0) : 0;
                nl = __builtin_popcountll(__ballot(mine));
                lab = t < nl ? (int)tgr[t] : a.blank;
                lab = min(max(lab, 0), V - 1);
            }
            const int n = b < a.B ? nl + 1 : 0;
            const int key = (lab << 8) | (t < nl ? t : (GLC - 1));
            // rank sort inside the wave (keys are unique per live entry: the column is part of the key)
            int rank = 0;
            for (int o = 0; o < n; ++o) {
                const int ko = __shfl(key, o, 64);
                rank += (ko < key) ? 1 : 0;
            }
            if (t < n) skey[u * 64 + rank] = key;
            if (t == 0) snum[u] = n;
        }
        __syncthreads();
        ub = (mc / a.L) - b0;
        nent = snum[ub];
    }

    // weight rows past V read as zeros (the descriptor's range check); their logits are masked out of the statistics and never stored
    const u32x4 rsw = rsrc_words(a.w, (unsigned)((int64_t)V * VD * 2));
    unsigned off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {      // piece p: chunk rows 2p, 2p + 1; LDS slot pc of row u holds 16-byte chunk (pc & 16) | ((pc ^ u) & 15)
        const int p = wave * 8 + k, u = 2 * p + (lane >> 5), pc = lane & 31;
        off[k] = (unsigned)(u * VD * 2 + (((pc & 16) | ((pc ^ u) & 15)) << 4));
    }
    auto dma = [&](int buf, int chunk, int j) {
        dma16_asm(rsw, off[j], (unsigned)chunk * (VC * VD * 2), lds_addr_of(smem + buf * VWBUF + (wave * 8 + j) * 1024));
    };
    const int u15 = vswap23(r) & 15;
    unsigned a1[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a1[kk] = (unsigned)(vswap23(r) * 512 + (((2 * kk + h) ^ u15) << 4));

    bf16x8 xb[16];
    {
        const bf16_t* xr = a.x16 + (int64_t)mc * VD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dma(0, 0, j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // the logits tile of a chunk in LDS: frame r's 64 values, 16-byte piece q (vocab 4q .. 4q + 3 of the chunk) in slot q ^ (r & 15);
    // written by the lane that owns frame r (pieces 2 (4t + 2s + h') ...), read back by 16 lanes per frame
    constexpr unsigned EB = 2u;      // bytes per stored logit
    const auto rsl = __builtin_amdgcn_make_buffer_rsrc(a.logits16, 0, (int)((int64_t)a.M * a.ldl * 2), 0x00020000);
    unsigned soff[8];      // read-back pass ps: frame 4 ps + (lane >> 4), piece lane & 15
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        const int mt = blockIdx.x * VBM + wave * 32 + 4 * ps + (lane >> 4);
        soff[ps] = mt < a.M ? (unsigned)((int64_t)mt * a.ldl * EB) + 4u * EB * (lane & 15) : 0x80000000u;
    }
    const unsigned trow = (unsigned)((lane >> 4) * 256), tx0 = (unsigned)((lane & 15) ^ (lane >> 4));      // pass ps: slot tx0 ^ (4 ps & 15)

    float mx = -INFINITY, sm = 0.f;      // this lane's running (max, sum exp(v - max)) over its half of the vocabulary
    bf16x8 A[16];
    f32x16 S[2], Sp[2];
    u32x4 outv[8];
#define VSTEP() __builtin_amdgcn_sched_barrier(0)
    // statistics / store work of one finished chunk (values in Sp), dealt over 32 steps; `chunk` = its index
    //   steps  0.. 7   the chunk's maximum (v_max3), masking the rows past V;  one tile write per step
    //   step   8       move the running maximum, rescale the running sum
    //   steps  8..23   sum += exp(v - max) two values per step;  steps 16..23 the tile's read-back, 24..31 its stores
    float cm = -INFINITY, nmx = 0.f, nm2 = 0.f;
    auto stat_step = [&](int chunk, int k) {
        const int nvalid = V - chunk * VC;      // >= 64 except in the last chunk
        if (k < 8) {
            // values 4k .. 4k + 3 of the lane: tile t = k >> 2, registers 4 (k & 3) ..; vocab (in chunk) 32 t + 16 (j >> 3) + 8 h + (j & 7)
            const int t = k >> 2, j0 = 4 * (k & 3);
            float v4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + e;
                const int vloc = 32 * t + 16 * (j >> 3) + 8 * h + (j & 7);
                v4[e] = vloc < nvalid ? Sp[t][j] : 0.f;          // what the pad columns of the row (V .. ldl - 1) receive
                Sp[t][j] = vloc < nvalid ? Sp[t][j] : -INFINITY;  // what the statistics see
            }
            cm = fmaxf(cm, fmaxf(fmaxf(Sp[t][j0], Sp[t][j0 + 1]), fmaxf(Sp[t][j0 + 2], Sp[t][j0 + 3])));
            // piece index within the frame's 64 values: vocab 32 t + 16 s + 8 h + 4 e' -> q = 8 t + 4 s + 2 h + e'  (s = j0 >> 3, e' = (j0 >> 2) & 1)
            const int q = 8 * t + 4 * (j0 >> 3) + 2 * h + ((j0 >> 2) & 1);
            *reinterpret_cast<f32x4*>(tile + r * 256 + ((q ^ (r & 15)) << 4)) = f32x4{v4[0], v4[1], v4[2], v4[3]};
        }
        if (k == 8) {
            const float nm = fmaxf(mx, cm);
            const float ref = nm == -INFINITY ? 0.f : nm;
            sm *= __builtin_amdgcn_exp2f((mx - ref) * 1.4426950408889634f);      // (mx = -inf: the sum is still 0)
            mx = nm;
            nmx = ref;
            nm2 = -ref * 1.4426950408889634f;
            cm = -INFINITY;
        }
        if (k >= 8 && k < 24) {
            const int e0 = 2 * (k - 8), t = e0 >> 4, j = e0 & 15;
            sm += __builtin_amdgcn_exp2f(__builtin_fmaf(Sp[t][j], 1.4426950408889634f, nm2)) +
                  __builtin_amdgcn_exp2f(__builtin_fmaf(Sp[t][j + 1], 1.4426950408889634f, nm2));
        }
        {
            if (k == 9) {
                // the chunk's tile is complete (steps 0..7, this wave's own LDS writes): pick up the labels that live in it.  The two
                // lane halves of a frame walk the same sorted list and take alternate entries.
                const int hi = (chunk + 1) * VC;
                const int* sk = skey + ub * 64;
                while (ptr < nent) {
                    const int key = sk[ptr];
                    const int lab = key >> 8;
                    if (lab >= hi) break;
                    if ((ptr & 1) == h) {
                        const int vloc = lab - chunk * VC, q = vloc >> 2;
                        const float v = *reinterpret_cast<const float*>(tile + r * 256 + ((q ^ (r & 15)) << 4) + (vloc & 3) * 4);
                        glab[(wave * 32 + r) * GLC + (key & 255)] = v;
                    }
                    ++ptr;
                }
            }
        }
        if (k >= 16 && k < 24)
            outv[k - 16] = *reinterpret_cast<const u32x4*>(tile + (k - 16) * 1024 + trow + ((tx0 ^ (unsigned)((4 * (k - 16)) & 15)) << 4));
        if (k >= 24) {
            // columns at or past the row stride do not exist (the last chunk overhangs it): the range check drops them
            const unsigned col = (unsigned)chunk * VC + 4u * (lane & 15);
            const unsigned o = col < (unsigned)a.ldl ? soff[k - 24] : 0x80000000u;
            const f32x4 f = __builtin_bit_cast(f32x4, outv[k - 24]);
            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
            const f16x4 b4 = {(_Float16)f[0], (_Float16)f[1], (_Float16)f[2], (_Float16)f[3]};      // round to nearest even
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, b4), rsl, o, chunk * (VC * 2), 0);
        }
        (void)nmx;
    };
    auto body = [&](int i, auto first_c, auto last_c) {
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        const unsigned char* w = smem + (i & 1) * VWBUF;
        const int nxt = i + 1 < NCH ? i + 1 : NCH - 1;
        if constexpr (!LAST) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) S[t][j] = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = *reinterpret_cast<const bf16x8*>(w + a1[(k >> 1) & 7] + (k & 1) * 16384 + ((k >> 1) >> 3) * 256);
            VSTEP();
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if constexpr (!LAST) {
                S[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k & 15], xb[k >> 1], S[k & 1], 0, 0, 0);
                if (k + 8 < 32) {
                    const int k8 = k + 8, ks = k8 >> 1;
                    A[k8 & 15] = *reinterpret_cast<const bf16x8*>(w + a1[ks & 7] + (k8 & 1) * 16384 + (ks >> 3) * 256);
                }
                if (k < 8) dma((i + 1) & 1, nxt, k);
            }
            if constexpr (!FIRST) stat_step(i - 1, k);
            VSTEP();
        }
        if constexpr (!LAST) {
#pragma unroll
            for (int t = 0; t < 2; ++t) Sp[t] = S[t];
        }
    };
#define VWAIT(NST)                                                                   \
    do {                                                                             \
        asm volatile("s_waitcnt vmcnt(" #NST ") lgkmcnt(0)" ::: "memory");           \
        __builtin_amdgcn_s_barrier();                                                \
        asm volatile("" ::: "memory");                                               \
    } while (0)
    body(0, std::true_type{}, std::false_type{});
    VWAIT(0);
    for (int i = 1; i < NCH; ++i) {
        body(i, std::false_type{}, std::false_type{});
        VWAIT(8);      // this iteration's 8 DMA pieces are older than its 8 stores
    }
    body(NCH, std::false_type{}, std::true_type{});
#undef VWAIT
#undef VSTEP
    // the two lane halves hold disjoint parts of the frame's vocabulary
    const float mo = __shfl_xor(mx, 32, 64), so = __shfl_xor(sm, 32, 64);
    const float mm = fmaxf(mx, mo);
    // (one explicit fma per lane half, the same expression in both: left to the compiler, a * b + c * d contracts around either product)
    const float tot = h == 0 ? __builtin_fmaf(sm, __expf(mx - mm), so * __expf(mo - mm)) : __builtin_fmaf(so, __expf(mo - mm), sm * __expf(mx - mm));
    const float lse = mm + logf(tot);
    if (h == 0 && m < a.M) a.lse[m] = lse;
    {
        // table rows of this wave's 32 frames: state pair (2 lane, 2 lane + 1) = (blank, label `lane`) per lane, 512 bytes per frame
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (this wave's own glab writes)
        const int Sb = 2 * (nent - 1) + 1;                       // 2 U_b + 1 live states of the LANE's utterance (uniform per frame)
        constexpr float L2E = 1.4426950408889634f;
#pragma unroll 4
        for (int f = 0; f < 32; ++f) {
            const int mf = blockIdx.x * VBM + wave * 32 + f;
            if (mf >= a.M) break;
            const float lf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lse), f));
            const int sbf = __builtin_amdgcn_readlane(Sb, f);
            const float* g = glab + (wave * 32 + f) * GLC;
            const float vb = g[GLC - 1], vl = g[lane];
            const f32x2 o = {(2 * lane < sbf) ? (vb - lf) * L2E : -INFINITY, (2 * lane + 1 < sbf) ? (vl - lf) * L2E : -INFINITY};
            *reinterpret_cast<f32x2*>(a.lp_ext + (int64_t)mf * 128 + 2 * lane) = o;
        }
    }
}

}  // namespace

extern "C" int asr_vocab_proj_ctc(void* stream, const void* x16, const void* w16, void* logits16, int64_t ldl, float* lse, float* lp_ext,
                                  const int64_t* targets, int B, int L, int V, int Umax, int blank, int d_model) {
    ASR_REQUIRE(d_model == VD, ASR_ERR_UNSUPPORTED, "asr_vocab_proj_ctc: d_model = %d (built for 256)", d_model);
    ASR_REQUIRE(x16 && w16 && logits16 && lse && lp_ext && targets && B > 0 && L > 0 && V > 0 && ldl >= V && ldl % 8 == 0, ASR_ERR_ARG,
                "asr_vocab_proj_ctc: bad arguments");
    ASR_REQUIRE(L >= VBM, ASR_ERR_UNSUPPORTED, "asr_vocab_proj_ctc: L = %d (a 128-frame block must not span more than two utterances)", L);
    ASR_REQUIRE(Umax >= 1 && Umax + 1 <= 64, ASR_ERR_UNSUPPORTED, "asr_vocab_proj_ctc: Umax = %d (one state pair per lane: U + 1 <= 64)", Umax);
    ASR_REQUIRE(blank >= 0 && blank < V && V < (1 << 22), ASR_ERR_ARG, "asr_vocab_proj_ctc: blank / V out of range");
    const int64_t M64 = (int64_t)B * L;
    ASR_REQUIRE(M64 * ldl * 2 < (1ll << 31) && M64 < (1ll << 24), ASR_ERR_UNSUPPORTED, "asr_vocab_proj_ctc: B * L * ldl out of range");
    ASR_REQUIRE(asr_aligned(x16, 16) && asr_aligned(w16, 16) && asr_aligned(logits16, 16) && asr_aligned(lp_ext, 16), ASR_ERR_ALIGN,
                "asr_vocab_proj_ctc: 16-byte aligned buffers required");
    const int M = (int)M64;
    VocabArgs a{(const bf16_t*)x16, (const bf16_t*)w16, lse, M, V, ldl, (_Float16*)logits16, lp_ext, targets, B, L, Umax, blank};
    hipLaunchKernelGGL(vocab_proj_ctc_kernel, dim3((M + VBM - 1) / VBM), dim3(256), 0, (hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_vocab_proj_ctc");
    return 0;
}
