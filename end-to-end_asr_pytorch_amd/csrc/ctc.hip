// Fused CTC loss for gfx950: logits -> log-softmax -> alpha/beta -> loss and d(loss)/d(logits).
// Replaces F.log_softmax + F.ctc_loss at loss.py:41-43 / ctcModel/loss.py:9-11 (aten `_ctc_loss`, Graves 2006).
//
// HBM-bound design: the [B,L,V] logits are streamed exactly once forward (row log-sum-exp + gather
// of the 2U+1 extended-label log-probs into a compact [B,L,S] table) and once backward (softmax recomputed from the
// saved row lse; gradient written once).  The T-long alpha / beta recursions never touch the V axis: they run on the
// compact table, TWO WAVEFRONTS per utterance (alpha forward, beta backward, meeting in the middle - each dependent chain is T/2
// long for the loss and T/2 more for the gradient) with no barriers inside the chains: lane i owns the (blank, label) state pairs
// i*NP..i*NP+NP-1, so a time step needs exactly one neighbour value (the previous lane's last label state), fetched
// with a DPP wave shift; the table is kept in the base-2 log domain so log-sum-exp is v_exp_f32 / v_log_f32 directly;
// table rows are prefetched 8 steps ahead into registers, leaving ~a dozen dependent VALU ops per step on the chain.
// Per-label occupancies are scattered into an LDS vector indexed by vocabulary id, so repeated labels need no global
// atomics and the gradient row is produced by one coalesced stream.
#include "asr_common.h"

namespace {

__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m));
}

// compact-table row stride = 128 * NP floats, NP = state pairs per lane of the recursion kernel (1, 2, 4 or 8): every lane of the
// recursion wave owns an in-bounds 2*NP-float slice of each row, so its loads/stores need no predicate and a row is written
// by one fully coalesced wave store (512 B for NP = 1).  Columns >= 2U+1 hold -inf.
__host__ __device__ __forceinline__ int ctc_np(int Umax) { return Umax + 1 <= 64 ? 1 : (Umax + 1 <= 128 ? 2 : (Umax + 1 <= 256 ? 4 : 8)); }
__host__ __device__ __forceinline__ int ctc_row_stride(int Umax) { return 128 * ctc_np(Umax); }

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// one workgroup per (b,t) row: (max, sum-exp) over V with every load of the row issued up front, then the gather of
// the extended labels' log-probs (stored base-2: (x - lse) * log2(e)) into the compact table row (stride Sp = 2U+2).
constexpr int LSE_UNR = 6;  // float4 per thread held in registers: V <= 256*4*6 = 6144 takes the two-pass register path

constexpr int CTC_RPB = 4;   // table rows per workgroup in the flag-pipelined launch (one store drain + one counter add for all of them)

// one table row (b, t): log-sum-exp over the vocabulary, then the gather of the extended labels' base-2 log-probs
__device__ __forceinline__ void ctc_lse_row(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ targets, int L, int V,
                                            int Umax, int blank, float* __restrict__ lse_out, float* __restrict__ lp_ext, int b, int t,
                                            bool publish) {
    const int row = b * L + t;
    const float* x = logits + (int64_t)row * ldl;
    const int tid = threadIdx.x;
    float m = -INFINITY, s = 0.f;
    const int mis = (int)((reinterpret_cast<uintptr_t>(x) >> 2) & 3);
    const int peel = min((4 - mis) & 3, V);
    const int nv4 = (V - peel) >> 2;
    const int tail0 = peel + nv4 * 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + peel);
    if (nv4 <= 256 * LSE_UNR) {
        f32x4 v[LSE_UNR];
#pragma unroll
        for (int j = 0; j < LSE_UNR; ++j) {
            const int i = tid + 256 * j;
            v[j] = (i < nv4) ? x4[i] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        }
        float xe = -INFINITY;  // peel / tail element of this thread
        if (tid < peel) xe = x[tid];
        else if (tid - peel < V - tail0 && tid >= peel) xe = x[tail0 + tid - peel];
        m = xe;
#pragma unroll
        for (int j = 0; j < LSE_UNR; ++j) m = fmaxf(m, fmaxf(fmaxf(v[j][0], v[j][1]), fmaxf(v[j][2], v[j][3])));
        const float ms = (m == -INFINITY) ? 0.f : m;
        s = __expf(xe - ms);
#pragma unroll
        for (int j = 0; j < LSE_UNR; ++j)
            s += (__expf(v[j][0] - ms) + __expf(v[j][1] - ms)) + (__expf(v[j][2] - ms) + __expf(v[j][3] - ms));
    } else {
        if (tid < peel) { m = x[tid]; s = 1.f; }
        for (int i = tid; i < nv4; i += 256) {
            const f32x4 v = x4[i];
            const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
            if (m4 > m) { s *= __expf(m - m4); m = m4; }
            s += (__expf(v[0] - m) + __expf(v[1] - m)) + (__expf(v[2] - m) + __expf(v[3] - m));
        }
        if (tid < V - tail0) lse_combine(m, s, x[tail0 + tid], 1.f);
    }
    // wave then block combine
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        lse_combine(m, s, m2, s2);
    }
    __shared__ float sm[4], ss[4];
    __shared__ float lse_sh;
    if ((tid & 63) == 0) { sm[tid >> 6] = m; ss[tid >> 6] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = sm[0], S = ss[0];
        for (int w = 1; w < 4; ++w) lse_combine(M, S, sm[w], ss[w]);
        const float l = M + logf(S);
        lse_sh = l;
        lse_out[row] = l;
    }
    __syncthreads();
    const float lse = lse_sh;
    // all Umax label slots are gathered (padding slots read label 0); the recursion masks the states beyond 2*tgt_len+1
    const int Sp = ctc_row_stride(Umax), Sb = 2 * Umax + 1;
    auto gather = [&](int sidx) {
        float v = -INFINITY;
        if (sidx < Sb) {
            int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            lab = min(max(lab, 0), V - 1);
            v = (x[lab] - lse) * LOG2E;
        }
        return v;
    };
    if (publish) {   // published rows: one 8-byte write-through store per (blank, label) state pair
        for (int pr = tid; pr < Sp / 2; pr += 256) {
            const f32x2 v2 = {gather(2 * pr), gather(2 * pr + 1)};
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(lp_ext + (int64_t)row * Sp + 2 * pr),
                               __builtin_bit_cast(unsigned long long, v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        for (int sidx = tid; sidx < Sp; sidx += 256) lp_ext[(int64_t)row * Sp + sidx] = gather(sidx);
    }
}

__global__ __launch_bounds__(256) void ctc_lse_gather_kernel(const float* __restrict__ logits, int64_t ldl,
                                                             const int32_t* __restrict__ in_len, const int64_t* __restrict__ targets,
                                                             int L, int V, int Umax, int blank,
                                                             float* __restrict__ lse_out, float* __restrict__ lp_ext, int chunk, int W,
                                                             int* __restrict__ arrivals, int64_t arr_stride, int nchunks, int Bn) {
    // arrivals == NULL: one launch over all rows, blockIdx.x = b*L + t.
    // arrivals != NULL: the flag-pipelined form (asr_ctc_loss_fwd): ONE launch covers every chunk, chunk-major - chunk c holds, for
    // every utterance, the forward frames [cW, (c+1)W) up to mid and the backward frames Tb-1-k, k in [cW, (c+1)W), above mid,
    // i.e. what the two recursion wavefronts consume in their c-th piece.  A workgroup takes CTC_RPB consecutive rows of one
    // (utterance, direction, chunk), stores them write-through (sc1), drains its stores, and one lane adds the row count to
    // that arrival counter at agent scope (MI355X_MICROARCH.md § visibility).
    if (!arrivals) {
        const int b = blockIdx.x / L, t = blockIdx.x - b * L;
        if (t >= in_len[b]) return;
        ctc_lse_row(logits, ldl, targets, L, V, Umax, blank, lse_out, lp_ext, b, t, false);
        return;
    }
    const int G = W / CTC_RPB;                       // row groups per (utterance, direction, chunk); W is a multiple of CTC_RPB
    int bid = blockIdx.x;
    chunk = bid / (Bn * 2 * G);
    bid -= chunk * (Bn * 2 * G);
    const int b = bid / (2 * G), gi = bid - b * 2 * G;
    const int dirc = gi / G, g0 = (gi - dirc * G) * CTC_RPB;
    const int Tb = min(in_len[b], L), mid = Tb >> 1;
    int count = 0;
    for (int r = 0; r < CTC_RPB; ++r) {
        const int k = chunk * W + g0 + r;
        const int t = dirc == 0 ? k : Tb - 1 - k;
        const bool valid = dirc == 0 ? (t <= mid && t < Tb) : (t > mid);
        if (!valid) break;                           // frames only run out at the end of a direction
        ctc_lse_row(logits, ldl, targets, L, V, Umax, blank, lse_out, lp_ext, b, t, true);
        ++count;
    }
    if (count == 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_fetch_add(arrivals + (int64_t)b * arr_stride + dirc * nchunks + chunk, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#ifndef CTC_PF1
#define CTC_PF1 16    // table rows the NP = 1 recursion keeps in flight
#endif
constexpr float NEG_BIG = -1.0e30f;   // max(m, NEG_BIG) keeps exp2(x - m) = 0 and m + log2(0) = -inf when every input is -inf
// base-2 log-sum-exp with the largest term's exp2(0) = 1 taken for granted (0 when every input is -inf): one quarter-rate
// transcendental fewer per call on the recursion's dependent chain (v_max3 / v_med3 / v_min3 are full rate)
__device__ __forceinline__ float l2se2(float a, float b) {
    const float m = fmaxf(a, b), ms = fmaxf(m, NEG_BIG);
    const float one = (m == -INFINITY) ? 0.f : 1.f;
    return ms + __builtin_amdgcn_logf(one + __builtin_amdgcn_exp2f(fminf(a, b) - ms));
}
__device__ __forceinline__ float l2se3(float a, float b, float c) {
    const float m = __builtin_fmaxf(__builtin_fmaxf(a, b), c), ms = fmaxf(m, NEG_BIG);
    const float md = __builtin_amdgcn_fmed3f(a, b, c), mn = __builtin_fminf(__builtin_fminf(a, b), c);
    const float one = (m == -INFINITY) ? 0.f : 1.f;
    return ms + __builtin_amdgcn_logf((one + __builtin_amdgcn_exp2f(md - ms)) + __builtin_amdgcn_exp2f(mn - ms));
}
// occupancy of a state = alpha * beta / y / p(l|x) (beta includes y_t like aten's); dead states (lp = -inf) give 0
__device__ __forceinline__ float occupancy(float a2, float b2, float lp2, float nll2) {
    return (lp2 == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(a2 + b2 - lp2 + nll2);
}
__device__ __forceinline__ float dpp_from_prev_lane(float v) {  // lane i <- lane i-1 ; lane 0 <- -inf
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0xff800000, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_next_lane(float v) {  // lane i <- lane i+1 ; lane 63 <- -inf
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0xff800000, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// One direction of the recursion as a register-resident chain.  lane i owns state pairs j = i*NP + p: e = state 2j (blank),
// o = state 2j+1 (label j); all values are base-2 logs.  DIRB = false: alpha (t increasing), true: beta (t decreasing).
// OCC = false: the new state row is stored;  OCC = true: the OTHER direction's stored row is read and replaced in place by the
// occupancies exp2(alpha + beta - lp + nll2).  Row t lives at al + row_of(t) * Sp where row_of(special_t) = special_row.
template <int NP, bool DIRB, bool OCC>
__device__ __forceinline__ void ctc_chain(const float* __restrict__ lp, float* __restrict__ al, int t_first, int nsteps, float (&e)[NP],
                                          float (&o)[NP], const float (&skip_add)[NP], const float (&madd)[2 * NP], float nll2,
                                          int special_t, int special_row) {
    constexpr int PF = NP == 1 ? CTC_PF1 : (NP == 2 ? 8 : (NP == 4 ? 4 : 2));
    constexpr int Sp = 128 * NP;
    if (nsteps <= 0) return;
    auto row_of = [&](int t) { return t == special_t ? special_row : t; };
    auto load_row = [&](const float* base, int64_t row, float (&dst)[2 * NP]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(base + row * Sp + 2 * q);
            dst[2 * q] = v[0];
            dst[2 * q + 1] = v[1];
        }
    };
    auto step = [&](int t, const float (&cf)[2 * NP], const float (&ca)[2 * NP]) {
        float en[NP], on[NP];
        if (!DIRB) {
            const float left = dpp_from_prev_lane(o[NP - 1]);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float ol = (p == 0) ? left : o[p - 1];
                en[p] = l2se2(e[p], ol) + cf[2 * p];
                on[p] = l2se3(o[p], e[p], ol + skip_add[p]) + cf[2 * p + 1];
            }
        } else {
            const float re = dpp_from_next_lane(e[0]);
            const float ro = dpp_from_next_lane(o[0]);
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const float er = (p == NP - 1) ? re : e[p + 1];
                const float orr = (p == NP - 1) ? ro : o[p + 1];
                en[p] = l2se2(e[p], o[p]) + cf[2 * p];
                on[p] = l2se3(o[p], er, orr + skip_add[p]) + cf[2 * p + 1];
            }
        }
        float* dst = al + (int64_t)row_of(t) * Sp;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            e[p] = en[p];
            o[p] = on[p];
            if (!OCC)
                *reinterpret_cast<f32x2*>(dst + 2 * p) = f32x2{e[p], o[p]};
            else
                *reinterpret_cast<f32x2*>(dst + 2 * p) =
                    f32x2{occupancy(ca[2 * p], e[p], cf[2 * p], nll2), occupancy(ca[2 * p + 1], o[p], cf[2 * p + 1], nll2)};
        }
    };
    const int dir = DIRB ? -1 : 1;
    float pf[PF][2 * NP], pa[PF][2 * NP];
    auto fetch = [&](int k0) {   // steps k0..k0+PF-1, clamped to the chain (over-fetched rows are never consumed)
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int k = min(k0 + i, nsteps - 1);
            const int t = t_first + dir * k;
            load_row(lp, t, pf[i]);
            if (OCC) load_row(al, row_of(t), pa[i]);
        }
    };
    fetch(0);
    int k0 = 0;
    for (; k0 + PF <= nsteps; k0 += PF) {
        float cf[PF][2 * NP], ca[PF][2 * NP];
#pragma unroll
        for (int i = 0; i < PF; ++i)
#pragma unroll
            for (int q = 0; q < 2 * NP; ++q) { cf[i][q] = pf[i][q] + madd[q]; ca[i][q] = OCC ? pa[i][q] : 0.f; }   // madd: -inf on states >= 2U+1
        fetch(k0 + PF);
#pragma unroll
        for (int i = 0; i < PF; ++i) step(t_first + dir * (k0 + i), cf[i], ca[i]);
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        float ca[2 * NP], cf[2 * NP];
#pragma unroll
        for (int q = 0; q < 2 * NP; ++q) { cf[q] = pf[i][q] + madd[q]; ca[q] = OCC ? pa[i][q] : 0.f; }
        if (k0 + i < nsteps) step(t_first + dir * (k0 + i), cf, ca);
    }
}

// Meet-in-the-middle CTC recursion: one workgroup (2 wavefronts) per utterance.  Wave 0 runs alpha forward, wave 1 runs beta
// backward, so the T-long dependent chain is cut in half for the loss (PHASE 0: they meet at mid = T/2 and
// p(l|x) = sum_s alpha_mid(s) beta_mid(s) / y_mid(s)) and again for the gradient (PHASE 1: each wave continues over the other
// half, turning the stored rows of the opposite direction into occupancies in place).  Workspace rows per utterance: L + 2
// (rows 0..mid hold alpha, mid+1..T-1 hold beta, row L holds beta_mid, row L+1 the flag-pipelined forward's arrival counters).
// PHASE 0 can be launched in `nchunk` pieces (chunk c advances each wavefront by W frames, resuming from the row the previous
// launch stored; the last one also takes beta onto row mid and computes the loss), so that the log-sum-exp pass over the next
// frames (ctc_lse_gather_kernel, same chunk geometry) runs on another stream while this latency-bound chain works on the
// previous ones.  W >= mid + 1 with nchunk = 1 is the single-launch form.
template <int NP, int PHASE>
__global__ __launch_bounds__(128) void ctc_mitm_kernel(const float* __restrict__ lp_ext, const int32_t* __restrict__ in_len,
                                                       const int64_t* __restrict__ targets, int32_t* __restrict__ tgt_len, int L,
                                                       int Umax, float* __restrict__ alpha, float* __restrict__ nll, int chunk, int W,
                                                       int last) {
    constexpr int Sp = 128 * NP;
    __shared__ float xch[64][2 * NP];
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = lane * NP;
    const int64_t* tg = targets + (int64_t)b * Umax;
    int U;
    if (PHASE == 0) {   // loss.py:40  targets.ne(0).sum(1)
        int n = 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) n += (j0 + p < Umax && tg[j0 + p] != 0) ? 1 : 0;
        U = (int)wave_sum((float)n);
        if (threadIdx.x == 0 && last) tgt_len[b] = U;
    } else {
        U = tgt_len[b];
    }
    const int Tb = min(in_len[b], L);
    if (Tb <= 0) {
        if (PHASE == 0 && threadIdx.x == 0 && last) nll[b] = (U == 0) ? 0.f : INFINITY;
        return;
    }
    const int mid = Tb >> 1;
    float skip_f[NP], skip_b[NP], madd[2 * NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int j = j0 + p;
        skip_f[p] = ((j >= 1 && j < U) && (tg[j] != tg[j - 1])) ? 0.f : -INFINITY;   // s-2 -> s, s = 2j+1
        skip_b[p] = ((j + 1 < U) && (tg[j] != tg[j + 1])) ? 0.f : -INFINITY;         // s -> s+2
        madd[2 * p] = (j <= U) ? 0.f : -INFINITY;                                     // blank state 2j exists for j <= U
        madd[2 * p + 1] = (j < U) ? 0.f : -INFINITY;                                  // label state 2j+1 for j < U
    }
    const float* lp = lp_ext + (int64_t)b * L * Sp + 2 * j0;
    float* al = alpha + (int64_t)b * (L + 2) * Sp + 2 * j0;
    auto load_row = [&](const float* base, int64_t row, float (&ev)[NP], float (&ov)[NP]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(base + row * Sp + 2 * q);
            ev[q] = v[0];
            ov[q] = v[1];
        }
    };
    auto load_lp = [&](int64_t row, float (&ev)[NP], float (&ov)[NP]) {   // table row with the states beyond 2U+1 masked
        load_row(lp, row, ev, ov);
#pragma unroll
        for (int q = 0; q < NP; ++q) { ev[q] += madd[2 * q]; ov[q] += madd[2 * q + 1]; }
    };
    auto store_row = [&](int64_t row, const float (&ev)[NP], const float (&ov)[NP]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<f32x2*>(al + row * Sp + 2 * q) = f32x2{ev[q], ov[q]};
    };
    float e[NP], o[NP], le[NP], lo[NP];

    if (PHASE == 0) {
        const int k0 = chunk * W;                      // first step of this launch (step k: alpha frame k / beta frame Tb-1-k)
        if (wave == 0) {        // alpha: frames 0 .. mid
            bool have = false;
            if (k0 == 0) {
                load_lp(0, le, lo);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    e[p] = (j0 + p == 0) ? le[p] : -INFINITY;
                    o[p] = (j0 + p == 0) ? lo[p] : -INFINITY;
                }
                store_row(0, e, o);
                ctc_chain<NP, false, false>(lp, al, 1, min(W - 1, mid), e, o, skip_f, madd, 0.f, -1, 0);
                have = W - 1 >= mid;
            } else if (k0 <= mid) {
                load_row(al, k0 - 1, e, o);
                ctc_chain<NP, false, false>(lp, al, k0, min(W, mid - k0 + 1), e, o, skip_f, madd, 0.f, -1, 0);
                have = k0 + W - 1 >= mid;
            }
            if (last && !have) load_row(al, mid, e, o);
        } else {                // beta: frames Tb-1 .. mid+1 in chunks; the last launch takes it onto mid (stored at row L)
            const int kmax = Tb - 2 - mid;             // last step whose frame is > mid (-1: none, Tb == 1)
            bool have = false;
            if (k0 == 0 && kmax >= 0) {
                load_lp(Tb - 1, le, lo);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    e[p] = (j0 + p == U) ? le[p] : -INFINITY;
                    o[p] = (j0 + p == U - 1) ? lo[p] : -INFINITY;
                }
                store_row(Tb - 1, e, o);
                ctc_chain<NP, true, false>(lp, al, Tb - 2, min(W - 1, kmax), e, o, skip_b, madd, 0.f, -1, 0);
                have = W - 1 >= kmax;
            } else if (k0 > 0 && k0 <= kmax) {
                load_row(al, Tb - k0, e, o);
                ctc_chain<NP, true, false>(lp, al, Tb - 1 - k0, min(W, kmax - k0 + 1), e, o, skip_b, madd, 0.f, -1, 0);
                have = k0 + W - 1 >= kmax;
            }
            if (last) {
                if (kmax < 0) {                        // Tb == 1: beta starts on the meeting frame itself
                    load_lp(mid, le, lo);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        e[p] = (j0 + p == U) ? le[p] : -INFINITY;
                        o[p] = (j0 + p == U - 1) ? lo[p] : -INFINITY;
                    }
                    store_row(L, e, o);
                } else {
                    if (!have) load_row(al, mid + 1, e, o);
                    ctc_chain<NP, true, false>(lp, al, mid, 1, e, o, skip_b, madd, 0.f, mid, L);
                }
#pragma unroll
                for (int p = 0; p < NP; ++p) { xch[lane][2 * p] = e[p]; xch[lane][2 * p + 1] = o[p]; }
            }
        }
        if (!last) return;
        __syncthreads();
        if (wave == 0) {        // log p(l|x) = lse_s( alpha_mid(s) + beta_mid(s) - lp_mid(s) )
            load_lp(mid, le, lo);
            float v[2 * NP], m = -INFINITY;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                v[2 * p] = (le[p] == -INFINITY) ? -INFINITY : e[p] + xch[lane][2 * p] - le[p];
                v[2 * p + 1] = (lo[p] == -INFINITY) ? -INFINITY : o[p] + xch[lane][2 * p + 1] - lo[p];
                m = fmaxf(m, fmaxf(v[2 * p], v[2 * p + 1]));
            }
            m = wave_max(m);
            const float ms = fmaxf(m, NEG_BIG);
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 2 * NP; ++q) sum += __builtin_amdgcn_exp2f(v[q] - ms);
            sum = wave_sum(sum);
            if (lane == 0) nll[b] = -(ms + __builtin_amdgcn_logf(sum)) * LN2;
        }
    } else {
        const float nll2 = nll[b] * LOG2E;
        float ae[NP], ao[NP];
        if (wave == 0) {
            load_row(al, mid, e, o);                    // alpha_mid: start state of the forward continuation
        } else {
            load_row(al, L, e, o);                      // beta_mid
            load_row(al, mid, ae, ao);                  // alpha_mid, for the occupancy of row mid
            load_lp(mid, le, lo);
        }
        __syncthreads();                                // wave 0 has read row mid before wave 1 overwrites it
        if (wave == 0) {
            ctc_chain<NP, false, true>(lp, al, mid + 1, Tb - 1 - mid, e, o, skip_f, madd, nll2, -1, 0);
        } else {
            float oe[NP], oo[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                oe[p] = occupancy(ae[p], e[p], le[p], nll2);
                oo[p] = occupancy(ao[p], o[p], lo[p], nll2);
            }
            store_row(mid, oe, oo);
            ctc_chain<NP, true, true>(lp, al, mid - 1, mid, e, o, skip_b, madd, nll2, -1, 0);
        }
    }
}

// Flag-pipelined forward recursion: the same two half-length chains as ctc_mitm_kernel<NP, 0>, launched ONCE on a second stream
// beside the (single, chunk-major) log-sum-exp launch.  Each wavefront waits, chunk by chunk, for the arrival counter of the W
// table rows it is about to consume (relaxed agent-scope poll, bounded), issues one agent-scope acquire so that its CU's L1 holds
// no stale line, and runs the chain over those rows - so the HBM-bound pass and the latency-bound recursion overlap inside one
// pair of launches, with no event hand-offs (7-14 us each on this stack) between chunks.
__device__ __forceinline__ bool ctc_wait_rows(const int* ctr, int need) {
    if (need <= 0) return true;
    bool ok = false;
    for (int spin = 0; spin < (1 << 21); ++spin) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) { ok = true; break; }
        __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return ok;
}

template <int NP>
__global__ __launch_bounds__(128) void ctc_mitm_flag_kernel(const float* __restrict__ lp_ext, const int32_t* __restrict__ in_len,
                                                            const int64_t* __restrict__ targets, int32_t* __restrict__ tgt_len, int L,
                                                            int Umax, float* __restrict__ alpha, float* __restrict__ nll, int W,
                                                            int nchunks, const int* __restrict__ arrivals, int64_t arr_stride) {
    constexpr int Sp = 128 * NP;
    __shared__ float xch[64][2 * NP];
    __shared__ int failed;
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = lane * NP;
    const int64_t* tg = targets + (int64_t)b * Umax;
    int n = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) n += (j0 + p < Umax && tg[j0 + p] != 0) ? 1 : 0;
    const int U = (int)wave_sum((float)n);
    if (threadIdx.x == 0) { tgt_len[b] = U; failed = 0; }
    const int Tb = min(in_len[b], L);
    if (Tb <= 0) {
        if (threadIdx.x == 0) nll[b] = (U == 0) ? 0.f : INFINITY;
        return;
    }
    __syncthreads();
    const int mid = Tb >> 1;
    float skip_f[NP], skip_b[NP], madd[2 * NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int j = j0 + p;
        skip_f[p] = ((j >= 1 && j < U) && (tg[j] != tg[j - 1])) ? 0.f : -INFINITY;
        skip_b[p] = ((j + 1 < U) && (tg[j] != tg[j + 1])) ? 0.f : -INFINITY;
        madd[2 * p] = (j <= U) ? 0.f : -INFINITY;
        madd[2 * p + 1] = (j < U) ? 0.f : -INFINITY;
    }
    const float* lp = lp_ext + (int64_t)b * L * Sp + 2 * j0;
    float* al = alpha + (int64_t)b * (L + 2) * Sp + 2 * j0;
    const int* arr_f = arrivals + (int64_t)b * arr_stride;      // forward-frame chunks
    const int* arr_b = arr_f + nchunks;                        // backward-frame chunks
    auto load_row = [&](const float* base, int64_t row, float (&ev)[NP], float (&ov)[NP]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const f32x2 v = *reinterpret_cast<const f32x2*>(base + row * Sp + 2 * q);
            ev[q] = v[0];
            ov[q] = v[1];
        }
    };
    auto load_lp = [&](int64_t row, float (&ev)[NP], float (&ov)[NP]) {
        load_row(lp, row, ev, ov);
#pragma unroll
        for (int q = 0; q < NP; ++q) { ev[q] += madd[2 * q]; ov[q] += madd[2 * q + 1]; }
    };
    auto store_row = [&](int64_t row, const float (&ev)[NP], const float (&ov)[NP]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) *reinterpret_cast<f32x2*>(al + row * Sp + 2 * q) = f32x2{ev[q], ov[q]};
    };
    float e[NP], o[NP], le[NP], lo[NP];
    bool ok = true;
    if (wave == 0) {            // alpha: frames 0 .. mid, chunk c = frames [cW, (c+1)W)
        for (int c = 0; c * W <= mid && ok; ++c) {
            const int k0 = c * W, rows = min(k0 + W, mid + 1) - k0;
            ok = ctc_wait_rows(arr_f + c, rows);
            if (!ok) break;
            if (c == 0) {
                load_lp(0, le, lo);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    e[p] = (j0 + p == 0) ? le[p] : -INFINITY;
                    o[p] = (j0 + p == 0) ? lo[p] : -INFINITY;
                }
                store_row(0, e, o);
                ctc_chain<NP, false, false>(lp, al, 1, rows - 1, e, o, skip_f, madd, 0.f, -1, 0);
            } else {
                ctc_chain<NP, false, false>(lp, al, k0, rows, e, o, skip_f, madd, 0.f, -1, 0);
            }
        }
    } else {                    // beta: frames Tb-1 .. mid+1 in chunks (step k <-> frame Tb-1-k), then onto mid (row L)
        const int kmax = Tb - 2 - mid;
        for (int c = 0; c * W <= kmax && ok; ++c) {
            const int k0 = c * W, rows = min(k0 + W, kmax + 1) - k0;
            ok = ctc_wait_rows(arr_b + c, rows);
            if (!ok) break;
            if (c == 0) {
                load_lp(Tb - 1, le, lo);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    e[p] = (j0 + p == U) ? le[p] : -INFINITY;
                    o[p] = (j0 + p == U - 1) ? lo[p] : -INFINITY;
                }
                store_row(Tb - 1, e, o);
                ctc_chain<NP, true, false>(lp, al, Tb - 2, rows - 1, e, o, skip_b, madd, 0.f, -1, 0);
            } else {
                ctc_chain<NP, true, false>(lp, al, Tb - 1 - k0, rows, e, o, skip_b, madd, 0.f, -1, 0);
            }
        }
        if (ok) {               // the meeting frame's row belongs to forward chunk mid / W
            const int cm = mid / W;
            ok = ctc_wait_rows(arr_f + cm, min(cm * W + W, mid + 1) - cm * W);
        }
        if (ok) {
            if (kmax < 0) {
                load_lp(mid, le, lo);
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    e[p] = (j0 + p == U) ? le[p] : -INFINITY;
                    o[p] = (j0 + p == U - 1) ? lo[p] : -INFINITY;
                }
                store_row(L, e, o);
            } else {
                ctc_chain<NP, true, false>(lp, al, mid, 1, e, o, skip_b, madd, 0.f, mid, L);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) { xch[lane][2 * p] = e[p]; xch[lane][2 * p + 1] = o[p]; }
        }
    }
    if (!ok && lane == 0) failed = 1;
    __syncthreads();
    if (failed) {               // a producer never arrived (cannot happen unless the pass faulted): fail loudly, never hang
        if (threadIdx.x == 0) nll[b] = __builtin_nanf("");
        return;
    }
    if (wave == 0) {
        load_lp(mid, le, lo);
        float v[2 * NP], m = -INFINITY;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            v[2 * p] = (le[p] == -INFINITY) ? -INFINITY : e[p] + xch[lane][2 * p] - le[p];
            v[2 * p + 1] = (lo[p] == -INFINITY) ? -INFINITY : o[p] + xch[lane][2 * p + 1] - lo[p];
            m = fmaxf(m, fmaxf(v[2 * p], v[2 * p + 1]));
        }
        m = wave_max(m);
        const float ms = fmaxf(m, NEG_BIG);
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 2 * NP; ++q) sum += __builtin_amdgcn_exp2f(v[q] - ms);
        sum = wave_sum(sum);
        if (lane == 0) nll[b] = -(ms + __builtin_amdgcn_logf(sum)) * LN2;
    }
}

__global__ void ctc_mean_kernel(const float* __restrict__ nll, const int32_t* __restrict__ tgt_len, int B, float* __restrict__ loss) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) s += nll[b] / (float)max(tgt_len[b], 1);
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s / (float)B;
}

// gradient stream: grid (RB, B); each workgroup walks rows t = blockIdx.x, += gridDim.x of utterance b.
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ in_len,
                                                       const int64_t* __restrict__ targets, const int32_t* __restrict__ tgt_len,
                                                       int B, int L, int V, int Umax, int blank, const float* __restrict__ lse,
                                                       const float* __restrict__ occ, const float* __restrict__ gout,
                                                       float* __restrict__ grad, int64_t ldg) {
    extern __shared__ float corr[];  // V floats
    const int b = blockIdx.y, tid = threadIdx.x;
    const int Sfull = ctc_row_stride(Umax), Sb = 2 * tgt_len[b] + 1;
    const int Tb = min(in_len[b], L);
    const float scale = gout[0] / ((float)B * (float)max(tgt_len[b], 1));
    for (int i = tid; i < V; i += 256) corr[i] = 0.f;
    __syncthreads();
    for (int t = blockIdx.x; t < L; t += gridDim.x) {
        const int64_t row = (int64_t)b * L + t;
        float* g = grad + row * ldg;
        const int gmis = (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3);
        if (t >= Tb) {  // padded frame: zero gradient, no reads
            const int peel = min((4 - gmis) & 3, V);
            if (tid < peel) g[tid] = 0.f;
            const int nv4 = (V - peel) >> 2;
            f32x4* g4 = reinterpret_cast<f32x4*>(g + peel);
            for (int i = tid; i < nv4; i += 256) g4[i] = f32x4{0, 0, 0, 0};
            const int tail0 = peel + nv4 * 4;
            if (tid < V - tail0) g[tail0 + tid] = 0.f;
            continue;
        }
        const float* x = logits + row * ldl;
        for (int sidx = tid; sidx < Sb; sidx += 256) {
            const int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            atomicAdd(&corr[lab], occ[((int64_t)b * (L + 2) + t) * Sfull + sidx]);
        }
        __syncthreads();
        const float l = lse[row];
        const int xmis = (int)((reinterpret_cast<uintptr_t>(x) >> 2) & 3);
        if (xmis == gmis) {
            const int peel = min((4 - gmis) & 3, V);
            if (tid < peel) g[tid] = scale * (__expf(x[tid] - l) - corr[tid]);
            const int nv4 = (V - peel) >> 2;
            const f32x4* x4 = reinterpret_cast<const f32x4*>(x + peel);
            f32x4* g4 = reinterpret_cast<f32x4*>(g + peel);
            for (int i = tid; i < nv4; i += 256) {
                const f32x4 v = x4[i];
                const int c = peel + i * 4;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = scale * (__expf(v[j] - l) - corr[c + j]);
                g4[i] = o;
            }
            const int tail0 = peel + nv4 * 4;
            if (tid < V - tail0) g[tail0 + tid] = scale * (__expf(x[tail0 + tid] - l) - corr[tail0 + tid]);
        } else {
            for (int c = tid; c < V; c += 256) g[c] = scale * (__expf(x[c] - l) - corr[c]);
        }
        __syncthreads();
        for (int sidx = tid; sidx < Sb; sidx += 256) {
            const int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            corr[lab] = 0.f;
        }
        __syncthreads();
    }
}

template <int PHASE>
int launch_recursion(hipStream_t s, const float* lp_ext, const int32_t* in_len, const int64_t* targets, int32_t* tgt_len, int B, int L,
                     int Umax, float* alpha, float* nll, int chunk, int W, int last) {
#define LAUNCH_MITM(NP_)                                                                                                              \
    hipLaunchKernelGGL((ctc_mitm_kernel<NP_, PHASE>), dim3(B), dim3(128), 0, s, lp_ext, in_len, targets, tgt_len, L, Umax, alpha, nll, \
                       chunk, W, last)
    switch (ctc_np(Umax)) {
        case 1: LAUNCH_MITM(1); break;
        case 2: LAUNCH_MITM(2); break;
        case 4: LAUNCH_MITM(4); break;
        default: LAUNCH_MITM(8); break;
    }
#undef LAUNCH_MITM
    return 0;
}

// two reusable events per host thread for the fork / join of the pipelined forward (no timing, so recording is cheap)
struct PipeEvents {
    hipEvent_t main_done = nullptr, aux_done = nullptr;
    bool ok() {
        if (!main_done) {
            if (hipEventCreateWithFlags(&main_done, hipEventDisableTiming) != hipSuccess) return false;
            if (hipEventCreateWithFlags(&aux_done, hipEventDisableTiming) != hipSuccess) return false;
        }
        return true;
    }
};

}  // namespace

extern "C" int asr_ctc_workspace_stride(int Umax) { return ctc_row_stride(Umax); }

extern "C" int asr_ctc_loss_fwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                int L, int V, int Umax, int blank, float* lse, float* lp_ext, float* alpha, float* nll,
                                int32_t* tgt_len, void* aux_stream, int n_chunks) {
    ASR_REQUIRE(logits && in_len && targets && lse && lp_ext && alpha && nll && tgt_len, ASR_ERR_ARG, "ctc_fwd: null pointer");
    ASR_REQUIRE(B > 0 && L > 0 && V > 1 && Umax > 0 && blank >= 0 && blank < V && ldl >= V, ASR_ERR_ARG, "ctc_fwd: bad sizes");
    ASR_REQUIRE(Umax + 1 <= 512, ASR_ERR_UNSUPPORTED, "ctc_fwd: Umax=%d too long (U+1 must be <= 512)", Umax);
    ASR_REQUIRE(asr_aligned(lp_ext, 16) && asr_aligned(alpha, 16), ASR_ERR_ALIGN, "ctc_fwd: workspaces must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream), s2 = static_cast<hipStream_t>(aux_stream);
    if (!s2 || s2 == s || n_chunks <= 1 || L < 64) {   // single stream: one pass over the logits, then the two half-length chains
        hipLaunchKernelGGL(ctc_lse_gather_kernel, dim3(B * L), dim3(256), 0, s, logits, ldl, in_len, targets, L, V, Umax, blank, lse,
                           lp_ext, 0, 0, nullptr, 0, 0, B);
        launch_recursion<0>(s, lp_ext, in_len, targets, tgt_len, B, L, Umax, alpha, nll, 0, L, 1);
        ASR_LAUNCH_CHECK("ctc_loss_fwd");
        return 0;
    }
    // flag-pipelined: ONE chunk-major log-sum-exp launch on `stream` publishes table rows chunk by chunk (write-through stores +
    // arrival counters in row L+1 of the alpha workspace), ONE recursion launch on `aux_stream` consumes them as they arrive
    static thread_local PipeEvents ev;
    ASR_REQUIRE(ev.ok(), ASR_ERR_UNSUPPORTED, "ctc_fwd: cannot create events");
    const int steps = L / 2 + 1;                           // alpha takes mid + 1 <= L/2 + 1 steps, beta at most as many
    const int W = ((steps + n_chunks - 1) / n_chunks + CTC_RPB - 1) / CTC_RPB * CTC_RPB;
    const int nc = (steps + W - 1) / W;
    const int Sp = ctc_row_stride(Umax);
    ASR_REQUIRE(2 * nc <= Sp, ASR_ERR_UNSUPPORTED, "ctc_fwd: too many chunks (%d) for the counter row", nc);
    int* arrivals = reinterpret_cast<int*>(alpha + (int64_t)(L + 1) * Sp);
    const int64_t arr_stride = (int64_t)(L + 2) * Sp;
#define HIP_OK(call)                                                          \
    do {                                                                      \
        hipError_t e__ = (call);                                              \
        if (e__ != hipSuccess) {                                              \
            asr_set_error("ctc_loss_fwd: %s", hipGetErrorString(e__));        \
            return (int)e__;                                                  \
        }                                                                     \
    } while (0)
    HIP_OK(hipMemset2DAsync(arrivals, (size_t)arr_stride * sizeof(float), 0, (size_t)2 * nc * sizeof(int), B, s));
    HIP_OK(hipEventRecord(ev.main_done, s));               // fork: the recursion starts behind the logits' producer and the memset
    HIP_OK(hipStreamWaitEvent(s2, ev.main_done, 0));
#define LAUNCH_FLAG(NP_)                                                                                                         \
    hipLaunchKernelGGL((ctc_mitm_flag_kernel<NP_>), dim3(B), dim3(128), 0, s2, lp_ext, in_len, targets, tgt_len, L, Umax, alpha, \
                       nll, W, nc, arrivals, arr_stride)
    switch (ctc_np(Umax)) {
        case 1: LAUNCH_FLAG(1); break;
        case 2: LAUNCH_FLAG(2); break;
        case 4: LAUNCH_FLAG(4); break;
        default: LAUNCH_FLAG(8); break;
    }
#undef LAUNCH_FLAG
    hipLaunchKernelGGL(ctc_lse_gather_kernel, dim3(nc * B * 2 * (W / CTC_RPB)), dim3(256), 0, s, logits, ldl, in_len, targets, L, V, Umax,
                       blank, lse, lp_ext, 0, W, arrivals, arr_stride, nc, B);
    HIP_OK(hipEventRecord(ev.aux_done, s2));               // join
    HIP_OK(hipStreamWaitEvent(s, ev.aux_done, 0));
#undef HIP_OK
    ASR_LAUNCH_CHECK("ctc_loss_fwd");
    return 0;
}

// bf16 gradient (the trainer's path: the gradient is consumed by bf16 MFMA GEMMs only, which would round the f32 image on load to
// exactly these values - half the write here, half the read in both of ctc_fc's backward GEMMs).  Rows are 16-byte aligned on
// both sides (ldl % 4 == 0, ldg % 8 == 0, host-checked); columns V .. ldg-1 are written as zeros: ldg is chosen by the caller
// so that the GEMM kernels can treat the rows as padded to their tile width.
__global__ __launch_bounds__(256) void ctc_grad_bf16_kernel(const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ in_len,
                                                            const int64_t* __restrict__ targets, const int32_t* __restrict__ tgt_len,
                                                            int B, int L, int V, int Umax, int blank, const float* __restrict__ lse,
                                                            const float* __restrict__ occ, const float* __restrict__ gout,
                                                            bf16_t* __restrict__ grad, int64_t ldg) {
    extern __shared__ float corr[];  // V floats (+ up to 3 pad entries read by the last vector group)
    const int b = blockIdx.y, tid = threadIdx.x;
    const int Sfull = ctc_row_stride(Umax), Sb = 2 * tgt_len[b] + 1;
    const int Tb = min(in_len[b], L);
    const float scale = gout[0] / ((float)B * (float)max(tgt_len[b], 1));
    const int ng = (int)(ldg >> 2);          // 4-column groups of a gradient row, pad included
    for (int i = tid; i < V + 4; i += 256) corr[i] = 0.f;
    __syncthreads();
    for (int t = blockIdx.x; t < L; t += gridDim.x) {
        const int64_t row = (int64_t)b * L + t;
        bf16x4* g4 = reinterpret_cast<bf16x4*>(grad + row * ldg);
        if (t >= Tb) {  // padded frame: zero gradient, no reads
            for (int i = tid; i < ng; i += 256) g4[i] = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
            continue;
        }
        const float* x = logits + row * ldl;
        for (int sidx = tid; sidx < Sb; sidx += 256) {
            const int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            atomicAdd(&corr[lab], occ[((int64_t)b * (L + 2) + t) * Sfull + sidx]);
        }
        __syncthreads();
        const float l = lse[row];
        const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
        for (int i = tid; i < ng; i += 256) {
            const int c = i * 4;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (c + 4 <= V) {
                const f32x4 v = x4[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = scale * (__expf(v[j] - l) - corr[c + j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < V) o[j] = scale * (__expf(x[c + j] - l) - corr[c + j]);
            }
            g4[i] = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        }
        __syncthreads();
        for (int sidx = tid; sidx < Sb; sidx += 256) {
            const int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            corr[lab] = 0.f;
        }
        __syncthreads();
    }
}

extern "C" int asr_ctc_mean(void* stream, const float* nll, const int32_t* tgt_len, int B, float* loss) {
    ASR_REQUIRE(nll && tgt_len && loss && B > 0, ASR_ERR_ARG, "ctc_mean: bad args");
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), nll, tgt_len, B, loss);
    ASR_LAUNCH_CHECK("ctc_mean");
    return 0;
}

extern "C" int asr_ctc_loss_bwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                int L, int V, int Umax, int blank, const float* lse, const float* lp_ext, float* alpha,
                                const float* nll, const int32_t* tgt_len, const float* gout, void* grad, int grad_dtype, int64_t ldg) {
    ASR_REQUIRE(logits && in_len && targets && lse && lp_ext && alpha && nll && tgt_len && gout && grad, ASR_ERR_ARG,
                "ctc_bwd: null pointer");
    ASR_REQUIRE(B > 0 && L > 0 && V > 1 && Umax > 0 && ldl >= V && ldg >= V, ASR_ERR_ARG, "ctc_bwd: bad sizes");
    ASR_REQUIRE(grad_dtype == ASR_F32 || grad_dtype == ASR_BF16, ASR_ERR_ARG, "ctc_bwd: bad grad_dtype");
    ASR_REQUIRE(grad_dtype == ASR_F32 || (ldg % 8 == 0 && ldl % 4 == 0 && asr_aligned(grad, 16) && asr_aligned(logits, 16)), ASR_ERR_ALIGN,
                "ctc_bwd: a bf16 gradient needs 16-byte aligned rows (ldg %% 8 == 0, ldl %% 4 == 0)");
    ASR_REQUIRE((size_t)(V + 4) * sizeof(float) <= 64 * 1024, ASR_ERR_UNSUPPORTED, "ctc_bwd: V=%d exceeds the LDS occupancy vector", V);
    ASR_REQUIRE(Umax + 1 <= 512, ASR_ERR_UNSUPPORTED, "ctc_bwd: Umax too long");
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_recursion<1>(s, lp_ext, in_len, targets, const_cast<int32_t*>(tgt_len), B, L, Umax, alpha, const_cast<float*>(nll), 0, L, 1);
    int rb = (2048 + B - 1) / B;  // ~2048 workgroups in flight
    if (rb > L) rb = L;
    if (rb < 1) rb = 1;
    if (grad_dtype == ASR_BF16)
        hipLaunchKernelGGL(ctc_grad_bf16_kernel, dim3(rb, B), dim3(256), (size_t)(V + 4) * sizeof(float), s, logits, ldl, in_len, targets,
                           tgt_len, B, L, V, Umax, blank, lse, alpha, gout, reinterpret_cast<bf16_t*>(grad), ldg);
    else
        hipLaunchKernelGGL(ctc_grad_kernel, dim3(rb, B), dim3(256), (size_t)V * sizeof(float), s, logits, ldl, in_len, targets, tgt_len, B, L,
                           V, Umax, blank, lse, alpha, gout, reinterpret_cast<float*>(grad), ldg);
    ASR_LAUNCH_CHECK("ctc_loss_bwd");
    return 0;
}
