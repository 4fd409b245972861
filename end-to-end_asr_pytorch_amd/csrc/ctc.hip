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
#include <stdlib.h>

#include <type_traits>

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


// A logits row held in registers (V <= 256 * 4 * LSE_UNR - 3): every load of the row is issued before anything waits on it.
struct CtcRowRegs {
    f32x4 v[LSE_UNR];
    float xe;      // the thread's unaligned head / tail element
};
__device__ __forceinline__ bool ctc_row_fits_regs(int V) { return ((V + 3) >> 2) <= 256 * LSE_UNR; }
__device__ __forceinline__ void ctc_row_load(const float* __restrict__ x, int V, CtcRowRegs& r) {
    const int tid = threadIdx.x;
    const int mis = (int)((reinterpret_cast<uintptr_t>(x) >> 2) & 3);
    const int peel = min((4 - mis) & 3, V);
    const int nv4 = (V - peel) >> 2;
    const int tail0 = peel + nv4 * 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + peel);
#pragma unroll
    for (int j = 0; j < LSE_UNR; ++j) {
        const int i = tid + 256 * j;
        r.v[j] = (i < nv4) ? x4[i] : f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    }
    r.xe = -INFINITY;
    if (tid < peel) r.xe = x[tid];
    else if (tid - peel < V - tail0 && tid >= peel) r.xe = x[tail0 + tid - peel];
}
__device__ __forceinline__ void ctc_row_reduce(const CtcRowRegs& r, float& m, float& s) {
    m = r.xe;
#pragma unroll
    for (int j = 0; j < LSE_UNR; ++j) m = fmaxf(m, fmaxf(fmaxf(r.v[j][0], r.v[j][1]), fmaxf(r.v[j][2], r.v[j][3])));
    const float ms = (m == -INFINITY) ? 0.f : m;
    s = __expf(r.xe - ms);
#pragma unroll
    for (int j = 0; j < LSE_UNR; ++j)
        s += (__expf(r.v[j][0] - ms) + __expf(r.v[j][1] - ms)) + (__expf(r.v[j][2] - ms) + __expf(r.v[j][3] - ms));
}

// a row too long for the register path: online (max, sum-exp) over a strided walk
__device__ __forceinline__ void ctc_row_stream(const float* __restrict__ x, int V, float& m, float& s) {
    const int tid = threadIdx.x;
    m = -INFINITY;
    s = 0.f;
    const int mis = (int)((reinterpret_cast<uintptr_t>(x) >> 2) & 3);
    const int peel = min((4 - mis) & 3, V);
    const int nv4 = (V - peel) >> 2;
    const int tail0 = peel + nv4 * 4;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + peel);
    if (tid < peel) { m = x[tid]; s = 1.f; }
    for (int i = tid; i < nv4; i += 256) {
        const f32x4 v = x4[i];
        const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (m4 > m) { s *= __expf(m - m4); m = m4; }
        s += (__expf(v[0] - m) + __expf(v[1] - m)) + (__expf(v[2] - m) + __expf(v[3] - m));
    }
    if (tid < V - tail0) lse_combine(m, s, x[tail0 + tid], 1.f);
}

// second half of a table row: block-combine the per-thread (max, sum-exp), then gather the extended labels' base-2 log-probs
__device__ __forceinline__ void ctc_row_finish(const float* __restrict__ x, const int64_t* __restrict__ targets, int V, int Umax, int blank,
                                               float* __restrict__ lse_out, float* __restrict__ lp_ext, int b, int row, float m, float s,
                                               bool publish) {
    const int tid = threadIdx.x;
    // wave then block combine
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        lse_combine(m, s, m2, s2);
    }
    // target length of this utterance (loss.py:40 targets.ne(0).sum(1)): rides on the same block reduction
    int nlab = 0;
    for (int i = tid; i < Umax; i += 256) nlab += targets[(int64_t)b * Umax + i] != 0 ? 1 : 0;
    nlab = (int)wave_sum((float)nlab);
    __shared__ float sm[4], ss[4];
    __shared__ int sn[4];
    __shared__ float lse_sh;
    if ((tid & 63) == 0) { sm[tid >> 6] = m; ss[tid >> 6] = s; sn[tid >> 6] = nlab; }
    __syncthreads();
    if (tid == 0) {
        float M = sm[0], S = ss[0];
        for (int w = 1; w < 4; ++w) lse_combine(M, S, sm[w], ss[w]);
        const float l = M + logf(S);
        lse_sh = l;
        lse_out[row] = l;
    }
    const int Ub = sn[0] + sn[1] + sn[2] + sn[3];
    __syncthreads();
    const float lse = lse_sh;
    // the table row holds the states of THIS utterance's extended label sequence (2 * tgt_len + 1 of them); everything beyond is
    // -inf, so the recursion needs no state mask of its own
    const int Sp = ctc_row_stride(Umax), Sb = 2 * Ub + 1;
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

// The fused forward's form of ctc_row_finish: the row's labels are gathered from an LDS image of the row instead of being loaded a
// second time.  Why: with 8 pass workgroups per CU streaming 17 KB rows, 32 CUs x 8 x 17 KB = 4.3 MB are in flight per XCD - the size
// of its L2 - so by the time a row's (max, sum) are reduced its lines are gone and each of the ~52 distinct gathered labels is one more
// 64-byte sector from the fabric: 52 x 64 B = 3.3 KB per 16.9 KB row, the 1.23 x FETCH_SIZE the round-4 counters showed (705 MB against
// 574).  The image is written from the registers the row was loaded into (16-byte stores, lane-consecutive: conflict-free) into the
// LDS the launch holds anyway for its recursion workgroups' rings and a pass workgroup never used.  Two barriers per row, as before:
// one before the image is overwritten (the previous row's gathers are done; it also covers sm / ss / sn), one before it is read -
// every thread then combines the four wave partials itself (same order, same bits as thread 0 did).
// rowbuf: >= 16 * ceil(V / 4) + 32 bytes, 16-byte aligned.  Same table bits as ctc_row_finish.
__device__ __forceinline__ void ctc_row_finish_lds(const CtcRowRegs& r, const float* __restrict__ x, const int64_t* __restrict__ targets,
                                                   int V, int Umax, int blank, float* __restrict__ lse_out, float* __restrict__ lp_ext,
                                                   int b, int row, float m, float s, float* __restrict__ rowbuf, bool no_gather) {
    const int tid = threadIdx.x;
    const int mis = (int)((reinterpret_cast<uintptr_t>(x) >> 2) & 3);
    const int peel = min((4 - mis) & 3, V);
    const int nv4 = (V - peel) >> 2;
    const int tail0 = peel + nv4 * 4;
    __shared__ float sm[4], ss[4];
    __shared__ int sn[4];
    __syncthreads();                                  // the previous row's readers are done with the image and the partials
    f32x4* img4 = reinterpret_cast<f32x4*>(rowbuf);
    float* imgx = rowbuf + (size_t)((V + 3) >> 2) * 4;   // head / tail elements, as thread tid holds them in r.xe
#pragma unroll
    for (int j = 0; j < LSE_UNR; ++j) {
        const int i = tid + 256 * j;
        if (i < nv4) img4[i] = r.v[j];
    }
    if (tid < 8) imgx[tid] = r.xe;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        lse_combine(m, s, m2, s2);
    }
    int nlab = 0;
    for (int i = tid; i < Umax; i += 256) nlab += targets[(int64_t)b * Umax + i] != 0 ? 1 : 0;
    nlab = (int)wave_sum((float)nlab);
    if ((tid & 63) == 0) { sm[tid >> 6] = m; ss[tid >> 6] = s; sn[tid >> 6] = nlab; }
    __syncthreads();
    float M = sm[0], S = ss[0];
    for (int w = 1; w < 4; ++w) lse_combine(M, S, sm[w], ss[w]);
    const float lse = M + logf(S);
    if (tid == 0) lse_out[row] = lse;
    const int Ub = sn[0] + sn[1] + sn[2] + sn[3];
    const int Sp = ctc_row_stride(Umax), Sb = 2 * Ub + 1;
    auto gather = [&](int sidx) {
        float v = -INFINITY;
        if (sidx < Sb) {
            int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            lab = min(max(lab, 0), V - 1);
            const int p = lab - peel;
            const float xv = no_gather ? 0.f : (lab < peel ? imgx[lab] : (p < nv4 * 4 ? rowbuf[p] : imgx[peel + lab - tail0]));
            v = (xv - lse) * LOG2E;
        }
        return v;
    };
    for (int pr = tid; pr < Sp / 2; pr += 256) {
        const f32x2 v2 = {gather(2 * pr), gather(2 * pr + 1)};
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(lp_ext + (int64_t)row * Sp + 2 * pr),
                           __builtin_bit_cast(unsigned long long, v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// one table row (b, t): log-sum-exp over the vocabulary, then the gather of the extended labels' base-2 log-probs
__device__ __forceinline__ void ctc_lse_row(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ targets, int L, int V,
                                            int Umax, int blank, float* __restrict__ lse_out, float* __restrict__ lp_ext, int b, int t,
                                            bool publish) {
    const int row = b * L + t;
    const float* x = logits + (int64_t)row * ldl;
    const int tid = threadIdx.x;
    float m = -INFINITY, s = 0.f;
    if (ctc_row_fits_regs(V)) {
        CtcRowRegs r;
        ctc_row_load(x, V, r);
        ctc_row_reduce(r, m, s);
    } else {
        ctc_row_stream(x, V, m, s);
    }
    ctc_row_finish(x, targets, V, Umax, blank, lse_out, lp_ext, b, row, m, s, publish);
}

__global__ __launch_bounds__(256) void ctc_lse_gather_kernel(const float* __restrict__ logits, int64_t ldl,
                                                             const int32_t* __restrict__ in_len, const int64_t* __restrict__ targets,
                                                             int L, int V, int Umax, int blank,
                                                             float* __restrict__ lse_out, float* __restrict__ lp_ext) {
    // one workgroup per table row, blockIdx.x = b*L + t (the two-launch form; the fused form is ctc_fused_fwd_kernel)
    const int b = blockIdx.x / L, t = blockIdx.x - b * L;
    if (t >= in_len[b]) return;
    ctc_lse_row(logits, ldl, targets, L, V, Umax, blank, lse_out, lp_ext, b, t, false);
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
// Table rows are addressed through buffer descriptors (wave-uniform base in scalar registers, the lane's column offset in ONE
// vector register, the row as a scalar offset): no per-row 64-bit address arithmetic on the vector pipe, one register pair per
// row in flight.
typedef __amdgpu_buffer_rsrc_t ctc_rsrc_t;
__device__ __forceinline__ ctc_rsrc_t ctc_rsrc(const float* uniform_base, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_base), 0, (int)bytes, 0x00020000);
}

// PFO > 0 overrides the prefetch depth: the fused forward (ctc_fused_fwd_kernel) runs its chains beside the streaming pass, where a
// table row takes 2-3 us to arrive instead of ~0.6 - the ring must cover that at ~0.1 us per step.
// lp_u / al_u: wave-uniform bases of the utterance's table / workspace; voff: the lane's byte offset into a row (8 * NP * lane).
template <int NP, bool DIRB, bool OCC, int PFO = 0>
__device__ __forceinline__ void ctc_chain(const float* __restrict__ lp_u, float* __restrict__ al_u, int voff, int L, int t_first,
                                          int nsteps, float (&e)[NP], float (&o)[NP], const float (&skip_add)[NP],
                                          const float (&madd)[2 * NP], float nll2, int special_t, int special_row) {
    constexpr int PF = PFO > 0 ? PFO : (NP == 1 ? CTC_PF1 : (NP == 2 ? 8 : (NP == 4 ? 4 : 2)));
    constexpr int Sp = 128 * NP, ROWB = Sp * 4;
    if (nsteps <= 0) return;
    const ctc_rsrc_t rlp = ctc_rsrc(lp_u, (int64_t)L * ROWB), ral = ctc_rsrc(al_u, (int64_t)(L + 2) * ROWB);
    auto row_of = [&](int t) { return t == special_t ? special_row : t; };
    auto load_row = [&](ctc_rsrc_t r, int row, float (&dst)[2 * NP]) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, voff + 8 * q, row * ROWB, 0));
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
        const int soff = row_of(t) * ROWB;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            e[p] = en[p];
            o[p] = on[p];
            const f32x2 w = !OCC ? f32x2{e[p], o[p]}
                                 : f32x2{occupancy(ca[2 * p], e[p], cf[2 * p], nll2), occupancy(ca[2 * p + 1], o[p], cf[2 * p + 1], nll2)};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, w), ral, voff + 8 * p, soff, 0);
        }
    };
    const int dir = DIRB ? -1 : 1;
    float pf[PF][2 * NP], pa[PF][2 * NP];
#pragma unroll
    for (int i = 0; i < PF; ++i) {      // steps 0..PF-1, clamped to the chain (over-fetched rows are never consumed)
        const int t = t_first + dir * min(i, nsteps - 1);
        load_row(rlp, t, pf[i]);
        if (OCC) load_row(ral, row_of(t), pa[i]);
    }
    int k0 = 0;
    // register ring: slot i is consumed for step k0 + i and refilled at once with the row of step k0 + PF + i (loads return in
    // order, so the wait before a slot's use leaves the PF - 1 younger loads in flight)
    for (; k0 + PF <= nsteps; k0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            float cf[2 * NP], ca[2 * NP];
#pragma unroll
            for (int q = 0; q < 2 * NP; ++q) { cf[q] = pf[i][q]; ca[q] = OCC ? pa[i][q] : 0.f; }   // (columns >= 2U+1 of a table row are -inf)
            const int tn = t_first + dir * min(k0 + PF + i, nsteps - 1);
            load_row(rlp, tn, pf[i]);
            if (OCC) load_row(ral, row_of(tn), pa[i]);
            step(t_first + dir * (k0 + i), cf, ca);
        }
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        float ca[2 * NP], cf[2 * NP];
#pragma unroll
        for (int q = 0; q < 2 * NP; ++q) { cf[q] = pf[i][q]; ca[q] = OCC ? pa[i][q] : 0.f; }
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
                                                       int last, float* __restrict__ alpha2 = nullptr) {
    // PHASE 2: PHASE 1 without the occupancies - the continued chains store their raw alpha / beta rows into the SECOND workspace
    // `alpha2` (same layout) next to the opposite direction's rows of PHASE 0; the gradient pass combines the two (occupancy() once
    // per state there): the chain's step then is PHASE 0's (no second row to load, no exp2), about half of PHASE 1's
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
    const float* lp_u = lp_ext + (int64_t)b * L * Sp;       // wave-uniform bases + the lane's byte offset: ctc_chain's addressing
    float* al_u = alpha + (int64_t)b * (L + 2) * Sp;
    const int voff = 8 * j0;
    const float* lp = lp_u + 2 * j0;
    float* al = al_u + 2 * j0;
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
                ctc_chain<NP, false, false>(lp_u, al_u, voff, L, 1, min(W - 1, mid), e, o, skip_f, madd, 0.f, -1, 0);
                have = W - 1 >= mid;
            } else if (k0 <= mid) {
                load_row(al, k0 - 1, e, o);
                ctc_chain<NP, false, false>(lp_u, al_u, voff, L, k0, min(W, mid - k0 + 1), e, o, skip_f, madd, 0.f, -1, 0);
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
                ctc_chain<NP, true, false>(lp_u, al_u, voff, L, Tb - 2, min(W - 1, kmax), e, o, skip_b, madd, 0.f, -1, 0);
                have = W - 1 >= kmax;
            } else if (k0 > 0 && k0 <= kmax) {
                load_row(al, Tb - k0, e, o);
                ctc_chain<NP, true, false>(lp_u, al_u, voff, L, Tb - 1 - k0, min(W, kmax - k0 + 1), e, o, skip_b, madd, 0.f, -1, 0);
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
                    ctc_chain<NP, true, false>(lp_u, al_u, voff, L, mid, 1, e, o, skip_b, madd, 0.f, mid, L);
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
    } else if (PHASE == 2) {
        float* a2_u = alpha2 + (int64_t)b * (L + 2) * Sp;
        if (wave == 0) {
            load_row(al, mid, e, o);                    // alpha_mid: start state of the forward continuation
            ctc_chain<NP, false, false>(lp_u, a2_u, voff, L, mid + 1, Tb - 1 - mid, e, o, skip_f, madd, 0.f, -1, 0);
        } else {
            load_row(al, L, e, o);                      // beta_mid
#pragma unroll
            for (int q = 0; q < NP; ++q) *reinterpret_cast<f32x2*>(a2_u + 2 * j0 + (int64_t)mid * Sp + 2 * q) = f32x2{e[q], o[q]};
            ctc_chain<NP, true, false>(lp_u, a2_u, voff, L, mid - 1, mid, e, o, skip_b, madd, 0.f, -1, 0);
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
            ctc_chain<NP, false, true>(lp_u, al_u, voff, L, mid + 1, Tb - 1 - mid, e, o, skip_f, madd, nll2, -1, 0);
        } else {
            float oe[NP], oo[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                oe[p] = occupancy(ae[p], e[p], le[p], nll2);
                oo[p] = occupancy(ao[p], o[p], lo[p], nll2);
            }
            store_row(mid, oe, oo);
            ctc_chain<NP, true, true>(lp_u, al_u, voff, L, mid - 1, mid, e, o, skip_b, madd, nll2, -1, 0);
        }
    }
}

// ---- fused forward: the streaming pass and the recursion in ONE launch ---------------------------------------------------------
// The log-sum-exp / gather pass is HBM-bound (thousands of workgroups, and it wants every wave slot of the chip: at 4 workgroups
// per CU it runs 132 us, at 8 per CU 120, as a kernel of its own 109), the recursion is a latency-bound chain on two wavefronts
// per utterance (56 us on its own).  Run one after the other the chain adds its whole length to the op; as two launches on two
// streams the hand-off events cost more than the overlap saved.  Here both are roles of one grid: the first B workgroups are
// RECURSION workgroups (wave 0: alpha, wave 1: beta of utterance blockIdx.x), all others are PASS workgroups that walk the
// table rows chunk-major - chunk c holds, for every utterance, the frames the two wavefronts consume in their c-th piece -
// store them write-through, drain, and add the row count to the (utterance, direction, chunk) arrival counter at agent scope.
// A recursion wavefront polls its chunk's counter (relaxed, s_sleep, bounded) and then runs the chain over the chunk.
// Pass workgroups never wait, so the grid needs no co-residency guarantee: whatever the dispatch order, every counter is
// eventually complete.
//
// One kernel = one register / LDS allocation for both roles, and the pass needs the occupancy: a register ring deep enough to
// cover a loaded HBM round trip (2-3 us = 24+ table rows at 0.1 us per step) cost 118-247 registers, i.e. 1-4 workgroups per
// CU, and the fused launch ran SLOWER than the two-launch form (163-380 us against 154).  So the chain's look-ahead lives in
// LDS: table rows arrive by LDS-DMA (global_load_lds, two 512-byte rows per wave instruction, no registers) into a 20-row ring
// per wavefront and are picked up by ds_read one pair ahead.  vmcnt bookkeeping is by hand (a pair's DMA is followed by exactly
// 3 * CTC_RING_PAIRS - 3 vector-memory operations - one DMA and two row stores per pair - before it is waited for); dummy DMAs
// past a chunk's end keep that count constant.  ~70 registers + 22.5 KiB LDS: 7 workgroups per CU.
// ring depth in row pairs (1 KiB each, + one dummy slot) per chain wavefront: 10 pairs -> 22.5 KiB per workgroup, 7 per CU;
// 8 pairs -> 18.5 KiB, 8 per CU

__device__ __forceinline__ bool ctc_wait_rows(const int* ctr, int need) {
    if (need <= 0) return true;
    for (int spin = 0; spin < (1 << 21); ++spin) {
        if (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need) return true;
        __builtin_amdgcn_s_sleep(4);
    }
    return false;
}

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void global_cvoid_t;

// One step of the NP = 1 recursion (lane i owns states 2i, 2i+1; c = the frame's table row entries of this lane).
template <bool DIRB>
__device__ __forceinline__ void ctc_step1(float& e, float& o, const f32x2 c, float skip_add) {
    float en, on;
    if (!DIRB) {
        const float left = dpp_from_prev_lane(o);
        en = l2se2(e, left) + c[0];
        on = l2se3(o, e, left + skip_add) + c[1];
    } else {
        const float re = dpp_from_next_lane(e), ro = dpp_from_next_lane(o);
        en = l2se2(e, o) + c[0];
        on = l2se3(o, re, ro + skip_add) + c[1];
    }
    e = en;
    o = on;
}

// A run of an EVEN number of steps of one direction's chain: steps k = 0..nsteps-1 over frames first + dir * k, all of them
// published (the caller has seen their arrival counters complete: a row that is read before its producer has written it would
// stay in this XCD's L2 and be served again, stale, when it is read for real).  skip0: frame `first` is the chain's initial
// state - its step is not taken (it only keeps the row pairs aligned to the chunk grid).  ring: this wavefront's LDS ring.
// avail / W: the chain runs as ONE pipeline over the whole direction and never reads ahead of the arrivals: `avail` is an LDS word in
// which a polling wavefront of the same workgroup keeps the number of COMPLETE chunks (W rows each) of this direction (-1: a producer
// never arrived); a pair's DMA is issued only once its chunk is complete, and the wavefront waits right there, its ring still primed -
// where a chain cut into runs at chunk boundaries drained and refilled its ring (a DMA round trip) at every boundary it reached in time.
// Returns false when the producer side failed.
template <bool DIRB, int P>
__device__ __forceinline__ bool ctc_lds_chain(const float* __restrict__ lp_u, ctc_rsrc_t ral, int lane, char* ring, int first, int nsteps,
                                              bool skip0, float& e, float& o, float skip_add, const volatile int* avail, int W) {
    const int npairs = nsteps >> 1;
    if (npairs <= 0) return true;
    int avail_pairs = 0;
    bool alive = true;
    auto gate = [&](int q) {            // pair q (steps 2q, 2q + 1: one chunk, W is even) has arrived
        while (q >= avail_pairs) {
            const int a = __builtin_amdgcn_readfirstlane(*avail);
            if (a < 0) { alive = false; return; }
            avail_pairs = a * (W >> 1);
            if (q >= avail_pairs) __builtin_amdgcn_s_sleep(2);
        }
    };
    const char* src_lane = reinterpret_cast<const char*>(lp_u) + lane * 16;
    const unsigned rd = (unsigned)reinterpret_cast<uintptr_t>((lds_void_t*)ring) + lane * 8;   // LDS byte address of this lane's column
    // pair q = steps 2q, 2q+1.  alpha: rows (first + 2q, + 1);  beta: rows (first - 2q - 1, first - 2q): the LOWER row sits in the
    // first half of the slot either way (the DMA copies 1 KiB of the table as it lies in memory).  sc1: the rows were published
    // write-through by other CUs, the loads bypass this CU's L1.  Pairs past the end re-read pair 0 into a dummy slot: that keeps
    // the count of vector-memory operations between a pair's DMA and its wait constant.
    auto dma = [&](int q, int slot) {
        const bool real = q < npairs;
        if (real) gate(q);
        const int qq = real ? q : 0;
        const int lo = DIRB ? first - 2 * qq - 1 : first + 2 * qq;
        __builtin_amdgcn_global_load_lds((global_cvoid_t*)(src_lane + (int64_t)lo * 512), (lds_void_t*)(ring + (real ? slot : P) * 1024), 16, 0, 16);
    };
    auto step = [&](int t, const f32x2 c) {
        ctc_step1<DIRB>(e, o, c, skip_add);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{e, o}), ral, lane * 8, t * 512, 0);
    };
#pragma unroll
    for (int j = 0; j < P; ++j) dma(j, j);
    if (!alive) return false;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the pair read ahead: lower / upper table row.  Every asm statement takes them in-out ("+v"): one live range in one register
    // pair from the ds_read to the wait - a fresh output per read would let the compiler place a register copy (a loop phi) between
    // a read and its wait, copying the register before the data lands.
    f32x2 nlo = {0.f, 0.f}, nhi = {0.f, 0.f};
    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512" : "+v"(nlo), "+v"(nhi) : "v"(rd) : "memory");
    const int dir = DIRB ? -1 : 1;
    for (int q0 = 0; q0 < npairs; q0 += P) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const int q = q0 + j;
            if (q < npairs) {
                // the pair read ahead one iteration ago: waited for and copied INSIDE one asm statement (a compiler-side copy of a
                // register that an asynchronous ds_read is still to write could be scheduled ahead of a separate wait)
                f32x2 clo, chi;
                asm volatile("s_waitcnt lgkmcnt(0)\n\tv_mov_b64 %[a], %[l]\n\tv_mov_b64 %[b], %[h]"
                             : [a] "=&v"(clo), [b] "=&v"(chi), [l] "+v"(nlo), [h] "+v"(nhi)::"memory");
                const f32x2 c0 = DIRB ? chi : clo, c1 = DIRB ? clo : chi;      // step 2q, step 2q + 1
                dma(q + P, j);                                                   // slot j has just been read: refill it
                if (!alive) return false;
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"(3 * P - 3) : "memory");  // pair q + 1 has landed
                if (j + 1 < P)
                    asm volatile("ds_read_b64 %0, %2 offset:%3\n\tds_read_b64 %1, %2 offset:%4"
                                 : "+v"(nlo), "+v"(nhi) : "v"(rd), "i"((j + 1) * 1024), "i"((j + 1) * 1024 + 512) : "memory");
                else
                    asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:512" : "+v"(nlo), "+v"(nhi) : "v"(rd) : "memory");
                if (!(skip0 && q == 0)) step(first + dir * 2 * q, c0);
                else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{e, o}), ral, lane * 8, first * 512, 0);   // (the initial row; keeps the op count)
                step(first + dir * (2 * q + 1), c1);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nlo), "+v"(nhi)::"memory");      // (the read-ahead of a pair nobody consumes)
    return true;
}

template <int P>
__global__ __launch_bounds__(256, 8) void ctc_fused_fwd_kernel(const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ in_len,
                                                               const int64_t* __restrict__ targets, int Bn, int L, int V, int Umax, int blank,
                                                               float* __restrict__ lse_out, float* __restrict__ lp_ext,
                                                               float* __restrict__ alpha, float* __restrict__ nll,
                                                               int32_t* __restrict__ tgt_len, int W, int nchunks, int* __restrict__ arrivals,
                                                               int64_t arr_stride, int* __restrict__ extra, float* __restrict__ mean_loss,
                                                               int dbg) {
    constexpr int Sp = 128;
    __shared__ __attribute__((aligned(16))) char ring[2][(P + 1) * 1024];
    // extra[8]: finished recursion workgroups (extra[0..7] spare)
    const int RPB = (dbg >> 8) & 0xff;                       // table rows per pass workgroup (W is a multiple of it)
    // dbg bit 7 (timeline experiment, needs the counters in a buffer of their own): every workgroup leaves its start / end time (100 MHz
    // counter, low words) in the spare row L + 1 of the alpha workspace - tools/ctc_timeline.py
    unsigned* stamp = (dbg & 128) ? reinterpret_cast<unsigned*>(alpha + ((int64_t)(blockIdx.x >> 6) * (L + 2) + (L + 1)) * Sp) + (blockIdx.x & 63) * 2
                                  : nullptr;
    unsigned long long stamp_t0 = 0;
    if (stamp && threadIdx.x == 0) {      // word 0: where it runs (HW_ID low 16 bits | XCC_ID << 16) - start times differ by < 5 us
        stamp_t0 = __builtin_amdgcn_s_memrealtime();
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        stamp[0] = (hw & 0xffff) | ((xcc & 0xf) << 16);
    }
    if ((int)blockIdx.x >= Bn) {
        // ---- PASS workgroup (persistent): walks row groups gid = pid, pid + npass, ... in chunk-major order; a group = RPB consecutive
        // rows of one (chunk, utterance, direction).  Only wave 0 stores (64 state pairs + the row's lse), so the group's arrival is
        // signalled by wave 0 alone, and not by draining its stores on the spot: vector-memory operations retire in order, so once
        // the NEXT group's first row has landed in wave 0's registers every older store has left - the add costs no wait at all.
        // Tried on this walk in round 4 and dropped (tools/ctc_timeline.py stamps every workgroup's end: equal shares of 16 rows finish
        // 30 us apart, the chains 12-14 us after the last of them): utterance b's rows produced on XCD b mod 8 (neutral: 133 / 130 and
        // 143 / 146 us); the last partial round dealt row by row (neutral); groups, single rows, or the last half of the walk handed out
        // by queue heads, device-scope or per XCD (170-400 us: a returning atomic per unit under a saturated memory system costs more
        // than the imbalance); the row by LDS-DMA into the idle ring with the next row's DMA in flight during the reduction (mean
        // workgroup 6 % faster, slowest 8 % slower: 151 against 141 us); the label gather issued behind the row's own loads (155 / 143).
        if (dbg & 32) return;      // (bit 5: timing experiment - the chains alone)
        const int G = W / RPB, npass = gridDim.x - Bn, total = nchunks * Bn * 2 * G;
        const bool w0 = threadIdx.x < 64;
        // (dbg bit 1: the round-4 gather by a second global load of each label - A/B and the FETCH_SIZE attribution; bit 3: no gather, timing only)
        const bool stage = !(dbg & 2) && ctc_row_fits_regs(V) && (size_t)((V + 3) >> 2) * 16 + 32 <= sizeof(ring);
        int* pending = nullptr;
        int pending_count = 0;
        // (Round 5, measured and dropped: a head start for chunk 0 - the workgroups whose first item is not of chunk 0 sleeping 2..24 us so
        // that the chains' first chunk has the memory system to itself: 135 us without, 146 / 147 / 148 / 151 / 154 / 158 us with
        // 2 / 4 / 6 / 12 / 16 / 24 us - the kernel simply ends that much later, the pass is what it waits for.  And chunk 0 alone dealt row by
        // row, every later chunk in 4-row items as before: 151.6 against 146.9 us on the same box.)
        for (int gid = blockIdx.x - Bn; gid < total; gid += npass) {
            int bid = gid;
            const int chunk = bid / (Bn * 2 * G);
            bid -= chunk * (Bn * 2 * G);
            const int b = bid / (2 * G), gi = bid - b * 2 * G;
            const int dirc = gi / G, g0 = (gi - dirc * G) * RPB;
            const int Tb = min(in_len[b], L), mid = Tb >> 1;
            int count = 0;
            for (int r = 0; r < RPB; ++r) {              // frames only run out at the end of a direction
                const int k = chunk * W + g0 + r;
                const int t = dirc == 0 ? k : Tb - 1 - k;
                count += (dirc == 0 ? (t <= mid && t < Tb) : (t > mid)) ? 1 : 0;
            }
            if (count == 0) continue;
            const int kb = chunk * W + g0;
            const int tstep = dirc == 0 ? 1 : -1, t0 = dirc == 0 ? kb : Tb - 1 - kb;
            for (int r = 0; r < count; ++r) {
                const int t = t0 + tstep * r, row = b * L + t;
                const float* x = logits + (int64_t)row * ldl;
                float m, sx;
                if (stage) {                             // labels gathered from an LDS image of the row (ctc_row_finish_lds)
                    CtcRowRegs rr;
                    ctc_row_load(x, V, rr);
                    ctc_row_reduce(rr, m, sx);
                    if (pending && w0) {                 // (this row's loads are back: the previous group's stores retired before them)
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (threadIdx.x == 0) __hip_atomic_fetch_add(pending, pending_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    pending = nullptr;
                    ctc_row_finish_lds(rr, x, targets, V, Umax, blank, lse_out, lp_ext, b, row, m, sx, reinterpret_cast<float*>(&ring[0][0]),
                                       (dbg & 8) != 0);
                    continue;
                }
                if (ctc_row_fits_regs(V)) {
                    CtcRowRegs rr;
                    ctc_row_load(x, V, rr);
                    ctc_row_reduce(rr, m, sx);
                } else {
                    ctc_row_stream(x, V, m, sx);
                }
                if (pending && w0) {                     // (this row's loads are back: the previous group's stores retired before them)
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (threadIdx.x == 0) __hip_atomic_fetch_add(pending, pending_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                pending = nullptr;
                ctc_row_finish(x, targets, V, Umax, blank, lse_out, lp_ext, b, row, m, sx, true);
            }
            pending = arrivals + (int64_t)b * arr_stride + dirc * nchunks + chunk;
            pending_count = count;
        }
        if (pending && w0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) __hip_atomic_fetch_add(pending, pending_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (stamp && threadIdx.x == 0) stamp[1] = (unsigned)__builtin_amdgcn_s_memrealtime();
        return;
    }
    // ---- RECURSION workgroup: one utterance; wave 0 runs alpha, wave 1 beta, waves 2 / 3 only keep the barriers company ----
    if (dbg & 4) return;       // (timing experiment: the pass alone)
    __shared__ float xch[64][2];
    __shared__ int failed;
    __shared__ volatile int avail_sh[2];      // complete chunks of the alpha / beta direction (-1: a producer never arrived)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    const int64_t* tg = targets + (int64_t)b * Umax;
    const int U = (int)wave_sum((float)((lane < Umax && tg[lane] != 0) ? 1 : 0));
    const int Tb = min(in_len[b], L), mid = Tb >> 1;
    if (threadIdx.x == 0) {
        failed = 0;
        avail_sh[0] = avail_sh[1] = 0;
        tgt_len[b] = U;
        if (Tb <= 0) nll[b] = (U == 0) ? 0.f : INFINITY;
    }
    __syncthreads();
    const float* lp_u = lp_ext + (int64_t)b * L * Sp;
    float* al_u = alpha + (int64_t)b * (L + 2) * Sp;
    float e = -INFINITY, o = -INFINITY;
    bool ok = true;
    if (Tb > 0 && wv >= 2) {
        // POLLING wavefronts (2: alpha's chunks, 3: beta's): the device-scope counter reads stay out of the chains' hand-counted
        // vector-memory queue; a chain asks LDS
        const int d = wv - 2, klast_d = d == 0 ? mid : Tb - 2 - mid;
        const int* arr_d = arrivals + (int64_t)b * arr_stride + d * nchunks;
        for (int c = 0; c * W <= klast_d; ++c) {
            const bool got = (dbg & 16) ? true : ctc_wait_rows(arr_d + c, min(c * W + W, klast_d + 1) - c * W);      // (bit 4: never wait - timing only)
            if (lane == 0) {
                avail_sh[d] = got ? c + 1 : -1;
                // every arrival of this chunk has been counted: the word goes back to zero here (the caller's buffer is handed back
                // zeroed without a pass over all B x 2 x chunks words by the last utterance, which sat at the very end of the launch)
                if (got) __hip_atomic_store(const_cast<int*>(arr_d + c), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (!got) break;
        }
    }
    if (Tb > 0 && wv < 2) {
        __builtin_amdgcn_s_setprio(3);                    // the chains outrank the pass workgroups sharing this CU's SIMDs
        const ctc_rsrc_t ral = ctc_rsrc(al_u, (int64_t)(L + 2) * Sp * 4);
        const f32x2* lpl = reinterpret_cast<const f32x2*>(lp_u) + lane;           // this lane's (blank, label) column
        auto row_sc1 = [&](int t) {      // a published row read around the L1 (8-byte agent-scope load)
            const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(lpl + (int64_t)t * 64), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
            return __builtin_bit_cast(f32x2, v);
        };
        auto store_row = [&](int row) {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f32x2{e, o}), ral, lane * 8, row * 512, 0);
        };
        // steps k = 0 .. klast of this direction: alpha frame k, beta frame Tb - 1 - k; step 0 is the initial state
        const int klast = wv == 0 ? mid : Tb - 2 - mid;
        const int f0 = wv == 0 ? 0 : Tb - 1, dir = wv == 0 ? 1 : -1;
        const float skip = wv == 0 ? (((lane >= 1 && lane < U) && (tg[lane] != tg[lane - 1])) ? 0.f : -INFINITY)      // s-2 -> s, s = 2 lane + 1
                                   : (((lane + 1 < U) && (tg[lane] != tg[lane + 1])) ? 0.f : -INFINITY);            // s -> s+2
        auto init_state = [&](int t) {
            const f32x2 r0 = row_sc1(t);
            if (wv == 0) { e = (lane == 0) ? r0[0] : -INFINITY; o = (lane == 0) ? r0[1] : -INFINITY; }
            else { e = (lane == U) ? r0[0] : -INFINITY; o = (lane == U - 1) ? r0[1] : -INFINITY; }
        };
        // one pipeline over the whole direction, gated on the chunk counts the polling wavefronts keep in LDS (below)
        const volatile int* av = &avail_sh[wv];
        auto wait_chunk = [&](const volatile int* a, int c) {      // chunk c of that direction is complete
            for (;;) {
                const int v = __builtin_amdgcn_readfirstlane(*a);
                if (v < 0) return false;
                if (v > c) return true;
                __builtin_amdgcn_s_sleep(2);
            }
        };
        const int n = klast + 1;                                  // steps 0 .. klast; step 0 is the initial state
        if (n > 0) {
            ok = wait_chunk(av, 0);
            if (ok) {
                init_state(f0);
                if (wv == 0) ok = ctc_lds_chain<false, P>(lp_u, ral, lane, ring[0], f0, n & ~1, true, e, o, skip, av, W);
                else ok = ctc_lds_chain<true, P>(lp_u, ral, lane, ring[1], f0, n & ~1, true, e, o, skip, av, W);
            }
            if (ok && (n & 1)) {                                  // an odd direction ends on a row that comes by itself
                ok = wait_chunk(av, (n - 1) / W);
                if (ok) {
                    if (n > 1) {
                        const f32x2 tail = row_sc1(f0 + dir * (n - 1));
                        if (wv == 0) ctc_step1<false>(e, o, tail, skip);
                        else ctc_step1<true>(e, o, tail, skip);
                    }
                    store_row(f0 + dir * (n - 1));
                }
            }
        }
        if (wv == 1) {          // beta onto the meeting frame (stored as row L); its row belongs to forward chunk mid / W
            if (ok) ok = wait_chunk(&avail_sh[0], mid / W);
            if (ok) {
                if (klast < 0) init_state(mid);        // Tb == 1: beta starts on the meeting frame itself
                else ctc_step1<true>(e, o, row_sc1(mid), skip);
                store_row(L);
            }
            xch[lane][0] = e;
            xch[lane][1] = o;
        }
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();                                      // (failed initialised; beta's states visible)
    if (!ok && lane == 0) failed = 1;                     // a producer never arrived (cannot happen unless the pass faulted)
    __syncthreads();
    if (wv != 0) return;
    if (Tb > 0) {
        if (failed) {                                     // fail loudly, never hang
            if (lane == 0) nll[b] = __builtin_nanf("");
        } else {
            // log p(l|x) = lse_s( alpha_mid(s) + beta_mid(s) - lp_mid(s) )   (table columns >= 2U+1 are -inf: dead states drop out)
            const unsigned long long rv = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(lp_u + (int64_t)mid * Sp + 2 * lane),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const f32x2 r = __builtin_bit_cast(f32x2, rv);
            const float v0 = (r[0] == -INFINITY) ? -INFINITY : e + xch[lane][0] - r[0];
            const float v1 = (r[1] == -INFINITY) ? -INFINITY : o + xch[lane][1] - r[1];
            const float m = wave_max(fmaxf(v0, v1));
            const float ms = fmaxf(m, NEG_BIG);
            const float sum = wave_sum(__builtin_amdgcn_exp2f(v0 - ms) + __builtin_amdgcn_exp2f(v1 - ms));
            if (lane == 0) nll[b] = -(ms + __builtin_amdgcn_logf(sum)) * LN2;
        }
    }
    // ---- the last utterance to finish closes the op: the batch mean (loss.py:41-43, what ctc_mean_kernel computes, same order) and the
    // op's own finished-count back to zero; the arrival counters went back to zero one by one as their chunks were seen complete (the
    // polling wavefronts), so a caller can keep ONE counter buffer per stream, zeroed once, and the op is one launch with no memset in
    // front and no reduction kernel behind.  (Round 6, measured and dropped: the last quarter / eighth of the walk handed out by tickets
    // from one agent-scope counter instead of equal static shares - 146 -> 171-174 us whatever the share; the pass workgroups' end times
    // stay 45 us apart either way: they differ in speed by where they run, not in share.)
    if (stamp && lane == 0) stamp[1] = (unsigned)__builtin_amdgcn_s_memrealtime();
    int last = 0;
    if (lane == 0) {
        __threadfence();                                  // nll[b] / tgt_len[b] before the count
        last = __hip_atomic_fetch_add(extra + 8, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == Bn - 1;
    }
    last = __shfl(last, 0, 64);
    if (!last) return;
    if (mean_loss) {
        float acc = 0.f;
        for (int i = lane; i < Bn; i += 64) {
            const float nl = __hip_atomic_load(nll + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int tl = __hip_atomic_load(tgt_len + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += nl / (float)max(tl, 1);
        }
        acc = wave_sum(acc);
        if (lane == 0) mean_loss[0] = acc / (float)Bn;
    }
    if (lane == 0) __hip_atomic_store(extra + 8, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (the queue heads reset themselves)
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
                                                       float* __restrict__ grad, int64_t ldg,
        const float* __restrict__ occ2 = nullptr, const float* __restrict__ lp_ext = nullptr, const float* __restrict__ nll = nullptr) {
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
            const int64_t oi = ((int64_t)b * (L + 2) + t) * Sfull + sidx;
            // occ2: the workspace holds raw alpha / beta rows (asr_ctc_loss_bwd with alpha2): the occupancy is formed here
            const float w = occ2 ? occupancy(occ[oi], occ2[oi], lp_ext[((int64_t)b * L + t) * Sfull + sidx], nll[b] * LOG2E) : occ[oi];
            atomicAdd(&corr[lab], w);
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
                     int Umax, float* alpha, float* nll, int chunk, int W, int last, float* alpha2 = nullptr) {
#define LAUNCH_MITM(NP_)                                                                                                              \
    hipLaunchKernelGGL((ctc_mitm_kernel<NP_, PHASE>), dim3(B), dim3(128), 0, s, lp_ext, in_len, targets, tgt_len, L, Umax, alpha, nll, \
                       chunk, W, last, alpha2)
    switch (ctc_np(Umax)) {
        case 1: LAUNCH_MITM(1); break;
        case 2: LAUNCH_MITM(2); break;
        case 4: LAUNCH_MITM(4); break;
        default: LAUNCH_MITM(8); break;
    }
#undef LAUNCH_MITM
    return 0;
}

}  // namespace

extern "C" int asr_ctc_workspace_stride(int Umax) { return ctc_row_stride(Umax); }

// chunk geometry of the fused forward (shared by asr_ctc_loss_fwd and asr_ctc_counter_words)
static inline void ctc_chunking(int L, int n_chunks, int& W, int& nc) {
    const int steps = L / 2 + 1;                           // alpha takes mid + 1 <= L/2 + 1 steps, beta at most as many
    W = ((steps + n_chunks - 1) / n_chunks + 7) / 8 * 8;   // (a multiple of every rows-per-workgroup choice, and even)
    nc = (steps + W - 1) / W;
}

// The fused form's recursion workgroups wait for pass workgroups of the same grid: they must never be able to occupy every slot
// the pass workgroups could run in.  Two of a CU's eight slots at most: B <= 2 x (CUs of the CURRENT device - a partition or a
// CU-masked run has fewer than 256).
static inline int ctc_fused_max_batch() {
    static int cached[64] = {0};      // per device ordinal (the attribute query is a driver call: once per device)
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;      // no device: the two-launch form
    if (dev >= 0 && dev < 64 && cached[dev] > 0) return 2 * cached[dev];
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) return 0;
    if (dev >= 0 && dev < 64) cached[dev] = n_cu;
    return 2 * n_cu;
}

extern "C" int64_t asr_ctc_counter_words(int B, int L, int n_chunks) {
    if (n_chunks <= 1 || L < 64 || B <= 0 || B > ctc_fused_max_batch()) return 0;
    int W, nc;
    ctc_chunking(L, n_chunks, W, nc);
    return (int64_t)B * 2 * nc + 16;      // arrival counters + the pass workgroups' queue heads (8) + the finished-utterance count
}

// ---- forward from finished table rows (asr_vocab_proj_ctc wrote them while the projection's logits passed through LDS): the two
// half-length alpha / beta chains and the batch mean, nothing else - no pass over logits at all
extern "C" int asr_ctc_loss_fwd_table(void* stream, const float* lp_ext, const int32_t* in_len, const int64_t* targets, int B, int L, int Umax,
                                      float* alpha, float* nll, int32_t* tgt_len, float* mean_loss) {
    ASR_REQUIRE(lp_ext && in_len && targets && alpha && nll && tgt_len, ASR_ERR_ARG, "ctc_fwd_table: null pointer");
    ASR_REQUIRE(B > 0 && L > 0 && Umax > 0, ASR_ERR_ARG, "ctc_fwd_table: bad sizes");
    ASR_REQUIRE(ctc_np(Umax) == 1, ASR_ERR_UNSUPPORTED, "ctc_fwd_table: Umax = %d (one state pair per lane: U + 1 <= 64)", Umax);
    ASR_REQUIRE(asr_aligned(lp_ext, 16) && asr_aligned(alpha, 16), ASR_ERR_ALIGN, "ctc_fwd_table: workspaces must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    launch_recursion<0>(s, lp_ext, in_len, targets, tgt_len, B, L, Umax, alpha, nll, 0, L, 1);
    if (mean_loss) hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, s, nll, tgt_len, B, mean_loss);
    ASR_LAUNCH_CHECK("ctc_loss_fwd_table");
    return 0;
}

static int ctc_loss_fwd_impl(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                             int L, int V, int Umax, int blank, float* lse, float* lp_ext, float* alpha, float* nll,
                             int32_t* tgt_len, void* zero_counters, int n_chunks, float* mean_loss) {
    ASR_REQUIRE(logits && in_len && targets && lse && lp_ext && alpha && nll && tgt_len, ASR_ERR_ARG, "ctc_fwd: null pointer");
    ASR_REQUIRE(B > 0 && L > 0 && V > 1 && Umax > 0 && blank >= 0 && blank < V && ldl >= V, ASR_ERR_ARG, "ctc_fwd: bad sizes");
    ASR_REQUIRE(Umax + 1 <= 512, ASR_ERR_UNSUPPORTED, "ctc_fwd: Umax=%d too long (U+1 must be <= 512)", Umax);
    ASR_REQUIRE(asr_aligned(lp_ext, 16) && asr_aligned(alpha, 16), ASR_ERR_ALIGN, "ctc_fwd: workspaces must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // The fused form's recursion workgroups (one per utterance, the grid's first blocks) WAIT for the pass workgroups behind them: they
    // must never be able to fill the chip on their own.  Up to 2 utterances per CU of the current device (2 of the 8 slots) they cannot;
    // larger batches take the two-launch form.
    if (n_chunks <= 1 || L < 64 || ctc_np(Umax) != 1 || B > ctc_fused_max_batch()) {   // two launches: one pass over the logits, then the two half-length chains
                                                          // (the fused form is written for one state pair per lane: U + 1 <= 64)
        hipLaunchKernelGGL(ctc_lse_gather_kernel, dim3(B * L), dim3(256), 0, s, logits, ldl, in_len, targets, L, V, Umax, blank, lse,
                           lp_ext);
        launch_recursion<0>(s, lp_ext, in_len, targets, tgt_len, B, L, Umax, alpha, nll, 0, L, 1);
        if (mean_loss) hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, s, nll, tgt_len, B, mean_loss);
        ASR_LAUNCH_CHECK("ctc_loss_fwd");
        return 0;
    }
    // fused: ONE launch - pass workgroups publish table rows chunk by chunk (write-through stores + arrival counters in row L+1 of
    // the alpha workspace), recursion workgroups of the same grid consume them as they arrive
    // ASR_AMD_CTC_DBG: timing / attribution builds of the launch (results invalid): bit 1 the labels gathered by a second global load
    // (round 4's form: FETCH_SIZE 705 MB against 562), bit 2 pass only, bit 3 no gather, bit 4 chains never wait, bit 5 chains only,
    // bit 7 workgroup end stamps (tools/ctc_timeline.py)
    constexpr int rpb = 4;          // table rows per pass workgroup and item (1 / 2 / 8: 151 / 147 / 154 us against 144)
    constexpr int ringp = 8;        // row pairs in the recursion's LDS ring (10: 23 KiB -> 6 workgroups per CU, 0.19 ms)
    static const int dbg = [] { const char* e = getenv("ASR_AMD_CTC_DBG"); return e ? atoi(e) : 0; }();
    int W, nc;
    ctc_chunking(L, n_chunks, W, nc);
    const int Sp = ctc_row_stride(Umax);
    // arrival counters: 2 * nc words per utterance - in a caller-zeroed buffer when one is handed in (the trainer's per-step zero
    // arena: no memset node in front of the launch), else in row L + 1 of the alpha workspace, zeroed here
    ASR_REQUIRE(2 * nc + 16 <= Sp, ASR_ERR_UNSUPPORTED, "ctc_fwd: too many chunks (%d) for the counter row", nc);
    int* arrivals = reinterpret_cast<int*>(zero_counters);
    int64_t arr_stride = 2 * nc;
    int* extra = arrivals ? arrivals + (int64_t)B * 2 * nc : nullptr;      // queue heads + finished count: behind the arrival counters,
    if (!arrivals) {                                                        // ... or behind utterance 0's in its workspace row
        arrivals = reinterpret_cast<int*>(alpha + (int64_t)(L + 1) * Sp);
        arr_stride = (int64_t)(L + 2) * Sp;
        extra = arrivals + 2 * nc;
        hipError_t e__ = hipMemset2DAsync(arrivals, (size_t)arr_stride * sizeof(float), 0, (size_t)(2 * nc + 16) * sizeof(int), B, s);
        if (e__ != hipSuccess) {
            asr_set_error("ctc_loss_fwd: %s", hipGetErrorString(e__));
            return (int)e__;
        }
    }
    // B recursion workgroups + persistent pass workgroups filling every remaining slot of the chip (7 or 8 per CU by the kernel's
    // LDS / register budget; more than fit would only queue)
    const int n_cu = ctc_fused_max_batch() / 2;      // (of the current device)
    const int groups = nc * B * 2 * (W / rpb);
    int npass = n_cu * 8 - B;      // 8 per CU: the kernel's LDS (ring of 8 row pairs, 18.6 KB) and register budget (<= 64)
    if (npass > groups) npass = groups;
    if (npass < 1) npass = 1;
    const int grid = B + npass;
    const int kdbg = (dbg & 0xff) | (rpb << 8);
    hipLaunchKernelGGL(ctc_fused_fwd_kernel<ringp>, dim3(grid), dim3(256), 0, s, logits, ldl, in_len, targets, B, L, V, Umax, blank, lse, lp_ext,
                       alpha, nll, tgt_len, W, nc, arrivals, arr_stride, extra, mean_loss, kdbg);
    ASR_LAUNCH_CHECK("ctc_loss_fwd");
    return 0;
}

extern "C" int asr_ctc_loss_fwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                int L, int V, int Umax, int blank, float* lse, float* lp_ext, float* alpha, float* nll,
                                int32_t* tgt_len, void* zero_counters, int n_chunks) {
    return ctc_loss_fwd_impl(stream, logits, ldl, in_len, targets, B, L, V, Umax, blank, lse, lp_ext, alpha, nll, tgt_len, zero_counters,
                             n_chunks, nullptr);
}

// asr_ctc_loss_fwd + asr_ctc_mean as ONE op: loss[0] = mean_b nll[b] / max(tgt_len[b], 1) comes out of the same launch (fused form: the
// last utterance to finish reduces the batch; otherwise the reduction kernel is queued here).
extern "C" int asr_ctc_loss_mean_fwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                     int L, int V, int Umax, int blank, float* lse, float* lp_ext, float* alpha, float* nll,
                                     int32_t* tgt_len, void* zero_counters, int n_chunks, float* loss) {
    ASR_REQUIRE(loss, ASR_ERR_ARG, "ctc_loss_mean_fwd: null loss pointer");
    return ctc_loss_fwd_impl(stream, logits, ldl, in_len, targets, B, L, V, Umax, blank, lse, lp_ext, alpha, nll, tgt_len, zero_counters,
                             n_chunks, loss);
}

// bf16 gradient (the trainer's path: the gradient is consumed by bf16 MFMA GEMMs only, which would round the f32 image on load to
// exactly these values - half the write here, half the read in both of ctc_fc's backward GEMMs).  Rows are 16-byte aligned on
// both sides (ldl % 4 == 0, ldg % 8 == 0, host-checked); columns V .. ldg-1 are written as zeros: ldg is chosen by the caller
// so that the GEMM kernels can treat the rows as padded to their tile width.
// LT = _Float16: the logits are the fp16 image asr_vocab_proj_ctc wrote (ldl % 8 == 0; half the read).
template <typename LT>
__global__ __launch_bounds__(256) void ctc_grad_bf16_kernel(const LT* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ in_len,
                                                            const int64_t* __restrict__ targets, const int32_t* __restrict__ tgt_len,
                                                            int B, int L, int V, int Umax, int blank, const float* __restrict__ lse,
                                                            const float* __restrict__ occ, const float* __restrict__ gout,
                                                            bf16_t* __restrict__ grad, int64_t ldg,
        const float* __restrict__ occ2 = nullptr, const float* __restrict__ lp_ext = nullptr, const float* __restrict__ nll = nullptr) {
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
        const LT* x = logits + row * ldl;
        for (int sidx = tid; sidx < Sb; sidx += 256) {
            const int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            const int64_t oi = ((int64_t)b * (L + 2) + t) * Sfull + sidx;
            // occ2: the workspace holds raw alpha / beta rows (asr_ctc_loss_bwd with alpha2): the occupancy is formed here
            const float w = occ2 ? occupancy(occ[oi], occ2[oi], lp_ext[((int64_t)b * L + t) * Sfull + sidx], nll[b] * LOG2E) : occ[oi];
            atomicAdd(&corr[lab], w);
        }
        __syncthreads();
        const float l = lse[row];
        typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
        typedef typename std::conditional<std::is_same<LT, float>::value, f32x4, f16x4>::type x4_t;
        const x4_t* x4 = reinterpret_cast<const x4_t*>(x);
        for (int i = tid; i < ng; i += 256) {
            const int c = i * 4;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            if (c + 4 <= V) {
                const x4_t xv = x4[i];
                const f32x4 v = {(float)xv[0], (float)xv[1], (float)xv[2], (float)xv[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = scale * (__expf(v[j] - l) - corr[c + j]);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (c + j < V) o[j] = scale * (__expf((float)x[c + j] - l) - corr[c + j]);
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

extern "C" int asr_ctc_loss_bwd_ex(void* stream, const void* logits, int logits_dtype, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                   int L, int V, int Umax, int blank, const float* lse, const float* lp_ext, float* alpha,
                                   const float* nll, const int32_t* tgt_len, const float* gout, void* grad, int grad_dtype, int64_t ldg,
                                   float* alpha2);
extern "C" int asr_ctc_loss_bwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                int L, int V, int Umax, int blank, const float* lse, const float* lp_ext, float* alpha,
                                const float* nll, const int32_t* tgt_len, const float* gout, void* grad, int grad_dtype, int64_t ldg,
                                float* alpha2) {
    return asr_ctc_loss_bwd_ex(stream, logits, ASR_F32, ldl, in_len, targets, B, L, V, Umax, blank, lse, lp_ext, alpha, nll, tgt_len, gout, grad,
                               grad_dtype, ldg, alpha2);
}

// logits_dtype = ASR_F16: the fp16 logits image of asr_vocab_proj_ctc (bf16 gradient only; ldl % 8 == 0).
extern "C" int asr_ctc_loss_bwd_ex(void* stream, const void* logits_v, int logits_dtype, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                   int L, int V, int Umax, int blank, const float* lse, const float* lp_ext, float* alpha,
                                   const float* nll, const int32_t* tgt_len, const float* gout, void* grad, int grad_dtype, int64_t ldg,
                                   float* alpha2) {
    const float* logits = static_cast<const float*>(logits_v);
    ASR_REQUIRE(logits_dtype == ASR_F32 || (logits_dtype == ASR_F16 && grad_dtype == ASR_BF16 && ldl % 8 == 0), ASR_ERR_ARG,
                "ctc_bwd: fp16 logits need a bf16 gradient and ldl %% 8 == 0");
    ASR_REQUIRE(logits && in_len && targets && lse && lp_ext && alpha && nll && tgt_len && gout && grad, ASR_ERR_ARG,
                "ctc_bwd: null pointer");
    ASR_REQUIRE(!alpha2 || asr_aligned(alpha2, 16), ASR_ERR_ALIGN, "ctc_bwd: alpha2 must be 16-byte aligned");
    ASR_REQUIRE(B > 0 && L > 0 && V > 1 && Umax > 0 && ldl >= V && ldg >= V, ASR_ERR_ARG, "ctc_bwd: bad sizes");
    ASR_REQUIRE(grad_dtype == ASR_F32 || grad_dtype == ASR_BF16, ASR_ERR_ARG, "ctc_bwd: bad grad_dtype");
    ASR_REQUIRE(grad_dtype == ASR_F32 || (ldg % 8 == 0 && ldl % 4 == 0 && asr_aligned(grad, 16) && asr_aligned(logits, 16)), ASR_ERR_ALIGN,
                "ctc_bwd: a bf16 gradient needs 16-byte aligned rows (ldg %% 8 == 0, ldl %% 4 == 0)");
    ASR_REQUIRE((size_t)(V + 4) * sizeof(float) <= 64 * 1024, ASR_ERR_UNSUPPORTED, "ctc_bwd: V=%d exceeds the LDS occupancy vector", V);
    ASR_REQUIRE(Umax + 1 <= 512, ASR_ERR_UNSUPPORTED, "ctc_bwd: Umax too long");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (alpha2)
        launch_recursion<2>(s, lp_ext, in_len, targets, const_cast<int32_t*>(tgt_len), B, L, Umax, alpha, const_cast<float*>(nll), 0, L, 1, alpha2);
    else
        launch_recursion<1>(s, lp_ext, in_len, targets, const_cast<int32_t*>(tgt_len), B, L, Umax, alpha, const_cast<float*>(nll), 0, L, 1);
    // ~2048 workgroups in flight (8 per CU: every wave slot).  In the training step this launch runs on the side stream beside the
    // decoder's small kernels, whose workgroups are fat - attention at 51 x 1000: eight waves of 162 registers + 128 KiB of LDS,
    // gemm_ln_small: eight waves of 178 - and find no CU to start on while this launch holds every wave slot (tools/op_timeline.py,
    // events, no profiler: cross attention 17.6 -> 154 us beside it).  Capping it at 6 / 5 / 4 workgroups per CU is not enough
    // (4 x 40 registers leave a SIMD 352, gemm_ln_small needs 368; and ctc_fc's two GEMMs behind it starve the same kernels), which
    // is what the 2.51-2.53 ms of those runs showed.  Under asr_launch_budget (trainer: 128): 3 workgroups per budgeted CU; with
    // the two GEMMs budgeted as well no decoder kernel stalls any more and the segment is 2.41 -> 2.32 ms - the branch's work
    // still shares the chip with the chain, it just no longer stops it.  (A software-pipelined form of the pass - next row's logits
    // and occupancies requested a row ahead, two correction vectors, two barriers per row - runs 546 -> 336 us under the budget, but at
    // 69 registers instead of 38 three workgroups per CU leave a SIMD 296 registers and gemm_ln_small stalls again: segment and step
    // unchanged; alone, at full grid, it is no faster with fp16 logits and 11 % slower with f32 ones.  Not kept.)
    const int budget = asr_launch_budget_current();      // (asr_hip.h: asr_launch_budget - 3 workgroups per budgeted CU)
    const int wgs = budget > 0 && budget * 3 < 2048 ? budget * 3 : 2048;
    int rb = (wgs + B - 1) / B;
    if (rb > L) rb = L;
    if (rb < 1) rb = 1;
    if (grad_dtype == ASR_BF16 && logits_dtype == ASR_F16)
        hipLaunchKernelGGL(ctc_grad_bf16_kernel<_Float16>, dim3(rb, B), dim3(256), (size_t)(V + 4) * sizeof(float), s, static_cast<const _Float16*>(logits_v),
                           ldl, in_len, targets, tgt_len, B, L, V, Umax, blank, lse, alpha, gout, reinterpret_cast<bf16_t*>(grad), ldg, alpha2, lp_ext, nll);
    else if (grad_dtype == ASR_BF16)
        hipLaunchKernelGGL(ctc_grad_bf16_kernel<float>, dim3(rb, B), dim3(256), (size_t)(V + 4) * sizeof(float), s, logits, ldl, in_len, targets,
                           tgt_len, B, L, V, Umax, blank, lse, alpha, gout, reinterpret_cast<bf16_t*>(grad), ldg, alpha2, lp_ext, nll);
    else
        hipLaunchKernelGGL(ctc_grad_kernel, dim3(rb, B), dim3(256), (size_t)V * sizeof(float), s, logits, ldl, in_len, targets, tgt_len, B, L,
                           V, Umax, blank, lse, alpha, gout, reinterpret_cast<float*>(grad), ldg, alpha2, lp_ext, nll);
    ASR_LAUNCH_CHECK("ctc_loss_bwd");
    return 0;
}
