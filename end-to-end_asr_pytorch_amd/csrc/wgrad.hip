// Weight gradient dW[N,K] (+)= dY[M,N]^T . X[M,K] for gfx950, bf16 operands, fp32 result - the split-M GEMM behind every nn.Linear's
// weight.grad in the reference's loss.backward() (solver.py:119; module.py:48-53, attention.py:33-60, transformer.py:148).
//
// What bounds this GEMM on MI355X is not the matrix pipe: the output is tiny (256 x 2048 fp32 = 2 MB) and the reduction dimension is
// the 32 000 frames, so 256 CUs can only be busy if ~8-64 workgroups work on the SAME output tile, and their partial tiles have to
// meet somewhere.  Measured (tools/probe/atomic_probe.hip): fp32 atomics retire at ~1.2 TB/s whichever XCDs touch which lines (they
// execute memory-side), plain 16-byte stores of the same partials at > 4 TB/s.  So:
//   * partials go to a slab [split][tile] with plain lane-linear 16-byte stores (fragment order - no transpose, 1 KiB per wave
//     instruction) and a second, tiny launch sums them in a FIXED order (4 interleaved split groups per element, each in split
//     order, then group 0..3) and writes / accumulates dW and the bias gradient.  No atomics on the result, no pre-zeroing, no
//     counters or fences, and a result that does not depend on timing: the kernel pair is deterministic.
//     (Tried in one launch: the last workgroup to arrive at a tile - or at each of 8 tile slices, with the slices' counters walked
//     in rotated order to spread the last arrivals - sums the partials.  __threadfence() per workgroup: 70 us of fence for a 25 us
//     GEMM; write-through stores + agent-scope loads instead: 49 us, of which ~12 us are the 8 dependent returning atomics and the
//     latency-bound sums of whoever arrives last.  The second launch costs ~3 us of stream time and reads the slab at full width.)
//   * one workgroup per CU: 4 waves = 2 column halves x 2 halves of each 64-row reduction step, every wave a 128 x 64 accumulator
//     (128 registers) fed by v_mfma_f32_32x32x16_bf16 - 12 transposing LDS reads per 8 MFMAs instead of the 16 per 8 of a 64 x 64
//     wave tile, which kept the LDS pipe as busy as the matrix pipe;
//   * operand tiles [64 m][128 cols] arrive row-major by LDS-DMA into a ring of four stages (requested three steps ahead, one
//     counted s_waitcnt vmcnt(8) + one barrier per step); both MFMA operands are ds_read_b64_tr_b16 fragments of them (8 consecutive
//     m for one output row / column);
//   * the instruction stream of a step is pinned: 16 slots of one MFMA + at most two transposing reads (for the fragment set one
//     half-step ahead) + at most one DMA request.
#include <stdlib.h>

#include "asr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct TnArgs {
    const bf16_t* A;      // dY [M, lda]
    const bf16_t* B;      // X  [M, ldb]
    float* C;             // dW [N, ldc]
    float* slab;          // [splits][tiles][TILE_F] partial tiles (fragment order) + their 128 column-sum partials
    float* colsum;        // [N] += column sums of dY, or null
    int64_t lda, ldb, ldc;
    int M, N, K, tiles_k, m_per_split, splits, accumulate, cs_all;
};
constexpr int TILE_F = 128 * 128 + 128;   // floats per slab tile: the partial tile + its column-sum partials

__device__ __forceinline__ int tr_sw(int row) { return ((row & 3) << 1) ^ (((row >> 3) & 1) << 3); }
__device__ __forceinline__ u32x2 tr8(const unsigned char* p) {
    const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    return __builtin_bit_cast(u32x2, v);
}
template <int I> struct IC { static constexpr int value = I; };

// bid / nblocks: this workgroup's index among the tiles x splits workgroups of ITS problem (the whole grid for a single launch, a slice
// of it in a grouped one)
__device__ __forceinline__ void tn_body(const TnArgs& a, const int bid, const int nblocks) {
    constexpr int STAGE = 16384;                        // one operand tile: 64 rows x 256 B
    __shared__ __attribute__((aligned(16))) unsigned char smem[8 * STAGE];   // A ring [4] | B ring [4]
    const int tiles = nblocks / a.splits;
    int tile, split;
    {
        // Workgroups go to the 8 XCDs round-robin (launched: nblocks rounded up to 8), and each XCD has its own L2: the (M-range, tile)
        // pairs are dealt to the XCDs in CONTIGUOUS runs of the M-range-major order - with a multiple of 8 M-ranges whole ranges per
        // XCD (every row of dY and X enters one L2 only), with fewer a run of one range's tiles, ordered by the block index of the
        // LARGER operand (column block of dY when N >= K, of X otherwise) so that its rows still enter one L2 only.  (Dealt by
        // workgroup index, 4 M-ranges were each fetched into all eight L2s: FETCH 272 MB per feed-forward weight gradient instead
        // of the 147.5 MB the operands hold; the slices of a grouped launch start at multiples of 8 for the same reason.)
        const int per = (nblocks + 7) >> 3;
        const int p = (bid & 7) * per + (bid >> 3);
        if (p >= nblocks) return;
        split = p / tiles;
        const int q = p - split * tiles;
        if (a.N >= a.K) tile = q;
        else {
            const int tn_count = tiles / a.tiles_k, tkq = q / tn_count;
            tile = (q - tkq * tn_count) * a.tiles_k + tkq;
        }
    }
    const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int m_begin = split * a.m_per_split, m_end = min(a.M, m_begin + a.m_per_split);
    const int nsteps = m_begin < m_end ? (m_end - m_begin + 63) >> 6 : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int th = wave & 1, mh = wave >> 1;
    const int i16 = lane & 15, hh = lane >> 5, colhalf = (lane >> 4) & 1;

    // ---- DMA side: this wave stages pieces 4w..4w+3 (4 rows x 256 B each) of both operand tiles
    const u32x4 ars = rsrc_words(a.A, (unsigned)((int64_t)a.M * a.lda * 2));
    const u32x4 brs = rsrc_words(a.B, (unsigned)((int64_t)a.M * a.ldb * 2));
    unsigned voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * (wave * 4 + i) + (lane >> 4);
        const int c = (lane & 15) ^ tr_sw(row);
        voffA[i] = (unsigned)(((int64_t)row * a.lda + n0 + c * 8) * 2);
        voffB[i] = (unsigned)(((int64_t)row * a.ldb + k0 + c * 8) * 2);
    }
    const unsigned smem0 = lds_addr_of(smem);
    const unsigned lda2 = (unsigned)a.lda * 2u, ldb2 = (unsigned)a.ldb * 2u;
    // (steps past the end of the range are requested at an offset beyond the buffer: no fetch, still counted by vmcnt)
    auto dma_step = [&](int i, int step, int stage) {       // request i (0..7) of the 8 this wave makes per step
        const unsigned m0 = (unsigned)(m_begin + step * 64);
        const bool live = step < nsteps;
        if (i < 4) dma16_asm(ars, voffA[i], live ? m0 * lda2 : 0x7ff00000u, smem0 + stage * STAGE + (wave * 4 + i) * 1024);
        else dma16_asm(brs, voffB[i - 4], live ? m0 * ldb2 : 0x7ff00000u, smem0 + (4 + stage) * STAGE + (wave * 4 + i - 4) * 1024);
    };

    // ---- fragment side: lane addresses of the transposing reads (see the file header of backward.hip for the rule)
    const int rowl = 32 * mh + 4 * hh + (i16 >> 2);
    const unsigned char* abase[4];
    const unsigned char* bbase0[2];
    const unsigned char* bbase1[2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        const int c = 4 * cb + 2 * colhalf + ((i16 & 3) >> 1);
        abase[cb] = smem + rowl * 256 + ((c ^ tr_sw(rowl)) << 4) + 8 * (i16 & 1);
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const int c = 4 * (2 * th + kb) + 2 * colhalf + ((i16 & 3) >> 1);
        bbase0[kb] = smem + 4 * STAGE + rowl * 256 + ((c ^ tr_sw(rowl)) << 4) + 8 * (i16 & 1);
        bbase1[kb] = bbase0[kb] + 2048 + (th ? -128 : 128);
    }
    // fragment f of set (stage S, half-step g): f = 0..3 the dY column blocks, 4..5 the X column blocks of this wave's half
    u32x4 F[2][6];
    auto frag_lo = [&](auto S, auto G, int f, u32x4& dst) {      // rows +0..3 (first transposing read of the pair)
        constexpr int off = decltype(S)::value * STAGE + decltype(G)::value * 4096;
        const u32x2 v = f < 4 ? tr8(abase[f] + off) : tr8(bbase0[f - 4] + off);
        dst[0] = v[0]; dst[1] = v[1];
    };
    auto frag_hi = [&](auto S, auto G, int f, u32x4& dst) {      // rows +8..11
        constexpr int off = decltype(S)::value * STAGE + decltype(G)::value * 4096;
        const u32x2 v = f < 4 ? tr8(abase[f] + off + 2048 + ((f & 2) ? -128 : 128)) : tr8(bbase1[f - 4] + off);
        dst[2] = v[0]; dst[3] = v[1];
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[cb][kb][i] = 0.f;
    // bias gradient for free: a product of the dY fragments with an all-ones operand holds their column sums in every column.  The
    // tiles_k workgroups that share a dY column block and an M-range take turns by step (cs_all: the tk == 0 workgroup takes all -
    // a single writer per element of colsum, the deterministic form); the two column halves of a workgroup take two blocks each.
    f32x16 cs[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { cs[0][i] = 0.f; cs[1][i] = 0.f; }
    const bool cs_on = a.colsum != nullptr;
    const bf16x8 ones = __builtin_bit_cast(bf16x8, u32x4{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    // prologue: stages 0..2 requested, fragment set (0, 0) read
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) dma_step(i, s, s);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int f = 0; f < 6; ++f) { frag_lo(IC<0>(), IC<0>(), f, F[0][f]); frag_hi(IC<0>(), IC<0>(), f, F[0][f]); }

    auto mma = [&](int slot, const u32x4 (&Fs)[6]) {
        const int cb = slot >> 1, kb = slot & 1;
        acc[cb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Fs[cb]), __builtin_bit_cast(bf16x8, Fs[4 + kb]),
                                                              acc[cb][kb], 0, 0, 0);
    };
    // (the wave's pair of dY fragments picked by SELECT, not by branch: with the two products written once per branch the compiler
    // merged the branches' accumulator tuples through 64 v_accvgpr_mov per call, in series with the MFMAs)
    auto colsum_mma = [&](const u32x4 (&Fs)[6]) {
        u32x4 f0, f1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f0[j] = th ? Fs[2][j] : Fs[0][j];
            f1[j] = th ? Fs[3][j] : Fs[1][j];
        }
        cs[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f0), ones, cs[0], 0, 0, 0);
        cs[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f1), ones, cs[1], 0, 0, 0);
    };
    auto step = [&](auto S, int t) {
        constexpr int S0 = decltype(S)::value, S1 = (S0 + 1) & 3, S3 = (S0 + 3) & 3;
        const bool cs_step = cs_on && (a.cs_all ? tk == 0 : (t % a.tiles_k) == tk);
        // half-step 0: MFMAs on F[0]; the reads of (S0, half-step 1) ride along
#pragma unroll
        for (int slot = 0; slot < 8; ++slot) {
            mma(slot, F[0]);
            if (slot < 6) { frag_lo(S, IC<1>(), slot, F[1][slot]); frag_hi(S, IC<1>(), slot, F[1][slot]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cs_step) colsum_mma(F[0]);
        // stage S1 has landed for everyone (this wave's requests for it are older than the 8 of the step after), and stage S3 - read
        // last during the previous step - is free
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int slot = 0; slot < 8; ++slot) {
            mma(slot, F[1]);
            if (slot < 6) { frag_lo(IC<S1>(), IC<0>(), slot, F[0][slot]); frag_hi(IC<S1>(), IC<0>(), slot, F[0][slot]); }
            dma_step(slot, t + 3, S3);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (cs_step) colsum_mma(F[1]);
    };
    for (int t = 0; t < nsteps; t += 4) {
        step(IC<0>(), t);
        if (t + 1 < nsteps) step(IC<1>(), t + 1);
        if (t + 2 < nsteps) step(IC<2>(), t + 2);
        if (t + 3 < nsteps) step(IC<3>(), t + 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- the two halves of the reduction steps meet: wave (th, mh) keeps column blocks 2 mh, 2 mh + 1 and hands the other two over
    f32x4* xch = reinterpret_cast<f32x4*>(smem);                 // [wave][blk 0..3][q 0..3][lane] 16 KiB per wave
    auto give = [&](auto G) {        // hands column blocks G, G + 1 to the wave of the other half
        constexpr int g0 = decltype(G)::value;
#pragma unroll
        for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x16& v = acc[g0 + cbl][kb];
                    xch[((wave * 4 + cbl * 2 + kb) * 4 + q) * 64 + lane] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
                }
    };
    if (mh) give(IC<0>()); else give(IC<2>());
    __syncthreads();
    f32x4 mine[2][2][4];                                         // [cbl][kb][q]: column block 2 mh + cbl
    auto take = [&](auto Kp) {
        constexpr int k0b = decltype(Kp)::value;
        const int other = wave ^ 2;
#pragma unroll
        for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x16& v = acc[k0b + cbl][kb];
                    const f32x4 o = xch[((other * 4 + cbl * 2 + kb) * 4 + q) * 64 + lane];
                    mine[cbl][kb][q] = f32x4{v[4 * q] + o[0], v[4 * q + 1] + o[1], v[4 * q + 2] + o[2], v[4 * q + 3] + o[3]};
                }
    };
    if (mh) take(IC<2>()); else take(IC<0>());
    // column sums: wave (th, mh) accumulated blocks 2 th, 2 th + 1 over ITS half of the rows; register i of a lane holds row
    // (i & 3) + 8 (i >> 2) + 4 hh of the block (every column the same).  Lanes 0 and 32 publish, the mh halves add up.
    float cs_tile = 0.f;     // after the exchange: thread tid < 128 holds the tile's column sum of dY column n0 + tid
    if (cs_on) {
        __syncthreads();
        float* cst = reinterpret_cast<float*>(smem + 5 * STAGE);     // [mh][128]
        if ((lane & 31) == 0) {
#pragma unroll
            for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
                for (int i = 0; i < 16; ++i) cst[mh * 128 + 32 * (2 * th + cbl) + (i & 3) + 8 * (i >> 2) + 4 * hh] = cs[cbl][i];
        }
        __syncthreads();
        if (tid < 128) cs_tile = cst[tid] + cst[128 + tid];
    }

    // element (cbl, kb, q, e) of `mine`: dW row n0 + 32 (2 mh + cbl) + e + 8 q + 4 hh, column k0 + 64 th + 32 kb + (lane & 31)
    auto write_c = [&](int w, int cbl, int kb, int q, int ln, const f32x4& v) {
        const int wth = w & 1, wmh = w >> 1;
        const int n = n0 + 32 * (2 * wmh + cbl) + 8 * q + 4 * (ln >> 5);
        const int k = k0 + 64 * wth + 32 * kb + (ln & 31);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < a.N) {
                float* p = a.C + (int64_t)(n + e) * a.ldc + k;
                *p = a.accumulate ? *p + v[e] : v[e];
            }
    };
    if (a.splits == 1) {
#pragma unroll
        for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int q = 0; q < 4; ++q) write_c(wave, cbl, kb, q, lane, mine[cbl][kb][q]);
        if (cs_on && tid < 128 && n0 + tid < a.N && (a.cs_all ? tk == 0 : true)) {
            if (a.cs_all) a.colsum[n0 + tid] += cs_tile;
            else atomicAdd(a.colsum + n0 + tid, cs_tile);
        }
        return;
    }

    // ---- partial tile -> slab (fragment order: f32x4 index ((wave * 2 + cbl) * 2 + kb) * 4 + q) * 64 + lane), summed by tn_reduce_kernel
    float* my = a.slab + ((int64_t)split * tiles + tile) * TILE_F;
#pragma unroll
    for (int cbl = 0; cbl < 2; ++cbl)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                reinterpret_cast<f32x4*>(my)[(((wave * 2 + cbl) * 2 + kb) * 4 + q) * 64 + lane] = mine[cbl][kb][q];
    if (cs_on && tid < 128) my[16384 + tid] = cs_tile;
}

__global__ __launch_bounds__(256, 1) void gemm_tn_v2_kernel(TnArgs a, int nblocks) { tn_body(a, blockIdx.x, nblocks); }

// Several weight gradients in one launch (the decoder's: 1632 rows each, a handful of tiles, 13-17 us apiece as single launches of
// which 2 us are work): problem i owns the workgroups first[i] .. first[i+1] - 1.
constexpr int TN_GROUP_MAX = 16;
struct TnGroup {
    int n;
    int first[TN_GROUP_MAX + 1];       // main launch: first workgroup of each problem (multiples of 8)
    int count[TN_GROUP_MAX];           // ... and how many of its slice are real (tiles x splits)
    int first_r[TN_GROUP_MAX + 1];     // reduce launch (tiles x 64 workgroups per problem with > 1 split, none otherwise)
    TnArgs p[TN_GROUP_MAX];
};
__global__ __launch_bounds__(256, 1) void gemm_tn_v2_group_kernel(TnGroup g) {
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.first[i + 1]) ++i;
    tn_body(g.p[i], (int)blockIdx.x - g.first[i], g.count[i]);
}

// dW tile = sum over the splits of the slab's partial tiles.  A workgroup owns 64 consecutive f32x4 of one tile (1 KiB: 4 output rows
// x 2 x 32 columns); its 4 waves take the splits s = g, g + 4, ... (up to 16 loads in flight per lane), LDS adds the four group sums in
// group order.  The first workgroup of a tile also sums the 128 column-sum partials.
__device__ __forceinline__ void tn_reduce_body(const TnArgs& a, const int tiles, const int bid) {
    __shared__ f32x4 part[3][64];
    const int tile = bid >> 6, blk = bid & 63;
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int64_t sstride = (int64_t)tiles * (TILE_F / 4);
    const f32x4* src = reinterpret_cast<const f32x4*>(a.slab + (int64_t)tile * TILE_F) + blk * 64 + lane;
    f32x4 s = {0, 0, 0, 0};
    int sp = g;
    for (; sp + 28 < a.splits; sp += 32) {
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(sp + 4 * u) * sstride];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; sp < a.splits; sp += 4) s += src[sp * sstride];
    if (g) part[g - 1][lane] = s;
    __syncthreads();
    if (g == 0) {
        s += part[0][lane];
        s += part[1][lane];
        s += part[2][lane];
        // f32x4 index blk * 64 + lane = ((w * 2 + cbl) * 2 + kb) * 4 + q) * 64 + lane of the producing workgroup
        const int q = blk & 3, kb = (blk >> 2) & 1, cbl = (blk >> 3) & 1, w = blk >> 4;
        const int n = n0 + 32 * (2 * (w >> 1) + cbl) + 8 * q + 4 * (lane >> 5);
        const int k = k0 + 64 * (w & 1) + 32 * kb + (lane & 31);
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (n + e < a.N) {
                float* p = a.C + (int64_t)(n + e) * a.ldc + k;
                *p = a.accumulate ? *p + s[e] : s[e];
            }
    } else if (blk == 0 && a.colsum && threadIdx.x >= 64 && threadIdx.x < 192 && (!a.cs_all || tk == 0)) {
        const int col = threadIdx.x - 64;
        float c = 0.f;
        for (int s2 = 0; s2 < a.splits; ++s2) c += a.slab[((int64_t)s2 * tiles + tile) * TILE_F + 16384 + col];
        if (n0 + col < a.N) {
            if (a.cs_all) a.colsum[n0 + col] += c;
            else atomicAdd(a.colsum + n0 + col, c);
        }
    }
}

__global__ __launch_bounds__(256) void tn_reduce_kernel(TnArgs a, int tiles) { tn_reduce_body(a, tiles, blockIdx.x); }
__global__ __launch_bounds__(256) void tn_reduce_group_kernel(TnGroup g) {
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.first_r[i + 1]) ++i;
    tn_reduce_body(g.p[i], (g.first_r[i + 1] - g.first_r[i]) >> 6, (int)blockIdx.x - g.first_r[i]);
}

}  // namespace

extern "C" int asr_gemm_tn(void* stream, const void* A, int a_dtype, int64_t lda, const void* Bm, int b_dtype, int64_t ldb, float* C,
                           int64_t ldc, int M, int N, int K, int zero_first, float* colsum, int max_workgroups);

// the M-split this kernel would use: splits x tiles ~ one workgroup per CU, ranges of whole 64-row steps, at least `min_rows` rows each
static void tn_v2_plan(int M, int N, int K, int max_workgroups, int* tiles_out, int* splits_out, int* mps_out) {
    constexpr int min_rows = 512;
    const int tiles = ((N + 127) / 128) * (K / 128);
    int target = 256;
    if (max_workgroups > 0 && max_workgroups < target) target = max_workgroups;
    int splits = (target + tiles / 2) / tiles;
    const int max_splits = (M + min_rows - 1) / min_rows;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    // (one launch without a slab for the decoder's 1632 rows - splits = 1, 26 serial steps per workgroup - measured slower than
    // 4 splits + the reduce launch: 17.1-17.6 vs 12.7-16.3 us)
    if (splits >= 8) splits = splits / 8 * 8;
    const int mps = ((M + splits - 1) / splits + 63) / 64 * 64;
    if (splits < 8) splits = (M + mps - 1) / mps;
    *tiles_out = tiles; *splits_out = splits; *mps_out = mps;
}

extern "C" int64_t asr_gemm_tn_ws_bytes(int M, int N, int K, int max_workgroups) {
    if (M <= 0 || N <= 0 || K <= 0 || K % 128) return 0;
    int tiles, splits, mps;
    tn_v2_plan(M, N, K, max_workgroups, &tiles, &splits, &mps);
    return 16 + (int64_t)splits * tiles * TILE_F * 4;
}

extern "C" int asr_gemm_tn_ws(void* stream, const void* A, int64_t lda, const void* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N,
                              int K, int accumulate, float* colsum, int max_workgroups, void* workspace, int64_t workspace_bytes,
                              int deterministic) {
    ASR_REQUIRE(A && Bm && C && M > 0 && N > 0 && K > 0, ASR_ERR_ARG, "gemm_tn_ws: bad args");
    const bool n_ok = N % 128 == 0 || lda >= (int64_t)(N + 127) / 128 * 128;
    const bool fits = (int64_t)M * lda * 2 < (1ll << 31) && (int64_t)M * ldb * 2 < (1ll << 31);
    if (!workspace || !asr_aligned(workspace, 16) || M < 64 || !n_ok || K % 128 || lda % 8 || ldb % 8 || !asr_aligned(A, 16) ||
        !asr_aligned(Bm, 16) || !fits || workspace_bytes < asr_gemm_tn_ws_bytes(M, N, K, max_workgroups))
        return asr_gemm_tn(stream, A, ASR_BF16, lda, Bm, ASR_BF16, ldb, C, ldc, M, N, K, accumulate ? 0 : 1, colsum, max_workgroups);
    TnArgs a;
    int tiles;
    tn_v2_plan(M, N, K, max_workgroups, &tiles, &a.splits, &a.m_per_split);
    a.A = (const bf16_t*)A; a.B = (const bf16_t*)Bm; a.C = C; a.colsum = colsum;
    a.slab = reinterpret_cast<float*>(workspace);
    a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.tiles_k = K / 128;
    a.accumulate = accumulate; a.cs_all = (deterministic || asr_deterministic()) ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gemm_tn_v2_kernel, dim3((tiles * a.splits + 7) / 8 * 8), dim3(256), 0, s, a, tiles * a.splits);
    ASR_LAUNCH_CHECK("gemm_tn_v2");
    if (a.splits > 1) {
        hipLaunchKernelGGL(tn_reduce_kernel, dim3(tiles * 64), dim3(256), 0, s, a, tiles);
        ASR_LAUNCH_CHECK("gemm_tn_reduce");
    }
    return 0;
}

// Up to 8 weight gradients in one pair of launches.  Every problem must be one asr_gemm_tn_ws itself takes (the same conditions; a
// group with a problem that is not is refused with ASR_ERR_UNSUPPORTED - the caller issues them one by one then); each brings its own
// workspace of asr_gemm_tn_ws_bytes(M, N, K, 0).  The chip's ~256 workgroup slots are shared out by output tiles.
extern "C" int asr_gemm_tn_ws_group_wgs(void* stream, int n, const asr_tn_problem_t* pr, int deterministic, int group_workgroups);
extern "C" int asr_gemm_tn_ws_group(void* stream, int n, const asr_tn_problem_t* pr, int deterministic) {
    return asr_gemm_tn_ws_group_wgs(stream, n, pr, deterministic, 0);
}

// group_workgroups: the launch's workgroup budget, shared out by output tiles (0: 128 = half the CUs for
// a launch beside the main chain).  A problem never gets fewer workgroups than it has tiles: with budget <= total tiles every
// problem runs UNSPLIT - one workgroup walks all M rows of its tile and writes dW itself, no slab, no reduce launch (the batched form
// of a whole encoder layer's, or two layers', weight gradients: modules._wg).
extern "C" int asr_gemm_tn_ws_group_wgs(void* stream, int n, const asr_tn_problem_t* pr, int deterministic, int group_workgroups) {
    ASR_REQUIRE(pr && n >= 1 && n <= TN_GROUP_MAX, ASR_ERR_ARG, "gemm_tn_ws_group: 1..%d problems", TN_GROUP_MAX);
    TnGroup g;
    g.n = n;
    int total_tiles = 0;
    for (int i = 0; i < n; ++i) {
        const asr_tn_problem_t& q = pr[i];
        ASR_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0, ASR_ERR_ARG, "gemm_tn_ws_group: problem %d: bad args", i);
        const bool n_ok = q.N % 128 == 0 || q.lda >= (int64_t)(q.N + 127) / 128 * 128;
        const bool fits = (int64_t)q.M * q.lda * 2 < (1ll << 31) && (int64_t)q.M * q.ldb * 2 < (1ll << 31);
        ASR_REQUIRE(q.workspace && asr_aligned(q.workspace, 16) && q.M >= 64 && n_ok && q.K % 128 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 &&
                        asr_aligned(q.A, 16) && asr_aligned(q.B, 16) && fits && q.workspace_bytes >= asr_gemm_tn_ws_bytes(q.M, q.N, q.K, 0),
                    ASR_ERR_UNSUPPORTED, "gemm_tn_ws_group: problem %d is not one the slab kernel takes", i);
        total_tiles += ((q.N + 127) / 128) * (q.K / 128);
    }
    g.first[0] = 0;
    g.first_r[0] = 0;
    for (int i = 0; i < n; ++i) {
        const asr_tn_problem_t& q = pr[i];
        TnArgs& a = g.p[i];
        int tiles;
        const int my_tiles = ((q.N + 127) / 128) * (q.K / 128);
        // the grouped launch runs on the trainer's side stream beside the main chain: half the CUs (a slab
        // workgroup takes a whole CU, see modules._WGRAD_SIDE_WGS)
        const int group_wgs = group_workgroups > 0 ? group_workgroups : 128;      // (96 / 160 / 192 / 256 / 64 on the side stream: S1 step +0.19 / -0.03 / -0.03 / +0.06 / +0.29 ms)
        int share = (int)((int64_t)group_wgs * my_tiles / total_tiles);
        if (share < my_tiles) share = my_tiles;
        tn_v2_plan(q.M, q.N, q.K, share, &tiles, &a.splits, &a.m_per_split);
        a.A = (const bf16_t*)q.A; a.B = (const bf16_t*)q.B; a.C = q.C; a.colsum = q.colsum;
        a.slab = reinterpret_cast<float*>(q.workspace);
        a.lda = q.lda; a.ldb = q.ldb; a.ldc = q.ldc; a.M = q.M; a.N = q.N; a.K = q.K; a.tiles_k = q.K / 128;
        a.accumulate = q.accumulate; a.cs_all = (deterministic || asr_deterministic()) ? 1 : 0;
        g.count[i] = tiles * a.splits;
        g.first[i + 1] = g.first[i] + (tiles * a.splits + 7) / 8 * 8;      // (a slice starts on XCD 0: tn_body's mapping)
        g.first_r[i + 1] = g.first_r[i] + (a.splits > 1 ? tiles * 64 : 0);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gemm_tn_v2_group_kernel, dim3(g.first[n]), dim3(256), 0, s, g);
    ASR_LAUNCH_CHECK("gemm_tn_v2_group");
    if (g.first_r[n] > 0) {
        hipLaunchKernelGGL(tn_reduce_group_kernel, dim3(g.first_r[n]), dim3(256), 0, s, g);
        ASR_LAUNCH_CHECK("gemm_tn_reduce_group");
    }
    return 0;
}
