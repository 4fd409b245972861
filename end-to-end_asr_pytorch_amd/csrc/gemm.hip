// MFMA projection GEMM for gfx950:  C[M,N] = epi(A[M,K] . W[N,K]^T)   ("NT": both operands K-contiguous).
//
// Tile 128x128, K-step = 128 bytes of K per LDS row (64 bf16 / 32 f32), 4 waves as 2(M) x 2(N), each wave a 64x64
// sub-tile = 4x4 MFMA 16x16 accumulators (v_mfma_f32_16x16x32_bf16, or the exact v_mfma_f32_16x16x4_f32 for the
// fp32 parity mode).  LDS rows are XOR-swizzled on the 16-byte chunk index (chunk ^ (row & 7)) so the ds_read_b128
// fragment reads (16 rows x one chunk per lane group) are bank-conflict free.  Global->LDS goes through registers
// (loads of tile k+1 are issued before the MFMAs of tile k) because M/N/K tails need guards and the f32->bf16
// conversion of activations happens on the way in.  blockIdx is remapped so each XCD walks a contiguous run of
// tiles (A row-panels are re-read from that XCD's L2 across the N tiles).
#include <stdlib.h>

#include "asr_common.h"
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, ROWB = 128, NT = 256;

template <typename CT> struct KTraits;
template <> struct KTraits<bf16_t> { static constexpr int KT = 64, CH = 8; };
template <> struct KTraits<float> { static constexpr int KT = 32, CH = 4; };

// raw registers for one 16-byte LDS chunk's worth of source data
template <typename AT, typename CT> struct Chunk;
template <typename T> struct Chunk<T, T> {
    u32x4 v;
    __device__ __forceinline__ void load(const T* p, bool ok) {
        v = ok ? *reinterpret_cast<const u32x4*>(p) : u32x4{0, 0, 0, 0};
    }
    __device__ __forceinline__ u32x4 get() const { return v; }
};
template <> struct Chunk<float, bf16_t> {
    f32x4 lo, hi;
    __device__ __forceinline__ void load(const float* p, bool ok) {
        if (ok) {
            lo = *reinterpret_cast<const f32x4*>(p);
            hi = *reinterpret_cast<const f32x4*>(p + 4);
        } else {
            lo = f32x4{0, 0, 0, 0};
            hi = lo;
        }
    }
    __device__ __forceinline__ u32x4 get() const {
        bf16x8 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r[i] = (bf16_t)lo[i];
            r[4 + i] = (bf16_t)hi[i];
        }
        return __builtin_bit_cast(u32x4, r);
    }
};

// ---- epilogues: receive 4 consecutive output columns n0..n0+3 of row m --------------------------------------
struct EpiDense {
    void* C;
    int c_dtype;
    int64_t ldc;
    const float* bias;
    unsigned flags;
    int M, N;
    const float* addend;      // optional f32 [M,N] (ld_add): C += addend  (residual-gradient accumulation)
    int64_t ld_add;
    const bf16_t* relu_mask;  // optional bf16 [M,N] (ld_mask): C = relu_mask > 0 ? C : 0  (ReLU backward)
    int64_t ld_mask;
    bool vec_ok;              // host-checked: every pointer / leading dimension allows 4-wide vector access
    bool wide_ok;             // ... and C rows allow 16-byte stores (the LDS-transposed full-cache-line epilogue)
    // ReLU sign bits (wide path only; layout: bits_slot below).  The FFN's first GEMM writes them next to its bf16 output; the hidden
    // gradient's GEMM then masks from 1 bit instead of re-reading 16
    unsigned char* bits_out;
    const unsigned char* bits_in;
    int64_t ld_bits;
    bool atomic_out;          // split-K (persistent kernels, f32 C, wide path): partial tiles are float-atomically added into a zeroed C
    bool c_is_zero;           // host-side only: the caller hands over a C that is already all zeros (no zeroing launch before a split-K)

    // fast path (kernel-uniform): the tile lies fully inside N and everything is vector-aligned -> no per-element logic
    struct Row { unsigned char* c; const float* add; const bf16_t* mask; };
    __device__ __forceinline__ bool fast(int n0_tile) const { return vec_ok && n0_tile + BN <= N; }
    __device__ __forceinline__ Row row(int m) const {
        return Row{reinterpret_cast<unsigned char*>(C) + (int64_t)m * ldc * (c_dtype == ASR_F32 ? 4 : 2),
                   addend ? addend + (int64_t)m * ld_add : nullptr, relu_mask ? relu_mask + (int64_t)m * ld_mask : nullptr};
    }
    // wide path (run_epilogue): math on the accumulator layout, stores from the LDS-transposed image
    __device__ __forceinline__ bool wide(int n0_tile) const { return wide_ok && n0_tile + BN <= N; }
    __device__ __forceinline__ int elem_size() const { return c_dtype == ASR_F32 ? 4 : 2; }
    __device__ __forceinline__ unsigned char* row_ptr(int m, int nw) const {
        return reinterpret_cast<unsigned char*>(C) + ((int64_t)m * ldc + nw) * (c_dtype == ASR_F32 ? 4 : 2);
    }
    __device__ __forceinline__ f32x4 apply(const Row& r, int n, f32x4 v) const {
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        if (flags & ASR_GEMM_RELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (r.add) v += *reinterpret_cast<const f32x4*>(r.add + n);
        if (r.mask) {
            const bf16x4 mk = *reinterpret_cast<const bf16x4*>(r.mask + n);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = ((float)mk[i] > 0.f) ? v[i] : 0.f;
        }
        return v;
    }
    // the wide epilogue issues every global read of a pass as one batch (kernel-uniform branches around whole batches) and
    // then only does arithmetic: with per-element `if (bias) load` the loads serialised, ~0.2 us each, 3 us per output tile
    // row cursor of the wide epilogue's read-back: the pointer of the first row once, then a kernel-uniform stride per step
    // (recomputing m * ldc per store cost two 64-bit multiplies = eight quarter-rate VALU ops per 16-byte store)
    struct Cur { unsigned char* p; };
    __device__ __forceinline__ void step(Cur& c, int rows) const { c.p += (int64_t)rows * ldc * (c_dtype == ASR_F32 ? 4 : 2); }
    // sign-bit image: per 64 x 64 wave sub-tile (mw / 64, nw / 64) 64 lanes x 8 bytes, lane = 8 (row & 7) + (col / 8 & 7), byte it =
    // rows 8 it + (row & 7): exactly what a lane of the read-back phase produces / consumes, as ONE 8-byte access per sub-tile
    __device__ __forceinline__ int64_t bits_slot(int mw, int nw, int lane) const { return (((int64_t)(mw >> 6) * (N >> 6) + (nw >> 6)) * 64 + lane) * 8; }
    __device__ __forceinline__ const unsigned char* bits_in_ptr(int mw, int nw, int lane) const { return bits_in + bits_slot(mw, nw, lane); }
    __device__ __forceinline__ unsigned char* bits_out_ptr(int mw, int nw, int lane) const { return bits_out + bits_slot(mw, nw, lane); }
    __device__ __forceinline__ bool has_bias() const { return bias != nullptr; }
    __device__ __forceinline__ bool has_add() const { return addend != nullptr; }
    __device__ __forceinline__ bool has_mask() const { return relu_mask != nullptr; }
    __device__ __forceinline__ bool has_bits_in() const { return bits_in != nullptr; }
    __device__ __forceinline__ bool has_bits_out() const { return bits_out != nullptr; }
    __device__ __forceinline__ Cur cur(int m, int nw) const { return Cur{row_ptr(m, nw)}; }
    __device__ __forceinline__ bool relu() const { return flags & ASR_GEMM_RELU; }
    __device__ __forceinline__ bool atomic() const { return atomic_out; }
    __device__ __forceinline__ f32x4 ld_bias(int n) const { return *reinterpret_cast<const f32x4*>(bias + n); }
    __device__ __forceinline__ f32x4 get_add(int m, int n) const { return *reinterpret_cast<const f32x4*>(addend + (int64_t)m * ld_add + n); }
    __device__ __forceinline__ bf16x4 get_mask(int m, int n) const { return *reinterpret_cast<const bf16x4*>(relu_mask + (int64_t)m * ld_mask + n); }
    __device__ __forceinline__ f32x4 post(int, f32x4 v) const { return v; }
    __device__ __forceinline__ void store_fast(const Row& r, int n, f32x4 v) const {
        v = apply(r, n, v);
        if (c_dtype == ASR_F32) {
            *reinterpret_cast<f32x4*>(r.c + (int64_t)n * 4) = v;
        } else {
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(r.c + (int64_t)n * 2) = o;
        }
    }
    __device__ __forceinline__ void store4(int m, int n0, f32x4 v) const {
        if (m >= M || n0 >= N) return;
        const int nv = min(4, N - n0);
        if (bias) {
            if (nv == 4 && ((reinterpret_cast<uintptr_t>(bias + n0) & 15) == 0)) {
                v += *reinterpret_cast<const f32x4*>(bias + n0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nv) v[i] += bias[n0 + i];
            }
        }
        if (flags & ASR_GEMM_RELU) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.f);
        }
        if (addend) {
            const float* ap = addend + (int64_t)m * ld_add + n0;
            if (nv == 4 && ((reinterpret_cast<uintptr_t>(ap) & 15) == 0)) {
                v += *reinterpret_cast<const f32x4*>(ap);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nv) v[i] += ap[i];
            }
        }
        if (relu_mask) {
            const bf16_t* mp = relu_mask + (int64_t)m * ld_mask + n0;
            if (nv == 4 && ((reinterpret_cast<uintptr_t>(mp) & 7) == 0)) {
                const bf16x4 mk = *reinterpret_cast<const bf16x4*>(mp);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = ((float)mk[i] > 0.f) ? v[i] : 0.f;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nv) v[i] = ((float)mp[i] > 0.f) ? v[i] : 0.f;
            }
        }
        const int64_t off = (int64_t)m * ldc + n0;
        if (c_dtype == ASR_F32) {
            float* p = reinterpret_cast<float*>(C) + off;
            if (nv == 4 && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
                *reinterpret_cast<f32x4*>(p) = v;
            } else if (nv == 4 && ((reinterpret_cast<uintptr_t>(p) & 7) == 0)) {
                *reinterpret_cast<f32x2*>(p) = f32x2{v[0], v[1]};
                *reinterpret_cast<f32x2*>(p + 2) = f32x2{v[2], v[3]};
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nv) p[i] = v[i];
            }
        } else {
            bf16_t* p = reinterpret_cast<bf16_t*>(C) + off;
            if (nv == 4 && ((reinterpret_cast<uintptr_t>(p) & 7) == 0)) {
                bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                *reinterpret_cast<bf16x4*>(p) = o;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nv) p[i] = (bf16_t)v[i];
            }
        }
    }
};

// EpiDense with its options fixed at compile time (MODE bits: 1 bias, 2 ReLU, 4 addend, 8 relu_mask, 16 bf16 output): the wide
// epilogue then is straight-line code.  With the options tested at run time every output fragment carried four scalar
// branches, and the epilogue took 1.8 us of a 4.9 us output tile.
template <unsigned MODE> struct EpiDenseS : EpiDense {
    __device__ __forceinline__ bool has_bias() const { return MODE & 1u; }
    __device__ __forceinline__ bool relu() const { return MODE & 2u; }
    __device__ __forceinline__ bool has_add() const { return MODE & 4u; }
    __device__ __forceinline__ bool has_mask() const { return MODE & 8u; }
    __device__ __forceinline__ bool has_bits_out() const { return MODE & 32u; }
    __device__ __forceinline__ bool has_bits_in() const { return MODE & 64u; }
    __device__ __forceinline__ bool atomic() const { return MODE & 128u; }
    __device__ __forceinline__ int elem_size() const { return (MODE & 16u) ? 2 : 4; }
    __device__ __forceinline__ unsigned char* row_ptr(int m, int nw) const {
        return reinterpret_cast<unsigned char*>(C) + ((int64_t)m * ldc + nw) * ((MODE & 16u) ? 2 : 4);
    }
    __device__ __forceinline__ EpiDense::Cur cur(int m, int nw) const { return EpiDense::Cur{row_ptr(m, nw)}; }
    __device__ __forceinline__ void step(EpiDense::Cur& c, int rows) const { c.p += (int64_t)rows * ldc * ((MODE & 16u) ? 2 : 4); }
};
inline unsigned dense_mode(const EpiDense& e) {
    return (e.bias ? 1u : 0u) | ((e.flags & ASR_GEMM_RELU) ? 2u : 0u) | (e.addend ? 4u : 0u) | (e.relu_mask ? 8u : 0u) |
           (e.c_dtype == ASR_BF16 ? 16u : 0u) | (e.bits_out ? 32u : 0u) | (e.bits_in ? 64u : 0u) | (e.atomic_out ? 128u : 0u);
}
template <unsigned MODE> __host__ __device__ inline EpiDenseS<MODE> dense_as(const EpiDense& e) {
    EpiDenseS<MODE> r;
    static_cast<EpiDense&>(r) = e;
    return r;
}

template <typename CT> struct EpiHeads {
    CT* out;
    int64_t proj_stride;
    const float* bias;
    int L, h, M, N;
    float scale_first;
    // fast path: N is a multiple of 64 and everything is aligned by construction, so every full tile qualifies.  The row
    // (b, t) split costs one integer division per ROW instead of per store, the head split one per 64-column wave slice.
    struct Row { CT* base; };
    __device__ __forceinline__ bool fast(int n0_tile) const { return n0_tile + BN <= N; }
    __device__ __forceinline__ Row row(int m) const {
        const int b = m / L, t = m - b * L;
        return Row{out + ((int64_t)b * h * L + t) * 64};
    }
    __device__ __forceinline__ bool wide(int n0_tile) const { return n0_tile + BN <= N; }
    __device__ __forceinline__ int elem_size() const { return (int)sizeof(CT); }
    __device__ __forceinline__ unsigned char* row_ptr(int m, int nw) const {   // nw: a multiple of 64 = one head's 64 columns
        const int b = m / L, t = m - b * L;
        const int slot = nw >> 6, which = slot / h, head = slot - which * h;
        return reinterpret_cast<unsigned char*>(out + which * proj_stride + (((int64_t)b * h + head) * L + t) * 64);
    }
    __device__ __forceinline__ bool has_bias() const { return bias != nullptr; }
    __device__ __forceinline__ bool has_add() const { return false; }
    __device__ __forceinline__ bool has_mask() const { return false; }
    __device__ __forceinline__ bool has_bits_in() const { return false; }
    __device__ __forceinline__ bool has_bits_out() const { return false; }
    // row cursor: (b, t) split by one division for the first row, then t += rows with a wrap into the next utterance's block
    struct Cur { unsigned char* p; int t; };
    __device__ __forceinline__ Cur cur(int m, int nw) const {
        const int b = m / L;
        return Cur{row_ptr(m, nw), m - b * L};
    }
    __device__ __forceinline__ void step(Cur& c, int rows) const {
        c.p += (int64_t)rows * 64 * (int)sizeof(CT);
        c.t += rows;
        while (c.t >= L) {
            c.t -= L;
            c.p += (int64_t)(h - 1) * L * 64 * (int)sizeof(CT);
        }
    }
    __device__ __forceinline__ const unsigned char* bits_in_ptr(int, int, int) const { return nullptr; }
    __device__ __forceinline__ unsigned char* bits_out_ptr(int, int, int) const { return nullptr; }
    __device__ __forceinline__ bool relu() const { return false; }
    __device__ __forceinline__ bool atomic() const { return false; }
    __device__ __forceinline__ f32x4 ld_bias(int n) const { return *reinterpret_cast<const f32x4*>(bias + n); }
    __device__ __forceinline__ f32x4 get_add(int, int) const { return f32x4{0, 0, 0, 0}; }
    __device__ __forceinline__ bf16x4 get_mask(int, int) const { return bf16x4{}; }
    __device__ __forceinline__ f32x4 post(int n, f32x4 v) const { return ((n >> 6) / h == 0) ? v * scale_first : v; }
    __device__ __forceinline__ void store_fast(const Row& r, int n, f32x4 v) const {
        if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
        const int slot = n >> 6;                       // global head slot = which * h + head
        const int which = slot / h, head = slot - which * h;
        if (which == 0) v *= scale_first;
        CT* p = r.base + which * proj_stride + (int64_t)head * L * 64 + (n & 63);
        if constexpr (sizeof(CT) == 4) {
            *reinterpret_cast<f32x4*>(p) = v;
        } else {
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(p) = o;
        }
    }
    __device__ __forceinline__ void store4(int m, int n0, f32x4 v) const {
        if (m >= M || n0 >= N) return;  // N is a multiple of 64
        if (bias) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n0);
            v += bv;
        }
        const int hd = h * 64;
        const int which = n0 / hd;
        const int rem = n0 - which * hd;
        const int head = rem >> 6, d = rem & 63;
        if (which == 0) v *= scale_first;
        const int b = m / L, t = m - b * L;
        CT* p = out + which * proj_stride + (((int64_t)b * h + head) * L + t) * 64 + d;
        if constexpr (sizeof(CT) == 4) {
            *reinterpret_cast<f32x4*>(p) = v;
        } else {
            bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            *reinterpret_cast<bf16x4*>(p) = o;
        }
    }
};

// Epilogue.  The MFMA accumulator layout gives a lane 4 consecutive output columns of ONE row, so direct stores write 16 rows x
// 32-byte (bf16) / 64-byte (f32) fragments per wave instruction - a quarter / half of each 128-byte line - and the output-bound
// projections (K = 256: FFN1, QKV, ctc_fc, their data gradients) ran at 1.5-2 TB/s of stores.  With `scratch` (this wave's 8 KiB
// of the LDS the K loop has finished with) the 64 x 64 sub-tile is transposed through LDS instead: epilogue math in the
// accumulator layout, XOR-swizzled LDS image, then 16 bytes per lane with 8 (bf16) / 16 (f32) adjacent lanes covering a
// row's whole 128 / 256 bytes.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// `after_loads` runs exactly once, right after the epilogue has ISSUED its first global reads (bias, sign bits, the first pass's
// addend / mask): the persistent kernels queue the next output tile's LDS-DMA there.  vmcnt retires in order, so with the DMA
// queued earlier (before the last K-tile's multiply) every one of those small reads had to wait for the whole prefetch first.
template <typename Epi, int SCRB = 4096, typename Hook = NoHook>
__device__ __forceinline__ void run_epilogue(const Epi& epi, const f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int r16, int q4,
                                             unsigned char* scratch = nullptr, Hook after_loads = Hook(), bool lead = true) {
    // lead: split-K - only the first K-split of a tile adds the bias / addend (kernel-uniform)
    if (scratch && epi.wide(n0)) {
        const int lane = q4 * 16 + r16;
        const int mw = m0 + wm * 64, nw = n0 + wn * 64;
        const bool bf16_out = epi.elem_size() == 2;
        const int ncol = nw + q4 * 4;                                   // + 16 j
        f32x4 bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = f32x4{0, 0, 0, 0};
        if (epi.has_bias() && lead) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = epi.ld_bias(ncol + 16 * j);
        }
        // sign-bit mask (bf16 outputs): one byte = the 8 consecutive columns a lane stores in the read-back below; all 8 byte
        // loads of the sub-tile (rows lane / 8 + 8 it) go out now, ahead of everything else
        u32x2 bt_in = {0u, 0u}, bt_out = {0u, 0u};
        if (epi.has_bits_in()) bt_in = *reinterpret_cast<const u32x2*>(epi.bits_in_ptr(mw, nw, lane));
        auto pass_rows = [&](auto IBc, int pass) {
            constexpr int IB = decltype(IBc)::value;                     // 16-row blocks of the sub-tile handled by this pass
            f32x4 av[IB][4];
            bf16x4 mk[IB][4];
            // addend: with f32 output (and no ReLU after it) it is added in the read-back layout below - 16 adjacent lanes read a
            // row's 256 bytes - instead of as 64-byte row fragments in the accumulator layout
            const bool add_late = epi.has_add() && lead && !bf16_out && !epi.relu() && !epi.has_mask() && !epi.atomic();
            f32x4 al[4 * IB];
            if (add_late) {
#pragma unroll
                for (int it = 0; it < 4 * IB; ++it) {
                    const int idx = it * 64 + lane;
                    al[it] = epi.get_add(min(mw + pass * IB * 16 + (idx >> 4), epi.M - 1), nw + (idx & 15) * 4);
                }
            }
            if (epi.has_add() && lead && !add_late) {
#pragma unroll
                for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) av[ii][j] = epi.get_add(min(mw + (pass * IB + ii) * 16 + r16, epi.M - 1), ncol + 16 * j);
            }
            if (epi.has_mask()) {
#pragma unroll
                for (int ii = 0; ii < IB; ++ii)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mk[ii][j] = epi.get_mask(min(mw + (pass * IB + ii) * 16 + r16, epi.M - 1), ncol + 16 * j);
            }
            if (pass == 0) after_loads();
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                const int i = pass * IB + ii, rl = ii * 16 + r16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    f32x4 v = acc[i][j] + bv[j];
                    if (epi.relu()) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) v[x] = fmaxf(v[x], 0.f);
                    }
                    if (epi.has_add() && lead && !add_late) v += av[ii][j];
                    if (epi.has_mask()) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) v[x] = ((float)mk[ii][j][x] > 0.f) ? v[x] : 0.f;
                    }
                    v = epi.post(ncol + 16 * j, v);
                    if (bf16_out) {
                        const bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                        const int c = 2 * j + (q4 >> 1);
                        *reinterpret_cast<bf16x4*>(scratch + rl * 128 + ((c ^ (rl & 7)) << 4) + (q4 & 1) * 8) = o;
                    } else {
                        const int c = 4 * j + q4;
                        *reinterpret_cast<f32x4*>(scratch + rl * 256 + ((c ^ (rl & 7)) << 4)) = v;
                    }
                }
            }
            if (bf16_out) {
                const int ch = lane & 7, mf = mw + pass * IB * 16 + (lane >> 3);
                auto rc = epi.cur(mf, nw);
#pragma unroll
                for (int it = 0; it < 2 * IB; ++it) {
                    const int rl = it * 8 + (lane >> 3), m = mf + it * 8;
                    u32x4 d = *reinterpret_cast<const u32x4*>(scratch + rl * 128 + ((ch ^ (rl & 7)) << 4));
                    if (epi.has_bits_in()) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            const int bi = (pass * 2 * IB + it) & 7;
                            const unsigned bt = bt_in[bi >> 2] >> ((bi & 3) * 8);
                            const unsigned lo = (bt >> (2 * x)) & 1u, hi = (bt >> (2 * x + 1)) & 1u;
                            d[x] &= ((0u - lo) & 0xFFFFu) | ((0u - hi) & 0xFFFF0000u);
                        }
                    }
                    if (m < epi.M) {
                        *reinterpret_cast<u32x4*>(rc.p + ch * 16) = d;
                        if (epi.has_bits_out()) {      // 8 consecutive bf16 outputs of this row -> one byte of sign bits (+0 and -0 are "off")
                            // branch-free: (magnitude + 0x7fff) carries into bit 15 / 31 of each half iff the magnitude is nonzero
                            unsigned w = 0;
#pragma unroll
                            for (int x = 0; x < 4; ++x) {
                                unsigned t = (d[x] & 0x7fff7fffu) + 0x7fff7fffu;
                                if (!epi.relu()) t &= ~d[x];                   // without a ReLU in front negative outputs are "off" too
                                w |= (t & 0x80008000u) >> (15 - 2 * x);        // half 0 -> bit 2x, half 1 -> bit 16 + 2x
                            }
                            const int bi = (pass * 2 * IB + it) & 7;
                            bt_out[bi >> 2] |= ((w & 0x55u) | ((w >> 15) & 0xAAu)) << ((bi & 3) * 8);
                        }
                    }
                    epi.step(rc, 8);
                }
            } else {
                if (epi.atomic()) {
                    // split-K: one row per wave instruction, lane = column - 64 lanes add 256 contiguous bytes (2 full cache lines).
                    // (16-byte chunks per lane meant 4 instructions of 4 rows x 16 strided dwords: 8 quarter-filled lines each, and
                    // the memory-side float atomics are paid per line)
                    auto rc = epi.cur(mw + pass * IB * 16, nw);
#pragma unroll
                    for (int rl = 0; rl < 16 * IB; ++rl) {
                        const float v = *reinterpret_cast<const float*>(scratch + rl * 256 + ((((lane >> 2) ^ (rl & 7))) << 4) + (lane & 3) * 4);
                        if (mw + pass * IB * 16 + rl < epi.M) atomicAdd(reinterpret_cast<float*>(rc.p) + lane, v);
                        epi.step(rc, 1);
                    }
                    return;
                }
                const int ch = lane & 15, mf = mw + pass * IB * 16 + (lane >> 4);
                auto rc = epi.cur(mf, nw);
#pragma unroll
                for (int it = 0; it < 4 * IB; ++it) {
                    const int rl = it * 4 + (lane >> 4), m = mf + it * 4;
                    u32x4 d = *reinterpret_cast<const u32x4*>(scratch + rl * 256 + ((ch ^ (rl & 7)) << 4));
                    if (add_late) d = __builtin_bit_cast(u32x4, __builtin_bit_cast(f32x4, d) + al[it]);
                    if (m < epi.M) *reinterpret_cast<u32x4*>(rc.p + ch * 16) = d;
                    epi.step(rc, 4);
                }
            }
        };
        if (bf16_out) {
            constexpr int IB = SCRB / 2048;          // 128 B per row
#pragma unroll
            for (int pass = 0; pass < 4 / IB; ++pass) pass_rows(std::integral_constant<int, IB>{}, pass);
            if (epi.has_bits_out()) *reinterpret_cast<u32x2*>(epi.bits_out_ptr(mw, nw, lane)) = bt_out;   // rows past M: zero bytes, in the pad
        } else {
            constexpr int IB = SCRB / 4096;          // 256 B per row
#pragma unroll
            for (int pass = 0; pass < 4 / IB; ++pass) pass_rows(std::integral_constant<int, IB>{}, pass);
        }
        return;
    }
    after_loads();
    if (epi.fast(n0)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + r16;
            if (m < epi.M) {
                const auto r = epi.row(m);
#pragma unroll
                for (int j = 0; j < 4; ++j) epi.store_fast(r, n0 + wn * 64 + j * 16 + q4 * 4, acc[i][j]);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + r16;
#pragma unroll
            for (int j = 0; j < 4; ++j) epi.store4(m, n0 + wn * 64 + j * 16 + q4 * 4, acc[i][j]);
        }
    }
}

template <typename AT, typename CT, typename Epi>
__global__ __launch_bounds__(NT, 2) void gemm_nt_kernel(const AT* __restrict__ A, int64_t lda, const CT* __restrict__ W,
                                                        int64_t ldw, int M, int N, int K, int tiles_n, int nwg, Epi epi) {
    constexpr int KT = KTraits<CT>::KT, CH = KTraits<CT>::CH;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BM * ROWB];
    unsigned char* As = smem;
    unsigned char* Bs = smem + BM * ROWB;

    // XCD-aware bijective remap (8 XCDs, blocks dealt round-robin): each XCD gets a contiguous tile range.
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, q4 = lane >> 4;

    Chunk<AT, CT> ra[4];
    Chunk<CT, CT> rb[4];
    const int nk = (K + KT - 1) / KT;

    // per-thread chunk pointers, hoisted: this thread always stages chunk c = tid & 7 of rows (tid >> 3) + 32 i
    const int cch = (tid & 7) * CH;
    const AT* aptr[4];
    const CT* wptr[4];
    bool aok[4], wok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        aok[i] = m0 + row < M;
        wok[i] = n0 + row < N;
        aptr[i] = A + (int64_t)min(m0 + row, M - 1) * lda + cch;
        wptr[i] = W + (int64_t)min(n0 + row, N - 1) * ldw + cch;
    }
    const bool rows_full = (m0 + BM <= M) && (n0 + BN <= N);   // block-uniform
    auto gload = [&](int kt) {
        const int kb = kt * KT;
        if (rows_full && kb + KT <= K) {                        // interior tile: no per-load predicates / branches
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i].load(aptr[i] + kb, true);
                rb[i].load(wptr[i] + kb, true);
            }
        } else {
            const bool kok = kb + cch < K;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i].load(aptr[i] + kb, aok[i] && kok);
                rb[i].load(wptr[i] + kb, wok[i] && kok);
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + NT * i;
            const int row = id >> 3, c = id & 7;
            const int off = row * ROWB + ((c ^ (row & 7)) << 4);
            *reinterpret_cast<u32x4*>(As + off) = ra[i].get();
            *reinterpret_cast<u32x4*>(Bs + off) = rb[i].get();
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int chunk = g * 4 + q4;
            u32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int arow = wm * 64 + i * 16 + r16;
                a[i] = *reinterpret_cast<const u32x4*>(As + arow * ROWB + ((chunk ^ (arow & 7)) << 4));
                const int brow = wn * 64 + i * 16 + r16;
                b[i] = *reinterpret_cast<const u32x4*>(Bs + brow * ROWB + ((chunk ^ (brow & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<CT>::run(b[j], a[i], acc[i][j]);  // D[n_local][m_local]
        }
    }
    __syncthreads();   // every wave is done with the operand tiles: their LDS becomes the epilogue's transpose scratch
    run_epilogue(epi, acc, m0, n0, wm, wn, r16, q4, smem + wave * 8192);
}

// ---- LDS-DMA fast path (bf16 x bf16, K % 64 == 0): global_load_lds_dwordx4 writes each 1-KiB piece (8 rows x 128 B) of the
// A / W tiles straight into LDS - no staging VGPRs, no ds_write - into a DOUBLE-buffered image; the loads of tile k+1 are in
// flight while tile k is multiplied, and there is one barrier per K-tile.  The DMA destination is lane-linear, so the XOR
// swizzle is applied to the per-lane SOURCE address (LDS slot p of row r is filled with chunk p ^ (r & 7)) and undone by the
// same XOR on the fragment reads.  Rows past M / N are clamped to the last valid row (their outputs are never stored).
// The kernel is PERSISTENT: a launch is at most 2 workgroups per CU, and each walks a sequence of output tiles with the K-tile
// stream running straight across tile boundaries - the DMA of the next tile's first K-tile is issued before the current tile's
// last multiply, so its latency and the epilogue's global stores overlap.  With K = 256 (4 K-tiles per output tile: every
// projection of the d_model = 256 configs) the un-overlapped prologue + epilogue was most of a workgroup's life.
// Tile order: hardware workgroup b sits on XCD b & 7 (round-robin dispatch); every XCD owns one contiguous range of tile ids
// (n fastest), so the workgroups that share an A row-panel run on the same L2 at about the same time.
template <typename Epi, bool KS1 = false>
__global__ __launch_bounds__(NT, 2) void gemm_nt_glds_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W,
                                                             int64_t ldw, int M, int N, int K, int tiles_n, int ntiles, int ksplit_rt,
                                                             Epi epi) {
    // KS1 (no split-K, the common case): the K-range arithmetic - 64-bit products and divisions by a run-time split count, several
    // hundred scalar instructions per output tile in the listing, as many as a K = 256 tile's whole K loop - folds away
    const int ksplit = KS1 ? 1 : ksplit_rt;
    // ksplit > 1 (host: few output tiles, long K, f32 C): `ntiles` counts (tile, K-split) pairs, split fastest; every pair is a
    // "tile" of the walk below with its own K range, and the epilogue adds atomically (epi.atomic()) into a zeroed C
    constexpr int KT = 64, TILE = BM * ROWB;  // 16 KiB per operand tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE];   // [buf][A|B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q4 = lane >> 4;
    // this workgroup's tile sequence: ids first, first + step, ... < end  (all inside its XCD's range)
    int first, end, step;
    {
        const int G = gridDim.x, xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        end = lo + q + (xcd < r ? 1 : 0);
        step = (G >> 3) + ((G & 7) > xcd ? 1 : 0);      // workgroups of this launch on this XCD
        first = lo + li;
    }
    if (first >= end) return;
    const int nk_all = K / KT;

    const bf16_t* asrc[4];
    const bf16_t* wsrc[4];
    auto k_begin = [&](int vt) { return (int)((int64_t)(vt % ksplit) * nk_all / ksplit); };
    auto k_count = [&](int vt) { return (int)((int64_t)(vt % ksplit + 1) * nk_all / ksplit) - k_begin(vt); };
    auto set_tile = [&](int vt) {   // per-lane source rows of the 4 pieces this wave stages per operand (piece p: tile rows 8p..8p+7)
        const int tile = vt / ksplit;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n, kb = k_begin(vt) * KT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 8 * (wave * 4 + i) + (lane >> 3);
            const int c = (lane & 7) ^ (row & 7);
            asrc[i] = A + (int64_t)min(tm * BM + row, M - 1) * lda + c * 8 + kb;
            wsrc[i] = W + (int64_t)min(tn * BN + row, N - 1) * ldw + c * 8 + kb;
        }
    };
    auto stage = [&](int buf, int kt) {
        unsigned char* base = smem + buf * 2 * TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + kt * KT),
                                             (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc[i] + kt * KT),
                                             (__attribute__((address_space(3))) void*)(base + TILE + p * 1024), 16, 0, 0);
        }
    };

    set_tile(first);
    stage(0, 0);
    int cur = 0;
    __syncthreads();   // drains the DMA (vmcnt) and publishes the first K-tile
    for (int tile = first; tile < end; tile += step) {
        const int otile = tile / ksplit;
        const int tm = otile / tiles_n, tn = otile - tm * tiles_n, nk = k_count(tile);
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            const unsigned char* As = smem + cur * 2 * TILE;
            const unsigned char* Bs = As + TILE;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int chunk = g * 4 + q4;
                u32x4 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int arow = wm * 64 + i * 16 + r16;
                    a[i] = *reinterpret_cast<const u32x4*>(As + arow * ROWB + ((chunk ^ (arow & 7)) << 4));
                    const int brow = wn * 64 + i * 16 + r16;
                    b[i] = *reinterpret_cast<const u32x4*>(Bs + brow * ROWB + ((chunk ^ (brow & 7)) << 4));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma<bf16_t>::run(b[j], a[i], acc[i][j]);
            }
            __syncthreads();   // next K-tile landed (the barrier's fence waits for the outstanding LDS-DMA) and `cur` is free to overwrite
            cur ^= 1;
        }
        // the buffer just multiplied (cur ^ 1 after the toggle) is the epilogue's scratch; buffer `cur` (multiplied one K-tile
        // earlier, every wave is past the barrier since) takes the next output tile's first K-tile, queued from inside the epilogue
        const bool more = tile + step < end;
        run_epilogue(epi, acc, tm * BM, tn * BN, wm, wn, r16, q4, smem + (cur ^ 1) * 2 * TILE + wave * 8192, [&]() {
            if (more) {
                set_tile(tile + step);
                stage(cur, 0);
            }
        }, tile % ksplit == 0);
        if (more) __syncthreads();   // drains the DMA and frees the scratch
    }
}

// ---- NN variant (data gradient): C[M,N] = A[M,Kr] . Bm[Kr,N], Bm row-major as the weight is stored ([out,in] with the
// reduction over `out`).  Same tile / MFMA loop; only the B staging differs: 4(k) x 4(n) register transposes + 8-byte LDS
// writes build the K-contiguous Bs[n][k] image, so no transposed weight copy ever exists in HBM.  bf16 MFMA only.
template <typename AT, typename Epi>
__global__ __launch_bounds__(NT, 2) void gemm_nn_kernel(const AT* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Bm,
                                                        int64_t ldb, int M, int N, int K, int tiles_n, int nwg, Epi epi) {
    constexpr int KT = 64, CH = 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * BM * ROWB];
    unsigned char* As = smem;
    unsigned char* Bs = smem + BM * ROWB;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q4 = lane >> 4;

    Chunk<AT, bf16_t> ra[4];
    u32x2 rb[2][4];
    const int nk = (K + KT - 1) / KT;
    const int cch = (tid & 7) * CH;
    const AT* aptr[4];
    bool aok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + 32 * i;
        aok[i] = m0 + row < M;
        aptr[i] = A + (int64_t)min(m0 + row, M - 1) * lda + cch;
    }
    // B blocks: this thread owns column group cg = tid & 31 and k-row groups rg = (tid >> 5) + 8 i
    const int bn = n0 + 4 * (tid & 31);
    const bool bnok = bn < N;
    const bf16_t* bptr = Bm + (int64_t)(4 * (tid >> 5)) * ldb + min(bn, N - 4 < 0 ? 0 : N - 4);
    const bool rows_full = (m0 + BM <= M) && (n0 + BN <= N);   // block-uniform
    auto gload = [&](int kt) {
        const int kb = kt * KT;
        if (rows_full && kb + KT <= K) {                        // interior tile: no per-load predicates / branches
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i].load(aptr[i] + kb, true);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    rb[i][kk] = *reinterpret_cast<const u32x2*>(bptr + (int64_t)(kb + 32 * i + kk) * ldb);
        } else {
            const bool kok = kb + cch < K;
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i].load(aptr[i] + kb, aok[i] && kok);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int k = kb + 4 * (tid >> 5) + 32 * i + kk;
                    rb[i][kk] = (k < K && bnok) ? *reinterpret_cast<const u32x2*>(bptr + (int64_t)(kb + 32 * i + kk) * ldb) : u32x2{0, 0};
                }
            }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = tid + NT * i;
            const int row = id >> 3, c = id & 7;
            *reinterpret_cast<u32x4*>(As + row * ROWB + ((c ^ (row & 7)) << 4)) = ra[i].get();
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int id = tid + NT * i;
            const int cg = id & 31, rg = id >> 5;
            u32x2 ct[4];
            transpose4x4_bf16(rb[i], ct);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int row = 4 * cg + dd;
                *reinterpret_cast<u32x2*>(Bs + row * ROWB + (((rg >> 1) ^ (row & 7)) << 4) + ((rg & 1) << 3)) = ct[dd];
            }
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    // split-K (gridDim.y > 1, host: pick_ksplit): this workgroup owns K-tiles [kt0, kt1) and adds its partial tile atomically
    const int kt0 = (int)((int64_t)blockIdx.y * nk / gridDim.y), kt1 = (int)((int64_t)(blockIdx.y + 1) * nk / gridDim.y);
    gload(kt0);
    for (int kt = kt0; kt < kt1; ++kt) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (kt + 1 < kt1) gload(kt + 1);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int chunk = g * 4 + q4;
            u32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int arow = wm * 64 + i * 16 + r16;
                a[i] = *reinterpret_cast<const u32x4*>(As + arow * ROWB + ((chunk ^ (arow & 7)) << 4));
                const int brow = wn * 64 + i * 16 + r16;
                b[i] = *reinterpret_cast<const u32x4*>(Bs + brow * ROWB + ((chunk ^ (brow & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<bf16_t>::run(b[j], a[i], acc[i][j]);
        }
    }
    __syncthreads();   // operand tiles are dead: their LDS is the epilogue's transpose scratch
    run_epilogue(epi, acc, m0, n0, wm, wn, r16, q4, smem + wave * 8192, NoHook(), blockIdx.y == 0);
}

// ---- NN variant, LDS-DMA + hardware-transpose form (bf16 A, K % 64 == 0, N % 128 == 0): the A tile is staged exactly like the
// glds NT kernel; the weight tile [64 k][128 n] is copied ROW-MAJOR (as stored) by LDS-DMA and its MFMA fragments - 8 consecutive
// k for one n - come from ds_read_b64_tr_b16 (see backward.hip gemm_tn_tr_kernel for the swizzle / conflict analysis).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ int tr_sw(int row) { return ((row & 3) << 1) ^ (((row >> 3) & 1) << 3); }
__device__ __forceinline__ u32x4 tr_frag(const unsigned char* tile, int row0, int col) {
    const int c = col >> 3, sub = (col & 7) * 2;
    const unsigned char* p0 = tile + row0 * 256 + ((c ^ tr_sw(row0)) << 4) + sub;
    const unsigned char* p1 = tile + (row0 + 4) * 256 + ((c ^ tr_sw(row0 + 4)) << 4) + sub;
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    const u32x2 a = __builtin_bit_cast(u32x2, v0), b = __builtin_bit_cast(u32x2, v1);
    return u32x4{a[0], a[1], b[0], b[1]};
}

template <typename Epi, bool KS1 = false>
__global__ __launch_bounds__(NT, 2) void gemm_nn_tr_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Bm,
                                                           int64_t ldb, int M, int N, int K, int tiles_n, int ntiles, int ksplit_rt,
                                                           Epi epi) {
    const int ksplit = KS1 ? 1 : ksplit_rt;     // (see gemm_nt_glds_kernel)
    constexpr int KT = 64, TILE = BM * ROWB;   // 16 KiB: A tile [128 m][64 k]; B tile [64 k][128 n]
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q4 = lane >> 4;
    // persistent, like gemm_nt_glds_kernel: <= 2 workgroups per CU, each walking its XCD's contiguous tile range with the K-tile
    // stream running across tile boundaries (one workgroup per tile left the first K-tile's latency and the epilogue exposed:
    // the K = 256 hidden-gradient GEMM spent 59 % of its wave cycles waiting)
    int first, end, step;
    {
        const int G = gridDim.x, xcd = blockIdx.x & 7, li = blockIdx.x >> 3;
        const int q = ntiles >> 3, r = ntiles & 7;
        const int lo = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        end = lo + q + (xcd < r ? 1 : 0);
        step = (G >> 3) + ((G & 7) > xcd ? 1 : 0);
        first = lo + li;
    }
    if (first >= end) return;
    // K need not be a multiple of 64: A's rows are readable and ZERO up to the next multiple (host contract: a padded gradient
    // buffer), the weight rows past K - 1 are clamped to the last one (their products meet those zeros)
    const int nk_all = (K + KT - 1) / KT;

    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
    int brow_k[4];
    auto k_begin = [&](int vt) { return (int)((int64_t)(vt % ksplit) * nk_all / ksplit); };      // split-K: see gemm_nt_glds_kernel
    auto k_count = [&](int vt) { return (int)((int64_t)(vt % ksplit + 1) * nk_all / ksplit) - k_begin(vt); };
    auto set_tile = [&](int vt) {
        const int tile = vt / ksplit;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n, kb = k_begin(vt) * KT;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int arow = 8 * (wave * 4 + i) + (lane >> 3);                   // A piece: 8 rows x 128 B
            asrc[i] = A + (int64_t)min(tm * BM + arow, M - 1) * lda + (((lane & 7) ^ (arow & 7)) * 8) + kb;
            const int brow = 4 * (wave * 4 + i) + (lane >> 4);                   // B piece: 4 k-rows x 256 B
            brow_k[i] = kb + brow;
            bsrc[i] = Bm + tn * BN + (((lane & 15) ^ tr_sw(brow)) * 8);
        }
    };
    auto stage = [&](int buf, int kt) {
        unsigned char* base = smem + buf * 2 * TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + kt * KT),
                                             (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (int64_t)min(brow_k[i] + kt * KT, K - 1) * ldb),
                                             (__attribute__((address_space(3))) void*)(base + TILE + p * 1024), 16, 0, 0);
        }
    };
    set_tile(first);
    stage(0, 0);
    int cur = 0;
    __syncthreads();
    for (int tile = first; tile < end; tile += step) {
        const int otile = tile / ksplit;
        const int tm = otile / tiles_n, tn = otile - tm * tiles_n, nk = k_count(tile);
        f32x4 acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
            const unsigned char* As = smem + cur * 2 * TILE;
            const unsigned char* Bt = As + TILE;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int chunk = g * 4 + q4;
                const int row0 = g * 32 + 8 * q4 + (r16 >> 2), csub = 4 * (r16 & 3);
                u32x4 a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int arow = wm * 64 + i * 16 + r16;
                    a[i] = *reinterpret_cast<const u32x4*>(As + arow * ROWB + ((chunk ^ (arow & 7)) << 4));
                    b[i] = tr_frag(Bt, row0, wn * 64 + i * 16 + csub);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma<bf16_t>::run(b[j], a[i], acc[i][j]);
            }
            __syncthreads();
            cur ^= 1;
        }
        // the buffer just multiplied (cur ^ 1 after the toggle) is the epilogue's scratch; buffer `cur` (multiplied one K-tile
        // earlier, every wave is past the barrier since) takes the next output tile's first K-tile, queued from inside the epilogue
        const bool more = tile + step < end;
        run_epilogue(epi, acc, tm * BM, tn * BN, wm, wn, r16, q4, smem + (cur ^ 1) * 2 * TILE + wave * 8192, [&]() {
            if (more) {
                set_tile(tile + step);
                stage(cur, 0);
            }
        }, tile % ksplit == 0);
        if (more) __syncthreads();   // drains the DMA and frees the scratch
    }
}


template <typename AT, typename CT, typename Epi>
int launch_gemm(hipStream_t s, const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K, const Epi& epi) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    hipLaunchKernelGGL((gemm_nt_kernel<AT, CT, Epi>), dim3(nwg), dim3(NT), 0, s, reinterpret_cast<const AT*>(A), lda,
                       reinterpret_cast<const CT*>(W), ldw, M, N, K, tiles_n, nwg, epi);
    ASR_LAUNCH_CHECK("gemm_nt");
    return 0;
}

bool dense_vec_ok(const EpiDense& e) {
    const size_t ca = e.c_dtype == ASR_F32 ? 16 : 8;
    return (e.ldc % 4 == 0) && asr_aligned(e.C, ca) && (!e.bias || asr_aligned(e.bias, 16)) &&
           (!e.addend || (e.ld_add % 4 == 0 && asr_aligned(e.addend, 16))) &&
           (!e.relu_mask || (e.ld_mask % 4 == 0 && asr_aligned(e.relu_mask, 8)));
}

bool dense_wide_ok(const EpiDense& e) {
    if (!e.vec_ok) return false;
    return e.c_dtype == ASR_F32 ? true : (e.ldc % 8 == 0 && asr_aligned(e.C, 16));
}

int check_operands(const void* A, int a_dtype, int64_t lda, const void* W, int w_dtype, int64_t ldw, int K) {
    ASR_REQUIRE(A && W, ASR_ERR_ARG, "gemm: null operand");
    ASR_REQUIRE((a_dtype == ASR_F32 || a_dtype == ASR_BF16) && (w_dtype == ASR_F32 || w_dtype == ASR_BF16), ASR_ERR_ARG,
                "gemm: bad dtype code");
    ASR_REQUIRE(!(a_dtype == ASR_BF16 && w_dtype == ASR_F32), ASR_ERR_UNSUPPORTED, "gemm: bf16 activations with f32 weights");
    const int ch = (w_dtype == ASR_BF16) ? 8 : 4;
    ASR_REQUIRE(K % ch == 0 && lda % ch == 0 && ldw % ch == 0, ASR_ERR_ALIGN,
                "gemm: K=%d lda=%lld ldw=%lld must be multiples of %d", K, (long long)lda, (long long)ldw, ch);
    ASR_REQUIRE(asr_aligned(A, 16) && asr_aligned(W, 16), ASR_ERR_ALIGN, "gemm: A/W must be 16-byte aligned");
    return 0;
}

// Split-K for the persistent kernels: the decoder's GEMMs have B*(U+1) = 1632 rows - 26 output tiles on a 256-CU chip, each a serial
// walk over K (32 K-tiles at K = 2048: ~40 us for 1.7 GFLOP).  With few tiles, a long K and a plain f32 C (no ReLU / mask / bf16
// rounding after the sum) the K range is cut so that ~all CUs get a (tile, split) pair; C is zeroed and the pairs add atomically.
__global__ __launch_bounds__(256) void zero_f32x4_kernel(f32x4* __restrict__ p, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = f32x4{0, 0, 0, 0};
}
// hipMemsetAsync of the 1.6 MB decoder-sized C took ~25 us on this stack; a plain kernel takes ~3
int zero_c(hipStream_t s, void* C, int64_t n) {   // n floats, n % 4 == 0, C 16-byte aligned (wide_ok)
    const int64_t n4 = n / 4;
    hipLaunchKernelGGL(zero_f32x4_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<f32x4*>(C), n4);
    ASR_LAUNCH_CHECK("gemm_zero_c");
    return 0;
}
int pick_ksplit(const EpiDense& e, int ntiles, int K) {
    constexpr int max_tiles = 64;      // (no split-K at all: step +0.24 ms; 128 tiles: +-0 .. +0.1 ms on S2)
    const int nk = K / 64;
    if (asr_deterministic() || !e.wide_ok || e.c_dtype != ASR_F32 || (e.flags & ASR_GEMM_RELU) || e.relu_mask || e.bits_in || e.bits_out ||
        e.N % BN != 0 || e.ldc != e.N || ntiles > max_tiles || nk < 8 || (const void*)e.addend == (const void*)e.C)
        return 1;
    int sp = nk / 4;                       // >= 4 K-tiles per split
    if (sp > 256 / ntiles) sp = 256 / ntiles;
    return sp < 2 ? 1 : sp;
}

template <typename Epi> int launch_glds(hipStream_t s, const void* A, int64_t lda, const void* W, int64_t ldw, int M, int N, int K,
                                        const Epi& epi, int ksplit = 1) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, ntiles = tiles_m * tiles_n * ksplit;
    constexpr int max_wg = 512;   // 2 per CU (64 KiB of LDS each); 384 .. 768: +0.1 .. 0.8 ms per step
    const int nwg = ntiles < max_wg ? ntiles : max_wg;
    if (ksplit == 1)
        hipLaunchKernelGGL((gemm_nt_glds_kernel<Epi, true>), dim3(nwg), dim3(NT), 0, s, reinterpret_cast<const bf16_t*>(A), lda,
                           reinterpret_cast<const bf16_t*>(W), ldw, M, N, K, tiles_n, ntiles, ksplit, epi);
    else
        hipLaunchKernelGGL((gemm_nt_glds_kernel<Epi, false>), dim3(nwg), dim3(NT), 0, s, reinterpret_cast<const bf16_t*>(A), lda,
                           reinterpret_cast<const bf16_t*>(W), ldw, M, N, K, tiles_n, ntiles, ksplit, epi);
    ASR_LAUNCH_CHECK("gemm_nt_glds");
    return 0;
}

template <typename Epi> int dispatch(hipStream_t s, const void* A, int a_dtype, int64_t lda, const void* W, int w_dtype,
                                     int64_t ldw, int M, int N, int K, const Epi& epi) {
    // (A/B on one MI355X after the epilogue / addressing diet: LDS-DMA double buffering +10..26 % at K = 2048, roughly neutral at
    // K = 256 (-7 % on the [32000,2048,256] FFN1 shape, +10 % on the narrow-N ones, whole train step 1-2 % faster), so it is the
    // default whenever K % 64 == 0)
    if (a_dtype == ASR_BF16 && w_dtype == ASR_BF16 && K % 64 == 0) {
        if constexpr (std::is_same<Epi, EpiDense>::value) {
            const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
            if (const int sp = pick_ksplit(epi, tiles, K); sp > 1) {
                if (!epi.c_is_zero)
                    if (int rc = zero_c(s, epi.C, (int64_t)M * N)) return rc;
                EpiDense e2 = epi;
                e2.atomic_out = true;
                if (dense_mode(e2) == (1u | 128u)) return launch_glds(s, A, lda, W, ldw, M, N, K, dense_as<1u | 128u>(e2), sp);   // decoder FFN2
                if (dense_mode(e2) == 128u) return launch_glds(s, A, lda, W, ldw, M, N, K, dense_as<128u>(e2), sp);
                return launch_glds(s, A, lda, W, ldw, M, N, K, e2, sp);
            }
            if (epi.wide_ok && !(epi.flags & ~ASR_GEMM_RELU)) {
                switch (dense_mode(epi)) {   // the combinations the model's projections use; anything else takes the run-time form
                    case 1u | 2u | 16u: return launch_glds(s, A, lda, W, ldw, M, N, K, dense_as<1u | 2u | 16u>(epi));   // FFN1
                    case 1u | 2u | 16u | 32u: return launch_glds(s, A, lda, W, ldw, M, N, K, dense_as<1u | 2u | 16u | 32u>(epi));   // FFN1 + sign bits
                    case 1u | 16u: return launch_glds(s, A, lda, W, ldw, M, N, K, dense_as<1u | 16u>(epi));
                    case 1u: return launch_glds(s, A, lda, W, ldw, M, N, K, dense_as<1u>(epi));                         // FFN2, fc, affine
                    case 0u: return launch_glds(s, A, lda, W, ldw, M, N, K, dense_as<0u>(epi));                         // vocab projections
                    default: break;
                }
            }
        }
        return launch_glds(s, A, lda, W, ldw, M, N, K, epi);
    }
    if (w_dtype == ASR_F32) return launch_gemm<float, float>(s, A, lda, W, ldw, M, N, K, epi);
    if (a_dtype == ASR_F32) return launch_gemm<float, bf16_t>(s, A, lda, W, ldw, M, N, K, epi);
    return launch_gemm<bf16_t, bf16_t>(s, A, lda, W, ldw, M, N, K, epi);
}

}  // namespace

extern "C" int asr_gemm_nt(void* stream, const void* A, int a_dtype, int64_t lda, const void* W, int w_dtype, int64_t ldw,
                           const float* bias, void* C, int c_dtype, int64_t ldc, int M, int N, int K, unsigned flags) {
    ASR_REQUIRE(M > 0 && N > 0 && K > 0 && C, ASR_ERR_ARG, "gemm: M=%d N=%d K=%d C=%p", M, N, K, C);
    ASR_REQUIRE(c_dtype == ASR_F32 || c_dtype == ASR_BF16, ASR_ERR_ARG, "gemm: bad c_dtype");
    if (int rc = check_operands(A, a_dtype, lda, W, w_dtype, ldw, K)) return rc;
    EpiDense epi{C, c_dtype, ldc, bias, flags & ASR_GEMM_RELU, M, N, nullptr, 0, nullptr, 0, false};
    epi.vec_ok = dense_vec_ok(epi);
    epi.wide_ok = dense_wide_ok(epi);
    epi.c_is_zero = (flags & ASR_GEMM_C_IS_ZERO) != 0;
    return dispatch(static_cast<hipStream_t>(stream), A, a_dtype, lda, W, w_dtype, ldw, M, N, K, epi);
}

extern "C" int asr_gemm_nt_ex(void* stream, const void* A, int a_dtype, int64_t lda, const void* W, int w_dtype, int64_t ldw,
                              const float* bias, void* C, int c_dtype, int64_t ldc, int M, int N, int K, unsigned flags,
                              const float* addend, int64_t ld_add, const void* relu_mask, int64_t ld_mask, void* relu_bits_out,
                              int64_t ld_bits) {
    ASR_REQUIRE(M > 0 && N > 0 && K > 0 && C, ASR_ERR_ARG, "gemm_ex: M=%d N=%d K=%d C=%p", M, N, K, C);
    ASR_REQUIRE(c_dtype == ASR_F32 || c_dtype == ASR_BF16, ASR_ERR_ARG, "gemm_ex: bad c_dtype");
    if (int rc = check_operands(A, a_dtype, lda, W, w_dtype, ldw, K)) return rc;
    EpiDense epi{C, c_dtype, ldc, bias, flags, M, N, addend, ld_add, reinterpret_cast<const bf16_t*>(relu_mask), ld_mask, false};
    epi.vec_ok = dense_vec_ok(epi);
    epi.wide_ok = dense_wide_ok(epi);
    if (relu_bits_out) {
        // sign bits are produced by the LDS-transposed epilogue of the LDS-DMA kernel only: full 128-column tiles, bf16 x bf16
        ASR_REQUIRE(epi.wide_ok && c_dtype == ASR_BF16 && N % 128 == 0 && ld_bits == N / 8 && a_dtype == ASR_BF16 && w_dtype == ASR_BF16 &&
                        K % 64 == 0 && (flags & ASR_GEMM_RELU),
                    ASR_ERR_UNSUPPORTED, "gemm_ex: relu_bits_out needs bf16 operands/output, ReLU, N %% 128 == 0, K %% 64 == 0, aligned rows");
        epi.bits_out = reinterpret_cast<unsigned char*>(relu_bits_out);
        epi.ld_bits = ld_bits;
    }
    return dispatch(static_cast<hipStream_t>(stream), A, a_dtype, lda, W, w_dtype, ldw, M, N, K, epi);
}

extern "C" int asr_proj_heads(void* stream, const void* X, int x_dtype, int64_t ldx, const void* W, int w_dtype, int64_t ldw,
                              const float* bias, void* out, int64_t proj_stride, int n_proj, int B, int L, int h, int K,
                              float scale_first) {
    ASR_REQUIRE(B > 0 && L > 0 && h > 0 && n_proj > 0 && K > 0 && out, ASR_ERR_ARG, "proj_heads: bad sizes");
    if (int rc = check_operands(X, x_dtype, ldx, W, w_dtype, ldw, K)) return rc;
    ASR_REQUIRE(asr_aligned(out, 16) && (proj_stride % 8 == 0), ASR_ERR_ALIGN, "proj_heads: out alignment");
    ASR_REQUIRE(!bias || asr_aligned(bias, 16), ASR_ERR_ALIGN, "proj_heads: bias must be 16-byte aligned");
    const int M = B * L, N = n_proj * h * 64;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (w_dtype == ASR_F32) {
        EpiHeads<float> epi{reinterpret_cast<float*>(out), proj_stride, bias, L, h, M, N, scale_first};
        return launch_gemm<float, float>(s, X, ldx, W, ldw, M, N, K, epi);
    }
    if (x_dtype == ASR_BF16 && K == 256 && ldx == 256 && ldw == 256) {
        // encoder-sized rows: the feed-forward kernel's structure (ffn.hip: x read once for all heads, store-bound); ASR_AMD_HEADS_ROWS=0
        // switches it off (read per call: the parity test compares the two kernels)
        const char* e = getenv("ASR_AMD_HEADS_ROWS");
        if ((!e || atoi(e) != 0) && M >= 16384) {
            const int rc = asr_proj_heads_rows(s, X, W, bias, out, proj_stride, n_proj, B, L, h, scale_first);
            if (rc != -2) return rc;
        }
    }
    EpiHeads<bf16_t> epi{reinterpret_cast<bf16_t*>(out), proj_stride, bias, L, h, M, N, scale_first};
    if (x_dtype == ASR_F32) return launch_gemm<float, bf16_t>(s, X, ldx, W, ldw, M, N, K, epi);
    if (K % 64 == 0)
        return launch_glds(s, X, ldx, W, ldw, M, N, K, epi);
    return launch_gemm<bf16_t, bf16_t>(s, X, ldx, W, ldw, M, N, K, epi);
}

extern "C" int asr_gemm_nn(void* stream, const void* A, int a_dtype, int64_t lda, const void* Bm, int64_t ldb, const float* bias, void* C,
                           int c_dtype, int64_t ldc, int M, int N, int K, const float* addend, int64_t ld_add, const void* relu_mask,
                           int64_t ld_mask, int mask_is_bits) {
    ASR_REQUIRE(A && Bm && C && M > 0 && N > 0 && K > 0, ASR_ERR_ARG, "gemm_nn: bad args");
    ASR_REQUIRE(lda % 8 == 0 && lda >= (K + 7) / 8 * 8 && N % 4 == 0 && ldb % 4 == 0 && asr_aligned(A, 16) && asr_aligned(Bm, 8),
                ASR_ERR_ALIGN, "gemm_nn: lda=%lld must be a multiple of 8 covering K=%d rounded up (pad columns must be zero), N=%d ldb=%lld of 4",
                (long long)lda, K, N, (long long)ldb);
    // columns K .. lda-1 of A are zero by contract: with lda >= K rounded up to 64 the LDS-DMA kernel takes a K that is no multiple of 64
    const bool k_ok = K % 64 == 0 || lda >= (int64_t)(K + 63) / 64 * 64;
    EpiDense epi{C, c_dtype, ldc, bias, 0u, M, N, addend, ld_add, mask_is_bits ? nullptr : reinterpret_cast<const bf16_t*>(relu_mask),
                 ld_mask, false};
    epi.vec_ok = dense_vec_ok(epi);
    epi.wide_ok = dense_wide_ok(epi);
    epi.c_is_zero = (mask_is_bits & 2) != 0;
    mask_is_bits &= 1;
    if (mask_is_bits && relu_mask) {
        ASR_REQUIRE(epi.wide_ok && a_dtype == ASR_BF16 && k_ok && N % 128 == 0 && ldb % 8 == 0 && asr_aligned(Bm, 16) &&
                        ld_mask == N / 8,
                    ASR_ERR_UNSUPPORTED, "gemm_nn: a sign-bit mask needs the LDS-DMA kernel's shapes (bf16 A, K %% 64 == 0, N %% 128 == 0)");
        epi.bits_in = reinterpret_cast<const unsigned char*>(relu_mask);
        epi.ld_bits = ld_mask;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN, nwg = tiles_m * tiles_n;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (a_dtype == ASR_BF16 && k_ok && N % 128 == 0 && ldb % 8 == 0 && asr_aligned(Bm, 16)) {
        const int budget = asr_launch_budget_current();
        const int max_wg = budget > 0 && budget < 512 ? budget : 512;   // 2 per CU, persistent (asr_launch_budget: fewer, beside latency-bound work)
        const int sp = pick_ksplit(epi, nwg, K);
        if (sp > 1) {
            if (!epi.c_is_zero)
                if (int rc = zero_c(s, C, (int64_t)M * N)) return rc;
            epi.atomic_out = true;
        }
        const int vtiles = nwg * sp;
        const int pwg = vtiles < max_wg ? vtiles : max_wg;
#define LAUNCH_NN_TR(E)                                                                                                          \
    do {                                                                                                                         \
        if (sp == 1)                                                                                                             \
            hipLaunchKernelGGL((gemm_nn_tr_kernel<decltype(E), true>), dim3(pwg), dim3(NT), 0, s, (const bf16_t*)A, lda,         \
                               (const bf16_t*)Bm, ldb, M, N, K, tiles_n, vtiles, sp, E);                                         \
        else                                                                                                                     \
            hipLaunchKernelGGL((gemm_nn_tr_kernel<decltype(E), false>), dim3(pwg), dim3(NT), 0, s, (const bf16_t*)A, lda,        \
                               (const bf16_t*)Bm, ldb, M, N, K, tiles_n, vtiles, sp, E);                                         \
    } while (0)
        const unsigned mode = epi.wide_ok ? dense_mode(epi) : 0xffu;
        if (mode == 4u) LAUNCH_NN_TR(dense_as<4u>(epi));                 // dX = dY . W + residual gradient (f32)
        else if (mode == (4u | 128u)) LAUNCH_NN_TR(dense_as<4u | 128u>(epi));   // ... split-K (decoder rows)
        else if (mode == 128u) LAUNCH_NN_TR(dense_as<128u>(epi));
        else if (mode == (8u | 16u)) LAUNCH_NN_TR(dense_as<8u | 16u>(epi));   // ReLU-masked hidden gradient (bf16)
        else if (mode == (64u | 16u)) LAUNCH_NN_TR(dense_as<64u | 16u>(epi));  // ... masked from the sign bits
        else if (mode == 16u) LAUNCH_NN_TR(dense_as<16u>(epi));
        else if (mode == 0u) LAUNCH_NN_TR(dense_as<0u>(epi));
        else LAUNCH_NN_TR(epi);
#undef LAUNCH_NN_TR
        ASR_LAUNCH_CHECK("gemm_nn_tr");
        return 0;
    }
    const int sp = pick_ksplit(epi, nwg, K);     // e.g. the decoder's vocabulary data gradient [1632 x 256 x 4234]: 26 tiles x 67 K-tiles
    if (sp > 1) {
        if (!epi.c_is_zero)
            if (int rc = zero_c(s, C, (int64_t)M * N)) return rc;
        epi.atomic_out = true;
    }
    if (a_dtype == ASR_F32)
        hipLaunchKernelGGL((gemm_nn_kernel<float, EpiDense>), dim3(nwg, sp), dim3(NT), 0, s, (const float*)A, lda, (const bf16_t*)Bm, ldb, M, N,
                           K, tiles_n, nwg, epi);
    else
        hipLaunchKernelGGL((gemm_nn_kernel<bf16_t, EpiDense>), dim3(nwg, sp), dim3(NT), 0, s, (const bf16_t*)A, lda, (const bf16_t*)Bm, ldb, M, N,
                           K, tiles_n, nwg, epi);
    ASR_LAUNCH_CHECK("gemm_nn");
    return 0;
}
