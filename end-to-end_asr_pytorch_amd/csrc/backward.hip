// Backward-pass kernels other than attention / CTC / CE for gfx950:
//   gemm_tn      weight gradient  dW[N,K] = sum_m dY[m,n] * X[m,k]   (both operands M-major -> transposing LDS staging)
//   colsum       bias gradient    db[n]   = sum_m dY[m,n]
//   layernorm_bwd  d(x+res) of the fused residual + LayerNorm (+mask) kernel, with dgamma / dbeta
//   embed_bwd    scatter-add of row gradients into the embedding table
//   adam_step    fused Adam over one flat parameter buffer (train.py:166-170: betas (0.9,0.98), eps 1e-9, no decay)
#include <stdlib.h>

#include "asr_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// gemm_tn: C[N,K] += A^T . B over an M-range.  Output tile 128(n) x 128(k); reduction step 64 rows of m.
// LDS images At[128 n][64 m], Bt[128 k][64 m] in the NT GEMM's swizzled row format (chunk ^ (row & 7)), filled by
// 4x4 register transposes (8-byte ds_write), so the MFMA inner loop is the NT kernel's.  Split over M (grid.y) with
// float atomics into the (pre-zeroed) fp32 output: weight-gradient tiles are few (N*K/16384) while M is long.
template <typename T> __device__ __forceinline__ u32x2 load4_bf16(const T* p, bool ok);
template <> __device__ __forceinline__ u32x2 load4_bf16<bf16_t>(const bf16_t* p, bool ok) {
    return ok ? *reinterpret_cast<const u32x2*>(p) : u32x2{0, 0};
}
template <> __device__ __forceinline__ u32x2 load4_bf16<float>(const float* p, bool ok) {
    const f32x4 v = ok ? *reinterpret_cast<const f32x4*>(p) : f32x4{0, 0, 0, 0};
    bf16x4 b = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    return __builtin_bit_cast(u32x2, b);
}

template <typename TA, typename TB>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const TA* __restrict__ A, int64_t lda, const TB* __restrict__ Bm, int64_t ldb,
                                                         float* __restrict__ C, int64_t ldc, int M, int N, int K, int tiles_k,
                                                         int m_per_split) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 128 * 128];
    unsigned char* As = smem;
    unsigned char* Bs = smem + 128 * 128;
    const int tn = blockIdx.x / tiles_k, tk = blockIdx.x - tn * tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int m_begin = blockIdx.y * m_per_split, m_end = min(M, m_begin + m_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q4 = lane >> 4;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // per-thread 4x4 blocks: column group cg = tid & 31 (4 columns), m-row groups rg = (tid >> 5) + 8 i (4 rows each), i = 0, 1.
    // Pointers are hoisted; the loop only adds the m offset.
    const int cg = tid & 31;
    const bool aok = n0 + 4 * cg < N, bok = k0 + 4 * cg < K;
    const TA* ap = A + (int64_t)(4 * (tid >> 5)) * lda + min(n0 + 4 * cg, N - 1);
    const TB* bp = Bm + (int64_t)(4 * (tid >> 5)) * ldb + min(k0 + 4 * cg, K - 1);
    u32x2 ra[2][4], rb[2][4];
    const bool cols_full = (n0 + 128 <= N) && (k0 + 128 <= K);      // block-uniform
    auto gload = [&](int m0) {
        if (cols_full && m0 + 64 <= m_end) {                         // interior tile: no per-load predicates / branches
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    ra[i][kk] = load4_bf16<TA>(ap + (int64_t)(m0 + 32 * i + kk) * lda, true);
                    rb[i][kk] = load4_bf16<TB>(bp + (int64_t)(m0 + 32 * i + kk) * ldb, true);
                }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int mo = 32 * i + kk;
                    const bool mok = m0 + 4 * (tid >> 5) + mo < m_end;
                    ra[i][kk] = load4_bf16<TA>(ap + (int64_t)(m0 + mo) * lda, mok && aok);
                    rb[i][kk] = load4_bf16<TB>(bp + (int64_t)(m0 + mo) * ldb, mok && bok);
                }
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rg = (tid >> 5) + 8 * i;
            u32x2 ca[4], cb[4];
            transpose4x4_bf16(ra[i], ca);
            transpose4x4_bf16(rb[i], cb);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int row = 4 * cg + dd;
                const int off = row * 128 + (((rg >> 1) ^ (row & 7)) << 4) + ((rg & 1) << 3);
                *reinterpret_cast<u32x2*>(As + off) = ca[dd];
                *reinterpret_cast<u32x2*>(Bs + off) = cb[dd];
            }
        }
    };

    if (m_begin < m_end) gload(m_begin);
    for (int m0 = m_begin; m0 < m_end; m0 += 64) {
        __syncthreads();
        lstore();
        __syncthreads();
        if (m0 + 64 < m_end) gload(m0 + 64);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int chunk = g * 4 + q4;
            u32x4 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int arow = wm * 64 + i * 16 + r16;
                a[i] = *reinterpret_cast<const u32x4*>(As + arow * 128 + ((chunk ^ (arow & 7)) << 4));
                const int brow = wn * 64 + i * 16 + r16;
                b[i] = *reinterpret_cast<const u32x4*>(Bs + brow * 128 + ((chunk ^ (brow & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<bf16_t>::run(b[j], a[i], acc[i][j]);   // D[k_local][n_local]
        }
    }
    // Accumulate into C with float atomics.  In the MFMA layout a lane holds one n-row and 4 k-columns, i.e. a wave
    // instruction would touch 16 rows x 16-byte pieces (the slow, scattered atomic shape).  The tile is therefore bounced
    // through LDS per wave (2 passes of 32 rows x 64 columns) so that every atomic wave-instruction adds 256 contiguous bytes.
    float* wlds = reinterpret_cast<float*>(smem) + wave * (32 * 64);   // 32 rows x 64 floats per wave (8 KB), XOR-swizzled columns
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = pass * 2 + ii;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(wlds + (ii * 16 + r16) * 64 + ((j * 16 + q4 * 4) ^ (r16 << 2))) = acc[i][j];
        }
        __builtin_amdgcn_wave_barrier();
        const int k = k0 + wn * 64 + lane;
        for (int rr = 0; rr < 32; ++rr) {
            const int n = n0 + wm * 64 + pass * 32 + rr;
            const float v = wlds[rr * 64 + (lane ^ ((rr & 15) << 2))];
            if (n < N && k < K) atomicAdd(C + (int64_t)n * ldc + k, v);
        }
    }
}

// ---- gemm_tn, LDS-DMA + hardware-transpose form (bf16 x bf16, full tiles): the M-major operand tiles [64 m][128 cols] are
// copied row-major straight into LDS by global_load_lds_dwordx4 (double-buffered, one barrier per 64-row step, no staging
// VGPRs, no transposing VALU) and the MFMA operands - 8 consecutive m for one output row/column - are fetched with
// ds_read_b64_tr_b16 (lane i of 16-lane group g receives column c0+i of rows r0+8g..+3; verified on hardware with
// tools/probe_tr.py).  A 16-byte chunk c of LDS row r holds global chunk c ^ sw(r), sw(r) = ((r&3)<<1) ^ (((r>>3)&1)<<3): the
// 8 (row, half-wave group) combinations of one transposing read then land on 8 distinct 32-byte bank segments (conflict-free).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ int tr_sw(int row) { return ((row & 3) << 1) ^ (((row >> 3) & 1) << 3); }
__device__ __forceinline__ u32x4 tr_frag(const unsigned char* tile, int row0, int col) {
    // rows row0..row0+3 and row0+4..row0+7 at 4 consecutive columns starting at `col` (this lane's address per the tr rule)
    const int c = col >> 3, sub = (col & 7) * 2;
    const unsigned char* p0 = tile + row0 * 256 + ((c ^ tr_sw(row0)) << 4) + sub;
    const unsigned char* p1 = tile + (row0 + 4) * 256 + ((c ^ tr_sw(row0 + 4)) << 4) + sub;
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    const u32x2 a = __builtin_bit_cast(u32x2, v0), b = __builtin_bit_cast(u32x2, v1);
    return u32x4{a[0], a[1], b[0], b[1]};
}

__global__ __launch_bounds__(256, 2) void gemm_tn_tr_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ Bm,
                                                            int64_t ldb, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                            int tiles_k, int m_per_split, int splits, float* __restrict__ colsum) {
    constexpr int TILE = 64 * 256;   // 64 rows x 128 bf16
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * TILE];   // [buf][A|B]
    // (tile, M-split) from the linear workgroup id.  Workgroups are dealt round-robin to the 8 XCDs; when the split count is a
    // multiple of 8 every XCD gets whole M-ranges (all output tiles of splits/8 ranges), so each row of dY and X is fetched into
    // one L2 only - with tile-major ids every XCD streamed a quarter of dY and half of X over ALL rows: 296 MB fetched for the
    // 147 MB of the FFN weight gradient (rocprofv3 FETCH_SIZE).
    const int tiles = gridDim.x / splits;
    int tile, split;
    if ((splits & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        split = xcd * (splits >> 3) + slot / tiles;
        tile = slot - (slot / tiles) * tiles;
    } else {
        split = blockIdx.x / tiles;
        tile = blockIdx.x - split * tiles;
    }
    const int tn = tile / tiles_k, tk = tile - tn * tiles_k;
    const int n0 = tn * 128, k0 = tk * 128;
    const int m_begin = split * m_per_split, m_end = min(M, m_begin + m_per_split);   // multiples of 64 by construction
    if (m_begin >= m_end) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, r16 = lane & 15, q4 = lane >> 4;

    // staging: piece p (4 rows x 256 B) = rows 4p..4p+3; wave w stages pieces 4w..4w+3 of each operand
    const bf16_t* asrc[4];
    const bf16_t* bsrc[4];
    int srow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 4 * (wave * 4 + i) + (lane >> 4);
        const int c = (lane & 15) ^ tr_sw(row);
        srow[i] = row;
        asrc[i] = A + (int64_t)row * lda + n0 + c * 8;
        bsrc[i] = Bm + (int64_t)row * ldb + k0 + c * 8;
    }
    auto stage = [&](int buf, int m0) {
        unsigned char* base = smem + buf * 2 * TILE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = wave * 4 + i;
            // rows past M (only in the last, partial step when M is not a multiple of 64) are clamped to a valid row; their A
            // fragments are zeroed before the MFMAs, so whatever they hold contributes nothing
            const int64_t mr = min(m0, M - 1 - srow[i]);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + mr * lda),
                                             (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + mr * ldb),
                                             (__attribute__((address_space(3))) void*)(base + TILE + p * 1024), 16, 0, 0);
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0, 0, 0, 0};

    // bias gradient for free: the workgroups of the first K-tile (and the wn = 0 waves) multiply their A fragments by an all-ones
    // operand as well - every row of that 16x16 product is the column sum of dY over the 32 m just consumed.
    const bool do_colsum = (colsum != nullptr) && (tk == 0) && (wn == 0);
    const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};   // bf16 1.0 x 8
    f32x4 cs[4] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};

    if (m_begin < m_end) stage(0, m_begin);
    __syncthreads();
    int cur = 0;
    for (int m0 = m_begin; m0 < m_end; m0 += 64) {
        if (m0 + 64 < m_end) stage(cur ^ 1, m0 + 64);
        const unsigned char* At = smem + cur * 2 * TILE;
        const unsigned char* Bt = At + TILE;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int row0 = g2 * 32 + 8 * q4 + (r16 >> 2);     // this lane's address row for the transposing read
            const int csub = 4 * (r16 & 3);
            u32x4 a[4], b[4];
            const bool live = m0 + g2 * 32 + 8 * q4 + 8 <= m_end;     // this lane's 8 reduction rows exist (M is a multiple of 8)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = tr_frag(At, row0, wm * 64 + i * 16 + csub);
                if (!live) a[i] = u32x4{0, 0, 0, 0};
                b[i] = tr_frag(Bt, row0, wn * 64 + i * 16 + csub);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Mma<bf16_t>::run(b[j], a[i], acc[i][j]);   // D[k_local][n_local]
            if (do_colsum) {
#pragma unroll
                for (int i = 0; i < 4; ++i) Mma<bf16_t>::run(ones, a[i], cs[i]);        // D[*][n_local] = sum_m A[m][n]
            }
        }
        __syncthreads();
        cur ^= 1;
    }
    if (do_colsum && q4 == 0) {   // rows are identical: lanes 0..15 (row group 0, register 0) hold the 16 column sums of fragment i
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (n0 + wm * 64 + i * 16 + r16 < N) atomicAdd(colsum + n0 + wm * 64 + i * 16 + r16, cs[i][0]);
    }
    float* wlds = reinterpret_cast<float*>(smem) + wave * (32 * 64);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
            const int i = pass * 2 + ii;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<f32x4*>(wlds + (ii * 16 + r16) * 64 + ((j * 16 + q4 * 4) ^ (r16 << 2))) = acc[i][j];
        }
        __builtin_amdgcn_wave_barrier();
        const int k = k0 + wn * 64 + lane;
        for (int rr = 0; rr < 32; ++rr) {
            const int n = n0 + wm * 64 + pass * 32 + rr;
            if (n >= N) break;      // ragged N (host: A's rows are readable up to the next multiple of 128): those output rows do not exist
            atomicAdd(C + (int64_t)n * ldc + k, wlds[rr * 64 + (lane ^ ((rr & 15) << 2))]);
        }
    }
}

// out[n] += sum_m A[m,n]: a workgroup owns a 256-column chunk (64 lanes x 4 columns per vector load) and a row range; its 4
// waves stride over the rows, partials meet in LDS, one float atomic per column per workgroup.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ A, int64_t lda, int M, int N, int rows_per_block,
                                                     float* __restrict__ out) {
    __shared__ float red[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 256 + lane * 4;
    const int m0 = blockIdx.y * rows_per_block, m1 = min(M, m0 + rows_per_block);
    f32x4 s = {0, 0, 0, 0};
    if (n < N) {
        const bool vec = (n + 3 < N) && ((lda & 3) == 0);
        for (int m = m0 + wave; m < m1; m += 4) {
            const T* p = A + (int64_t)m * lda + n;
            if (vec) {
                if constexpr (sizeof(T) == 4) {
                    s += *reinterpret_cast<const f32x4*>(p);
                } else {
                    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
                    s += f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (n + e < N) s[e] += to_f32(p[e]);
            }
        }
    }
    *reinterpret_cast<f32x4*>(&red[wave][lane * 4]) = s;
    __syncthreads();
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < N) atomicAdd(out + c, (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

// narrow contiguous matrices (N <= 128 columns, lda == N: the conv layers' [positions, 32 channels] gradients): the kernel above
// would keep 8 of 64 lanes busy.  The matrix is walked as a flat array of 4-element chunks; a thread's chunks all belong to the
// same column group because its stride is a multiple of the chunks per row.
template <typename T>
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const T* __restrict__ A, int64_t nchunks, int cpr, float* __restrict__ out) {
    __shared__ f32x4 red[256];
    f32x4 s = {0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x; c < nchunks; c += stride) {
        if constexpr (sizeof(T) == 4) {
            s += *reinterpret_cast<const f32x4*>(A + c * 4);
        } else {
            const bf16x4 v = *reinterpret_cast<const bf16x4*>(A + c * 4);
            s += f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < cpr) {                      // thread g < cpr owns column group g: threads g, g + cpr, ... hold its partials
        f32x4 t = {0, 0, 0, 0};
        for (int i = threadIdx.x; i < 256; i += cpr) t += red[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(out + threadIdx.x * 4 + e, t[e]);
    }
}

// ---------------------------------------------------------------------------------------------------------
#ifndef LNB_RPW_
#define LNB_RPW_ 2
#endif
constexpr int LNB_RPW = LNB_RPW_;             // rows per wave handled TOGETHER: all loads issued up front, reductions interleaved
#ifndef LNB_MAX_BLOCKS
#define LNB_MAX_BLOCKS 256
#endif
constexpr int LNB_WAVES = 8;                  // waves per workgroup: 256 workgroups x 8 waves fill the chip like 512 x 4 did, with half the
                                              // contended column atomics at the end (512 workgroups: ~8 us of a 31 us launch)
constexpr int LNB_ROWS = LNB_WAVES * LNB_RPW;  // rows per workgroup

// MAXJ = float4 column groups per lane: 1 for D <= 256, 2 for D <= 512 (the reference's shipped width), 4 up to D = 1024 (that one
// spills: 143 registers to scratch - D = 512 ran on it until round 5, 3x slower than its bytes allow).
template <int MAXJ>
__global__ __launch_bounds__(64 * LNB_WAVES) void add_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ s,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const int32_t* __restrict__ row_len,
                                                                float* __restrict__ ds, void* __restrict__ ds16, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, float* __restrict__ dbias, int M, int L, int D,
                                                                asr_dropout_t drop_x_in, asr_dropout_t drop_y_in,
                                                                const float* __restrict__ beta_y = nullptr) {
    __shared__ float red[3][LNB_WAVES][256 * MAXJ];
    const asr_dropout_t drop_x = drop_resolve(drop_x_in), drop_y = drop_resolve(drop_y_in);
    const float scx = drop_scale(drop_x), scy = drop_scale(drop_y);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float invD = 1.f / (float)D;
    f32x4 ag[MAXJ], ab[MAXJ], as[MAXJ], gam[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        ag[j] = f32x4{0, 0, 0, 0}; ab[j] = f32x4{0, 0, 0, 0}; as[j] = f32x4{0, 0, 0, 0};
        const int c = lane * 4 + 256 * j;
        gam[j] = (c < D) ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0, 0, 0, 0};
    }
    // persistent: a bounded number of workgroups stride over the rows keeping the per-column partials in registers, so the
    // contended dgamma / dbeta / dbias atomics (every workgroup hits the same D addresses) happen once per workgroup, not per 16 rows.
    // The rows of iteration i+1 are loaded (into a second register set) BEFORE iteration i is reduced and stored: with the loads
    // issued only after the previous stores a wave had nothing in flight for most of an iteration, and 8 waves per CU could not
    // cover that (2.8 TB/s).
    struct Rows {
        f32x4 d[LNB_RPW][MAXJ], xh[LNB_RPW][MAXJ];
        float mu[LNB_RPW], rs[LNB_RPW];
        int len[LNB_RPW], b[LNB_RPW], t[LNB_RPW];
        bool live[LNB_RPW];
    };
    auto load_rows = [&](Rows& R, int64_t row0) {
#pragma unroll
        for (int r = 0; r < LNB_RPW; ++r) {
            const int64_t row = row0 + r;
            R.live[r] = row < M;
            const int64_t rr = R.live[r] ? row : (int64_t)M - 1;
            R.b[r] = (int)(rr / L);
            R.t[r] = (int)(rr - (int64_t)R.b[r] * L);
            R.len[r] = row_len ? row_len[R.b[r]] : L;          // the loads below do not wait for it: masked rows are zeroed afterwards
            R.mu[r] = mean[rr];
            R.rs[r] = rstd[rr];
#pragma unroll
            for (int j = 0; j < MAXJ; ++j) {
                const int c = lane * 4 + 256 * j;
                if (c < D) {
                    R.d[r][j] = *reinterpret_cast<const f32x4*>(dy + rr * D + c);
                    R.xh[r][j] = *reinterpret_cast<const f32x4*>(s + rr * D + c);
                } else {
                    R.d[r][j] = f32x4{0, 0, 0, 0};
                    R.xh[r][j] = f32x4{0, 0, 0, 0};
                }
            }
        }
    };
    const int64_t stride = (int64_t)gridDim.x * LNB_ROWS;
    int64_t row0 = (int64_t)blockIdx.x * LNB_ROWS + wave * LNB_RPW;
    Rows cur;
    if (row0 < M) load_rows(cur, row0);
    for (; row0 < M; row0 += stride) {
    Rows nxt;
    const bool more = row0 + stride < M;
    if (more) load_rows(nxt, row0 + stride);
    uint32_t subx[LNB_RPW];
#pragma unroll
    for (int r = 0; r < LNB_RPW; ++r) {
        const bool keep = cur.live[r] && cur.t[r] < cur.len[r];
        subx[r] = drop_x.thr16 ? drop_subkey(drop_x, (uint32_t)cur.b[r]) : 0u;
        const uint32_t suby = drop_y.thr16 ? drop_subkey(drop_y, (uint32_t)cur.b[r]) : 0u;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = lane * 4 + 256 * j;
            if (!keep) cur.d[r][j] = f32x4{0, 0, 0, 0};
            if (drop_y.thr16 && c < D) cur.d[r][j] = drop4(drop_y, suby, (uint32_t)cur.t[r], D >> 1, c, cur.d[r][j], scy);
        }
    }
    // per-row reductions (independent chains) and outputs
    float s1[LNB_RPW], s2[LNB_RPW];
#pragma unroll
    for (int r = 0; r < LNB_RPW; ++r) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            // beta_y: `s` is the LayerNorm's OUTPUT y (the forward did not keep the pre-norm sum): x^ = (y - beta) / gamma
            if (beta_y) {
                const int c = lane * 4 + 256 * j;
                const f32x4 bt = (c < D) ? *reinterpret_cast<const f32x4*>(beta_y + c) : f32x4{0, 0, 0, 0};
                const bool keepr = cur.live[r] && cur.t[r] < cur.len[r];
#pragma unroll
                for (int e = 0; e < 4; ++e) cur.xh[r][j][e] = (keepr && gam[j][e] != 0.f) ? (cur.xh[r][j][e] - bt[e]) / gam[j][e] : 0.f;
            } else
                cur.xh[r][j] = (cur.xh[r][j] - cur.mu[r]) * cur.rs[r];
            ag[j] += cur.d[r][j] * cur.xh[r][j];
            ab[j] += cur.d[r][j];
            const f32x4 g = cur.d[r][j] * gam[j];
            const f32x4 gx = g * cur.xh[r][j];
            a1 += (g[0] + g[1]) + (g[2] + g[3]);
            a2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        }
        s1[r] = a1;
        s2[r] = a2;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int r = 0; r < LNB_RPW; ++r) {
            s1[r] += __shfl_xor(s1[r], o, 64);
            s2[r] += __shfl_xor(s2[r], o, 64);
        }
    }
#pragma unroll
    for (int r = 0; r < LNB_RPW; ++r) {
        if (!cur.live[r]) continue;
        const int64_t row = row0 + r;
        const float m1 = s1[r] * invD, m2 = s2[r] * invD;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int c = lane * 4 + 256 * j;
            if (c < D) {
                f32x4 o = (cur.d[r][j] * gam[j] - m1 - cur.xh[r][j] * m2) * cur.rs[r];
                *reinterpret_cast<f32x4*>(ds + row * D + c) = o;                                  // gradient wrt the residual
                if (drop_x.thr16) o = drop4(drop_x, subx[r], (uint32_t)cur.t[r], D >> 1, c, o, scx);   // gradient wrt x (dropout's input)
                as[j] += o;
                if (ds16) {
                    bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(ds16) + row * D + c) = ob;
                }
            }
        }
    }
    if (more) cur = nxt;
    }  // row loop
    // workgroup reduction of the dgamma / dbeta / dbias partials, then one atomic per column
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
        const int c = lane * 4 + 256 * j;
        *reinterpret_cast<f32x4*>(&red[0][wave][c]) = ag[j];
        *reinterpret_cast<f32x4*>(&red[1][wave][c]) = ab[j];
        *reinterpret_cast<f32x4*>(&red[2][wave][c]) = as[j];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 64 * LNB_WAVES) {
        float g = 0.f, bt = 0.f, bs = 0.f;
#pragma unroll
        for (int w = 0; w < LNB_WAVES; ++w) { g += red[0][w][c]; bt += red[1][w][c]; bs += red[2][w][c]; }
        atomicAdd(dgamma + c, g);
        atomicAdd(dbeta + c, bt);
        if (dbias) atomicAdd(dbias + c, bs);
    }
}

__global__ __launch_bounds__(256) void embed_bwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dy, int M, int U,
                                                        int D, int V, float* __restrict__ demb, asr_dropout_t drop_in) {
    const asr_dropout_t drop = drop_resolve(drop_in);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    const uint32_t b = (uint32_t)(row / U), u = (uint32_t)(row - (int64_t)b * U);
    const uint32_t sub = drop.thr16 ? drop_subkey(drop, b) : 0u;
    const float sc = drop_scale(drop);
    for (int c = lane; c < D; c += 64) {
        float g = dy[row * D + c];
        if (drop.thr16) {
            const uint32_t w = drop_word(drop, sub, u * (uint32_t)((D + 1) >> 1) + ((uint32_t)c >> 1));
            g = ((c & 1) ? drop_keep_hi(drop, w) : drop_keep_lo(drop, w)) ? g * sc : 0.f;
        }
        atomicAdd(demb + id * D + c, g);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, bf16_t* __restrict__ p16, int64_t n, float lr, float b1,
                                                   float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;      // torch.optim.Adam: sqrt(v)/sqrt(bias_correction2) + eps
        const float pn = p[i] - (lr / bc1) * (mi / denom);
        p[i] = pn;
        if (p16) p16[i] = (bf16_t)pn;                          // bf16 MFMA shadow refreshed in the same pass
    }
}

// ---- step state on the device (asr_hip.h: asr_step_tick) ----
__global__ void step_tick_kernel(uint32_t* state, float k, float init_lr, float warmup, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t step = state[0] + 1u;
    const double n = (double)step;
    const double lr = (double)k * (double)init_lr * fmin(1.0 / sqrt(n), n * pow((double)warmup, -1.5));
    state[0] = step;
    reinterpret_cast<float*>(state)[1] = (float)lr;
    reinterpret_cast<float*>(state)[2] = (float)(1.0 - pow((double)b1, n));
    reinterpret_cast<float*>(state)[3] = (float)sqrt(1.0 - pow((double)b2, n));
}
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, bf16_t* __restrict__ p16, int64_t n,
                                                       const uint32_t* __restrict__ state, float b1, float b2, float eps, float gscale) {
    const float lr = reinterpret_cast<const float*>(state)[1], bc1 = reinterpret_cast<const float*>(state)[2],
                bc2_sqrt = reinterpret_cast<const float*>(state)[3];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        const float pn = p[i] - (lr / bc1) * (mi / denom);
        p[i] = pn;
        if (p16) p16[i] = (bf16_t)pn;
    }
}

template <typename TA, typename TB>
int launch_tn(hipStream_t s, const void* A, int64_t lda, const void* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K) {
    const int tiles_n = (N + 127) / 128, tiles_k = (K + 127) / 128;
    const int tiles = tiles_n * tiles_k;
    int splits = (512 + tiles - 1) / tiles;
    const int max_splits = (M + 511) / 512;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    int m_per_split = ((M + splits - 1) / splits + 63) / 64 * 64;
    splits = (M + m_per_split - 1) / m_per_split;
    hipLaunchKernelGGL((gemm_tn_kernel<TA, TB>), dim3(tiles, splits), dim3(256), 0, s, (const TA*)A, lda, (const TB*)Bm, ldb, C, ldc, M, N,
                       K, tiles_k, m_per_split);
    ASR_LAUNCH_CHECK("gemm_tn");
    return 0;
}

}  // namespace

extern "C" int asr_colsum(void* stream, const void* A, int a_dtype, int64_t lda, int M, int N, float* out, int zero_first);

extern "C" int asr_gemm_tn(void* stream, const void* A, int a_dtype, int64_t lda, const void* Bm, int b_dtype, int64_t ldb, float* C,
                           int64_t ldc, int M, int N, int K, int zero_first, float* colsum, int max_workgroups) {
    ASR_REQUIRE(A && Bm && C && M > 0 && N > 0 && K > 0, ASR_ERR_ARG, "gemm_tn: bad args");
    // (no `ld >= width` requirement: overlapping row windows - conv1d as a GEMM, lda = C < K = w*C - are legitimate operands; the
    // caller guarantees that rows may be read 4-element-group-wise up to the rounded-up width)
    ASR_REQUIRE(lda % 4 == 0 && ldb % 4 == 0, ASR_ERR_ALIGN, "gemm_tn: lda/ldb must be multiples of 4");
    ASR_REQUIRE(asr_aligned(A, a_dtype == ASR_F32 ? 16 : 8) && asr_aligned(Bm, b_dtype == ASR_F32 ? 16 : 8), ASR_ERR_ALIGN,
                "gemm_tn: operand alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (zero_first) {
        if (ldc == K) {
            hipError_t e = hipMemsetAsync(C, 0, (size_t)N * K * sizeof(float), s);
            if (e != hipSuccess) { asr_set_error("gemm_tn memset: %s", hipGetErrorString(e)); return (int)e; }
        } else {
            hipError_t e = hipMemset2DAsync(C, ldc * sizeof(float), 0, (size_t)K * sizeof(float), N, s);
            if (e != hipSuccess) { asr_set_error("gemm_tn memset2d: %s", hipGetErrorString(e)); return (int)e; }
        }
    }
    // N need not be a multiple of 128 when A's rows can be READ up to the next one (lda covers it: a padded gradient buffer)
    const bool n_ok = N % 128 == 0 || lda >= (int64_t)(N + 127) / 128 * 128;
    if (a_dtype == ASR_BF16 && b_dtype == ASR_BF16 && M % 8 == 0 && M >= 64 && n_ok && K % 128 == 0 && lda % 8 == 0 &&
        ldb % 8 == 0 && asr_aligned(A, 16) && asr_aligned(Bm, 16)) {
        const int tiles_n = (N + 127) / 128, tiles_k = K / 128, tiles = tiles_n * tiles_k;
        // 512 = 2 resident per CU when the kernel has the chip to itself; a caller that runs it BESIDE other kernels (the trainer's
        // weight-gradient stream) asks for 256: one per CU and half the M-splits (half the atomic epilogue) - 14.54 -> 14.18 ms per step
        const int target_wgs = max_workgroups > 0 ? max_workgroups : 512;
        int splits = (target_wgs + tiles - 1) / tiles;
        constexpr int min_rows = 512;
        const int max_splits = (M + min_rows - 1) / min_rows;
        if (splits > max_splits) splits = max_splits;
        if (splits < 1) splits = 1;
        if (splits >= 8) splits = splits / 8 * 8;          // whole M-ranges per XCD (see the kernel's id mapping)
        const int m_per_split = ((M + splits - 1) / splits + 63) / 64 * 64;
        if (splits < 8) splits = (M + m_per_split - 1) / m_per_split;   // (with >= 8 the count stays a multiple of 8; empty ranges exit)
        hipLaunchKernelGGL(gemm_tn_tr_kernel, dim3(tiles * splits), dim3(256), 0, s, (const bf16_t*)A, lda, (const bf16_t*)Bm, ldb, C, ldc,
                           M, N, K, tiles_k, m_per_split, splits, colsum);
        ASR_LAUNCH_CHECK("gemm_tn_tr");
        return 0;
    }
    if (colsum) {   // shapes the fused path does not cover: separate column-sum kernel (accumulating)
        if (int rc = asr_colsum(stream, A, a_dtype, lda, M, N, colsum, 0)) return rc;
    }
    if (a_dtype == ASR_F32 && b_dtype == ASR_F32) return launch_tn<float, float>(s, A, lda, Bm, ldb, C, ldc, M, N, K);
    if (a_dtype == ASR_F32 && b_dtype == ASR_BF16) return launch_tn<float, bf16_t>(s, A, lda, Bm, ldb, C, ldc, M, N, K);
    if (a_dtype == ASR_BF16 && b_dtype == ASR_F32) return launch_tn<bf16_t, float>(s, A, lda, Bm, ldb, C, ldc, M, N, K);
    return launch_tn<bf16_t, bf16_t>(s, A, lda, Bm, ldb, C, ldc, M, N, K);
}

extern "C" int asr_colsum(void* stream, const void* A, int a_dtype, int64_t lda, int M, int N, float* out, int zero_first) {
    ASR_REQUIRE(A && out && M > 0 && N > 0, ASR_ERR_ARG, "colsum: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (zero_first) {
        hipError_t e = hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s);
        if (e != hipSuccess) { asr_set_error("colsum memset: %s", hipGetErrorString(e)); return (int)e; }
    }
    const int col_blocks = (N + 255) / 256;
    int row_blocks = (1024 + col_blocks - 1) / col_blocks;          // ~1024 workgroups
    if (row_blocks > (M + 31) / 32) row_blocks = (M + 31) / 32;
    if (lda == N && N % 4 == 0 && N <= 128 && 256 % (N / 4) == 0 && asr_aligned(A, 16)) {
        const int cpr = N / 4;
        const int64_t nchunks = (int64_t)M * cpr;
        int64_t nb = (nchunks + 256 * 8 - 1) / (256 * 8);
        if (nb > 1024) nb = 1024;
        if (nb < 1) nb = 1;
        if (a_dtype == ASR_F32)
            hipLaunchKernelGGL(colsum_narrow_kernel<float>, dim3((unsigned)nb), dim3(256), 0, s, (const float*)A, nchunks, cpr, out);
        else
            hipLaunchKernelGGL(colsum_narrow_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, s, (const bf16_t*)A, nchunks, cpr, out);
        ASR_LAUNCH_CHECK("colsum_narrow");
        return 0;
    }
    const int rows_per_block = (M + row_blocks - 1) / row_blocks;
    dim3 grid(col_blocks, (M + rows_per_block - 1) / rows_per_block);
    if (a_dtype == ASR_F32)
        hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)A, lda, M, N, rows_per_block, out);
    else
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)A, lda, M, N, rows_per_block, out);
    ASR_LAUNCH_CHECK("colsum");
    return 0;
}

extern "C" int asr_add_layernorm_bwd(void* stream, const float* dy, const float* s, const float* mean, const float* rstd,
                                     const float* gamma, const int32_t* row_len, float* ds, void* ds16, float* dgamma, float* dbeta,
                                     float* dbias, int B, int L, int D, asr_dropout_t drop_x, asr_dropout_t drop_y) {
    ASR_REQUIRE(dy && s && mean && rstd && gamma && ds && dgamma && dbeta, ASR_ERR_ARG, "layernorm_bwd: null pointer");
    ASR_REQUIRE(drop_x.thr16 < 65536u && drop_y.thr16 < 65536u, ASR_ERR_ARG, "layernorm_bwd: dropout thr16 must be < 65536");
    ASR_REQUIRE(B > 0 && L > 0 && D > 0 && D <= 1024 && D % 4 == 0, ASR_ERR_UNSUPPORTED, "layernorm_bwd: D=%d", D);
    const int M = B * L;
    int blocks = (M + LNB_ROWS - 1) / LNB_ROWS;
    constexpr int max_blocks = LNB_MAX_BLOCKS;      // persistent workgroups (tools/abl_lnb.sh, alone on the chip: 256 / 192 / 128 / 96 / 64 -> 23.7 / 24.3 / 28.8 / 33.4 / 46.1 us at [32000 x 256], 18.0 / 17.5 / 18.5 / 20.7 / 26.3 us at [10688 x 512])
    if (blocks > max_blocks) blocks = max_blocks;
    if (D <= 256)
        hipLaunchKernelGGL(add_layernorm_bwd_kernel<1>, dim3(blocks), dim3(64 * LNB_WAVES), 0, static_cast<hipStream_t>(stream), dy, s, mean, rstd, gamma,
                           row_len, ds, ds16, dgamma, dbeta, dbias, M, L, D, drop_x, drop_y);
    else if (D <= 512)      // d_model = 512, the width every shipped recipe of the reference trains (egs/aishell/recipes/*.sh): no spills
        hipLaunchKernelGGL(add_layernorm_bwd_kernel<2>, dim3(blocks), dim3(64 * LNB_WAVES), 0, static_cast<hipStream_t>(stream), dy, s, mean, rstd, gamma,
                           row_len, ds, ds16, dgamma, dbeta, dbias, M, L, D, drop_x, drop_y);
    else
        hipLaunchKernelGGL(add_layernorm_bwd_kernel<4>, dim3(blocks), dim3(64 * LNB_WAVES), 0, static_cast<hipStream_t>(stream), dy, s, mean, rstd, gamma,
                           row_len, ds, ds16, dgamma, dbeta, dbias, M, L, D, drop_x, drop_y);
    ASR_LAUNCH_CHECK("add_layernorm_bwd");
    return 0;
}

// asr_add_layernorm_bwd for a forward that kept the LayerNorm's OUTPUT instead of its pre-norm sum (the output stays alive anyway: it is
// the next sub-layer's input and residual): x^ = (y - beta) / gamma (0 where gamma is 0 and in masked rows), everything else as above.
extern "C" int asr_add_layernorm_bwd_y(void* stream, const float* dy, const float* y, const float* rstd, const float* gamma,
                                       const float* beta, const int32_t* row_len, float* ds, void* ds16, float* dgamma, float* dbeta,
                                       float* dbias, int B, int L, int D, asr_dropout_t drop_x) {
    ASR_REQUIRE(dy && y && rstd && gamma && beta && ds && dgamma && dbeta, ASR_ERR_ARG, "layernorm_bwd_y: null pointer");
    ASR_REQUIRE(drop_x.thr16 < 65536u, ASR_ERR_ARG, "layernorm_bwd_y: dropout thr16 must be < 65536");
    ASR_REQUIRE(B > 0 && L > 0 && D > 0 && D <= 1024 && D % 4 == 0, ASR_ERR_UNSUPPORTED, "layernorm_bwd_y: D=%d", D);
    const int M = B * L;
    int blocks = (M + LNB_ROWS - 1) / LNB_ROWS;
    constexpr int max_blocks = LNB_MAX_BLOCKS;      // persistent workgroups (tools/abl_lnb.sh, alone on the chip: 256 / 192 / 128 / 96 / 64 -> 23.7 / 24.3 / 28.8 / 33.4 / 46.1 us at [32000 x 256], 18.0 / 17.5 / 18.5 / 20.7 / 26.3 us at [10688 x 512])
    if (blocks > max_blocks) blocks = max_blocks;
    const asr_dropout_t none{0, 0, 0, nullptr};
    if (D <= 256)
        hipLaunchKernelGGL(add_layernorm_bwd_kernel<1>, dim3(blocks), dim3(64 * LNB_WAVES), 0, static_cast<hipStream_t>(stream), dy, y, rstd, rstd, gamma,
                           row_len, ds, ds16, dgamma, dbeta, dbias, M, L, D, drop_x, none, beta);
    else if (D <= 512)
        hipLaunchKernelGGL(add_layernorm_bwd_kernel<2>, dim3(blocks), dim3(64 * LNB_WAVES), 0, static_cast<hipStream_t>(stream), dy, y, rstd, rstd, gamma,
                           row_len, ds, ds16, dgamma, dbeta, dbias, M, L, D, drop_x, none, beta);
    else
        hipLaunchKernelGGL(add_layernorm_bwd_kernel<4>, dim3(blocks), dim3(64 * LNB_WAVES), 0, static_cast<hipStream_t>(stream), dy, y, rstd, rstd, gamma,
                           row_len, ds, ds16, dgamma, dbeta, dbias, M, L, D, drop_x, none, beta);
    ASR_LAUNCH_CHECK("add_layernorm_bwd_y");
    return 0;
}

extern "C" int asr_embed_bwd(void* stream, const int64_t* ids, const float* dy, int B, int U, int D, int V, float* demb,
                             asr_dropout_t drop) {
    ASR_REQUIRE(ids && dy && demb && B > 0 && U > 0 && D > 0 && V > 0 && drop.thr16 < 65536u, ASR_ERR_ARG, "embed_bwd: bad args");
    const int M = B * U;
    hipLaunchKernelGGL(embed_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), ids, dy, M, U, D, V, demb,
                       drop);
    ASR_LAUNCH_CHECK("embed_bwd");
    return 0;
}

extern "C" int asr_adam_step(void* stream, float* p, const float* g, float* m, float* v, void* p16, int64_t n, float lr, float beta1,
                             float beta2, float eps, int step, float grad_scale) {
    ASR_REQUIRE(p && g && m && v && n > 0 && step >= 1, ASR_ERR_ARG, "adam: bad args");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v,
                       reinterpret_cast<bf16_t*>(p16), n, lr, beta1, beta2, eps, (float)bc1, (float)sqrt(bc2), grad_scale);
    ASR_LAUNCH_CHECK("adam_step");
    return 0;
}

extern "C" int asr_step_tick(void* stream, uint32_t* state, float k, float init_lr, float warmup, float beta1, float beta2) {
    ASR_REQUIRE(state && warmup > 0.f, ASR_ERR_ARG, "step_tick: bad args");
    hipLaunchKernelGGL(step_tick_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), state, k, init_lr, warmup, beta1, beta2);
    ASR_LAUNCH_CHECK("step_tick");
    return 0;
}

extern "C" int asr_adam_step_dev(void* stream, float* p, const float* g, float* m, float* v, void* p16, int64_t n,
                                 const uint32_t* state, float beta1, float beta2, float eps, float grad_scale) {
    ASR_REQUIRE(p && g && m && v && state && n > 0, ASR_ERR_ARG, "adam_dev: bad args");
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v,
                       reinterpret_cast<bf16_t*>(p16), n, state, beta1, beta2, eps, grad_scale);
    ASR_LAUNCH_CHECK("adam_step_dev");
    return 0;
}
