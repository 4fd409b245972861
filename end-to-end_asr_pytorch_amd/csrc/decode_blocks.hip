// Whole sub-layers of the per-token decode step in ONE launch each (Decoder.batch_decode / batch_beam_decode,
// Decoder_CIF.recognize_beam: src/transformer/decoder.py:138-234, 425-552 call PositionwiseFeedForward `module.py:48-53` and
// MultiheadAttention `attention.py:33-62` once per layer and token).
//
// The step feeds M = B (or B * beam) <= a few hundred rows through 6 layers.  As separate GEMM / attention / LayerNorm launches that
// is ~12 dependent kernels per layer, each 4.5 us of dispatch for 1-6 us of work (rocprofv3: 76 nodes of 5-11 us per token).  Here a
// sub-layer is one grid: the workgroups of a row block split the work (hidden units / heads), add their partial output rows with
// float atomics into a zeroed accumulator, and the LAST one to arrive (a counter per row block) applies bias + residual + LayerNorm,
// writes the outputs and leaves accumulator and counter zeroed for the next launch.  (Measured alternatives, S1 greedy decode of 32
// utterances x 50 tokens: separate launches 32.4 ms; per-part slabs summed by the last arrival in a fixed order - plain stores
// between __threadfence() 31.8 ms, write-through 8-byte stores / agent-scope loads 30.5 ms; float atomics 27.0 ms.)  bf16 MFMA 16x16x32, operands straight from L2
// (the weights of a layer are 0.5-2 MiB; every row block reads all of them - right for <= ~20 row blocks, wrong for the training
// shapes, which keep the tiled GEMMs).  d_model = 256 only.
#include "asr_common.h"

namespace {

constexpr int DM = 256;

__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mean and 1 / std of row r16 over its 256 columns (a lane holds v[j]: columns 64 wave + 16 j + 4 q4 .. + 3); two workgroup barriers
__device__ __forceinline__ void row_stats(const f32x4 (&v)[4], float eps, float (&red)[2][16][4], float& mean, float& rstd) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q4 = lane >> 4;
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (q4 == 0) red[0][r16][wave] = sum;
    __syncthreads();
    mean = ((red[0][r16][0] + red[0][r16][1]) + (red[0][r16][2] + red[0][r16][3])) * (1.f / DM);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 d = v[j] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (q4 == 0) red[1][r16][wave] = q;
    __syncthreads();
    const float var = ((red[1][r16][0] + red[1][r16][1]) + (red[1][r16][2] + red[1][r16][3])) * (1.f / DM);
    rstd = 1.0f / sqrtf(var + eps);
}

// bias + residual + LayerNorm of a row block's accumulated rows by its last workgroup: lane (r16, q4) of wave w owns row r16,
// columns 64 w + 16 j + 4 q4 .. + 3 (gemm_ln_small_kernel's epilogue); re-zeroes the accumulator rows and the counter
__device__ __forceinline__ void finish_rows(float* __restrict__ accbuf, int* __restrict__ counter, const float* __restrict__ bias,
                                            const float* __restrict__ res, const f32x4* resv, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, float* __restrict__ y32, bf16_t* __restrict__ y16, int row0,
                                            int rows, int M, float eps, float (&red)[2][16][4], bf16_t* ylds = nullptr) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q4 = lane >> 4;
    const int row = row0 + r16;
    const bool live = r16 < rows && row < M;
    const int rr = live ? row : min(row0, M - 1);
    f32x4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = wave * 64 + 16 * j + 4 * q4;
        const float* ap = accbuf + (int64_t)rr * DM + c;
        v[j] = f32x4{ld_agent(ap), ld_agent(ap + 1), ld_agent(ap + 2), ld_agent(ap + 3)};
        if (bias) v[j] += *reinterpret_cast<const f32x4*>(bias + c);
        v[j] += resv ? resv[j] : *reinterpret_cast<const f32x4*>(res + (int64_t)rr * DM + c);
    }
    float mean, rstd;
    row_stats(v, eps, red, mean, rstd);
    if (live) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wave * 64 + 16 * j + 4 * q4;
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
            const f32x4 o = (v[j] - mean) * rstd * g + bt;
            *reinterpret_cast<f32x4*>(y32 + (int64_t)row * DM + c) = o;
            if (y16) *reinterpret_cast<bf16x4*>(y16 + (int64_t)row * DM + c) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            *reinterpret_cast<f32x4*>(accbuf + (int64_t)row * DM + c) = f32x4{0, 0, 0, 0};
            if (ylds) *reinterpret_cast<bf16x4*>(ylds + r16 * (DM + 8) + c) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
        }
    } else if (ylds) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<bf16x4*>(ylds + r16 * (DM + 8) + wave * 64 + 16 * j + 4 * q4) = bf16x4{(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
    }
    if (threadIdx.x == 0) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// adds this workgroup's partial rows (acc[j]: row r16, columns 64 wave + 16 j + 4 q4 .. + 3) and tells whether it was the last of the
// row block's `n_parts` to arrive
__device__ __forceinline__ bool add_partial(float* __restrict__ accbuf, int* __restrict__ counter, const f32x4 (&acc)[4], int row0, int rows,
                                            int M, int n_parts, int& last_flag) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q4 = lane >> 4;
    if (r16 < rows && row0 + r16 < M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* ap = accbuf + (int64_t)(row0 + r16) * DM + wave * 64 + 16 * j + 4 * q4;
#pragma unroll
            for (int x = 0; x < 4; ++x) atomicAdd(ap + x, acc[j][x]);
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) last_flag = (atomicAdd(counter, 1) == n_parts - 1) ? 1 : 0;
    __syncthreads();
    const bool last = last_flag != 0;
    if (last) __threadfence();
    return last;
}

// y = LayerNorm(relu(x W1^T + b1) W2^T + b2 + x): grid (d_ff / 128, row blocks); a workgroup owns 128 hidden units of 16 rows.
// PRE: x itself is the end of the attention sub-layer in front, x = LayerNorm0(ctx Wo^T + bo + res) (attention.py:58-60), computed
// by EVERY workgroup of the row block for its 16 rows (16 x 256 x 256: cheaper than the launch it replaces) - ctx is the attention
// output [M, 256] bf16, res the attention sub-layer's input.
template <bool PRE>
__global__ __launch_bounds__(256) void decode_ffn_kernel(const bf16_t* __restrict__ x16, const float* __restrict__ x32,
                                                         const bf16_t* __restrict__ Wo, const float* __restrict__ bo,
                                                         const float* __restrict__ gamma0, const float* __restrict__ beta0, float eps0,
                                                         const bf16_t* __restrict__ W1, const float* __restrict__ b1,
                                                         const bf16_t* __restrict__ W2, const float* __restrict__ b2,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ accbuf, int* __restrict__ counters, float* __restrict__ y32,
                                                         bf16_t* __restrict__ y16, int M, int d_ff, float eps) {
    __shared__ __attribute__((aligned(16))) bf16_t H[16][128 + 8];
    __shared__ __attribute__((aligned(16))) bf16_t X[PRE ? 16 : 1][DM + 8];
    __shared__ float red[2][16][4];
    __shared__ int last_flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q4 = lane >> 4;
    const int hs = blockIdx.x, row0 = blockIdx.y * 16;
    const int arow = min(row0 + r16, M - 1);
    u32x4 ar[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) ar[s] = *reinterpret_cast<const u32x4*>(x16 + (int64_t)arow * DM + s * 32 + q4 * 8);     // (PRE: ctx rows)
    f32x4 xres[4];
    if constexpr (PRE) {
        f32x4 a0[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0[j] = f32x4{0, 0, 0, 0};
            u32x4 wr[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) wr[s] = *reinterpret_cast<const u32x4*>(Wo + (int64_t)(wave * 64 + 16 * j + r16) * DM + s * 32 + q4 * 8);
#pragma unroll
            for (int s = 0; s < 8; ++s) Mma<bf16_t>::run(wr[s], ar[s], a0[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wave * 64 + 16 * j + 4 * q4;
            a0[j] += *reinterpret_cast<const f32x4*>(x32 + (int64_t)arow * DM + c);
            if (bo) a0[j] += *reinterpret_cast<const f32x4*>(bo + c);
        }
        float mean, rstd;
        row_stats(a0, eps0, red, mean, rstd);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wave * 64 + 16 * j + 4 * q4;
            xres[j] = (a0[j] - mean) * rstd * *reinterpret_cast<const f32x4*>(gamma0 + c) + *reinterpret_cast<const f32x4*>(beta0 + c);
            *reinterpret_cast<bf16x4*>(&X[r16][c]) = bf16x4{(bf16_t)xres[j][0], (bf16_t)xres[j][1], (bf16_t)xres[j][2], (bf16_t)xres[j][3]};
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 8; ++s) ar[s] = *reinterpret_cast<const u32x4*>(&X[r16][s * 32 + q4 * 8]);
    }
    // hidden slice: wave w computes hidden units 128 hs + 32 w + 16 j + (4 q4 + reg), j = 0, 1
    f32x4 a1[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    u32x4 w1r[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 8; ++s)
            w1r[j][s] = *reinterpret_cast<const u32x4*>(W1 + (int64_t)(hs * 128 + wave * 32 + 16 * j + r16) * DM + s * 32 + q4 * 8);
    // the second GEMM's weights are independent of the first: fetch them now, under the first GEMM and the LDS round trip
    u32x4 w2r[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            w2r[j][s] = *reinterpret_cast<const u32x4*>(W2 + (int64_t)(wave * 64 + 16 * j + r16) * d_ff + hs * 128 + s * 32 + q4 * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 8; ++s) Mma<bf16_t>::run(w1r[j][s], ar[s], a1[j]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = wave * 32 + 16 * j + 4 * q4;
        f32x4 v = a1[j] + *reinterpret_cast<const f32x4*>(b1 + hs * 128 + c);
        *reinterpret_cast<bf16x4*>(&H[r16][c]) = bf16x4{(bf16_t)fmaxf(v[0], 0.f), (bf16_t)fmaxf(v[1], 0.f), (bf16_t)fmaxf(v[2], 0.f),
                                                        (bf16_t)fmaxf(v[3], 0.f)};
    }
    __syncthreads();
    f32x4 a2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a2[j] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const u32x4 hfrag = *reinterpret_cast<const u32x4*>(&H[r16][s * 32 + q4 * 8]);
#pragma unroll
        for (int j = 0; j < 4; ++j) Mma<bf16_t>::run(w2r[j][s], hfrag, a2[j]);
    }
    if (!add_partial(accbuf, counters + blockIdx.y, a2, row0, 16, M, (int)gridDim.x, last_flag)) return;
    finish_rows(accbuf, counters + blockIdx.y, b2, x32, PRE ? xres : nullptr, gamma, beta, y32, y16, row0, 16, M, eps, red);
}

// Self-attention of the ONE new position of every row against that row's K / V cache, with the projections around it:
//   q, k, v = x Wq^T + bq, x Wk^T + bk, x Wv^T + bv;  cache[row, head, t] = k, v;  p = softmax(q . K[0..t] / sqrt(64));
//   y = LayerNorm((p V) Wo^T + bo + x)                                                            (attention.py:33-62 for Lq = 1)
// grid (heads, row blocks): a workgroup owns ONE head of 16 rows - its 3 x 64 projection columns, the cache write, the attention
// of its 16 (row, head) pairs (wave w: rows 4 w .. 4 w + 3; scores with a lane per cached position, the weighted sum with a lane
// per head dimension), and that head's 64-column slice of the output projection as a partial of all 256 output columns.
// caches [N, h, Tmax, 64] bf16; the position t is read from state[0].
template <int ROWS>   // rows of a block: 16, or 4 (one per wave - four times the workgroups and a quarter of the serial attention walk)
__global__ __launch_bounds__(256) void decode_self_attn_kernel(const bf16_t* __restrict__ x16, const float* __restrict__ x32,
                                                               const bf16_t* __restrict__ Wqkv, const float* __restrict__ bqkv,
                                                               const bf16_t* __restrict__ Wo, const float* __restrict__ bo,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               bf16_t* __restrict__ k_cache, bf16_t* __restrict__ v_cache,
                                                               const int32_t* __restrict__ state, float* __restrict__ accbuf,
                                                               int* __restrict__ counters, float* __restrict__ y32, bf16_t* __restrict__ y16,
                                                               int M, int h, int Tmax, float eps, const bf16_t* __restrict__ Wq2,
                                                               const float* __restrict__ bq2, bf16_t* __restrict__ q2_out, int Lq2, int h2,
                                                               float q2_scale) {
    __shared__ float qs[16][64], ks[16][64], vs[16][64];          // this head's new q (scaled), k, v per row
    __shared__ __attribute__((aligned(16))) bf16_t O[16][64 + 8]; // attention output rows, the output projection's A operand
    __shared__ float red[2][16][4];
    __shared__ __attribute__((aligned(16))) bf16_t Kst[4][64][64 + 8];   // per wave: a 64-position chunk of the current row's cached keys ...
    __shared__ __attribute__((aligned(16))) bf16_t Vst[4][64][64 + 8];   // ... and values (rows padded by 16 bytes: conflict-free row reads)
    __shared__ int last_flag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q4 = lane >> 4;
    const int head = blockIdx.x, row0 = blockIdx.y * ROWS;
    const int t = min(max(state[0], 0), Tmax - 1);
    const int arow = min(row0 + min(r16, ROWS - 1), M - 1);
    u32x4 ar[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) ar[s] = *reinterpret_cast<const u32x4*>(x16 + (int64_t)arow * DM + s * 32 + q4 * 8);
    // projections: wave w computes columns 16 w .. 16 w + 15 of this head's q, k and v (Wqkv rows: [q | k | v] x (h * 64))
    f32x4 pa[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        pa[g] = f32x4{0, 0, 0, 0};
        const bf16_t* wp = Wqkv + (int64_t)(g * h * 64 + head * 64 + wave * 16 + r16) * DM + q4 * 8;
        u32x4 wr[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) wr[s] = *reinterpret_cast<const u32x4*>(wp + s * 32);
#pragma unroll
        for (int s = 0; s < 8; ++s) Mma<bf16_t>::run(wr[s], ar[s], pa[g]);
    }
    {   // lane: row r16, columns 16 wave + 4 q4 + reg of the head
        const int c = wave * 16 + 4 * q4;
        const f32x4 qv = (pa[0] + *reinterpret_cast<const f32x4*>(bqkv + head * 64 + c)) * 0.125f;      // 1 / sqrt(d_k = 64)
        const f32x4 kv = pa[1] + *reinterpret_cast<const f32x4*>(bqkv + h * 64 + head * 64 + c);
        const f32x4 vv = pa[2] + *reinterpret_cast<const f32x4*>(bqkv + 2 * h * 64 + head * 64 + c);
        *reinterpret_cast<f32x4*>(&qs[r16][c]) = qv;
        // the cached copies are bf16 (what later steps read): this step attends to the same rounded values
        const bf16x4 kb = {(bf16_t)kv[0], (bf16_t)kv[1], (bf16_t)kv[2], (bf16_t)kv[3]};
        const bf16x4 vb = {(bf16_t)vv[0], (bf16_t)vv[1], (bf16_t)vv[2], (bf16_t)vv[3]};
        *reinterpret_cast<f32x4*>(&ks[r16][c]) = f32x4{(float)kb[0], (float)kb[1], (float)kb[2], (float)kb[3]};
        *reinterpret_cast<f32x4*>(&vs[r16][c]) = f32x4{(float)vb[0], (float)vb[1], (float)vb[2], (float)vb[3]};
        if (r16 < ROWS && row0 + r16 < M) {
            const int64_t slot = (((int64_t)(row0 + r16) * h + head) * Tmax + t) * 64 + c;
            *reinterpret_cast<bf16x4*>(k_cache + slot) = kb;
            *reinterpret_cast<bf16x4*>(v_cache + slot) = vb;
        }
    }
    __syncthreads();
    // attention: wave w, rows 4 w + i; per 64-position chunk the cached rows are staged through LDS with 16-byte loads (one latency
    // round trip per chunk), scores with a lane per position, the weighted sum with a lane per head dimension
    for (int i = 0; i < ROWS / 4; ++i) {
        const int lr = wave * (ROWS / 4) + i, row = min(row0 + lr, M - 1);
        const bf16_t* kc = k_cache + ((int64_t)row * h + head) * Tmax * 64;
        const bf16_t* vc = v_cache + ((int64_t)row * h + head) * Tmax * 64;
        float m_run = -INFINITY, l_run = 0.f, o_acc = 0.f;               // o_acc: dimension `lane` of the output
        for (int p0 = 0; p0 <= t; p0 += 64) {
            const int n_old = min(64, t - p0);                           // cached positions of this chunk (position t comes from ks / vs)
            u32x4 kr[8], vr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {                                // piece u * 64 + lane: position (piece >> 3), 16-byte column (piece & 7)
                const int piece = u * 64 + lane, pp = piece >> 3, cc = piece & 7;
                if (pp < n_old) {
                    kr[u] = *reinterpret_cast<const u32x4*>(kc + (int64_t)(p0 + pp) * 64 + cc * 8);
                    vr[u] = *reinterpret_cast<const u32x4*>(vc + (int64_t)(p0 + pp) * 64 + cc * 8);
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int piece = u * 64 + lane, pp = piece >> 3, cc = piece & 7;
                if (pp < n_old) {
                    *reinterpret_cast<u32x4*>(&Kst[wave][pp][cc * 8]) = kr[u];
                    *reinterpret_cast<u32x4*>(&Vst[wave][pp][cc * 8]) = vr[u];
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int pos = p0 + lane;
            float sc = -INFINITY;
            if (lane < n_old) {
                float d = 0.f;
#pragma unroll
                for (int c8 = 0; c8 < 8; ++c8) {
                    const bf16x8 kb = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(&Kst[wave][lane][c8 * 8]));
#pragma unroll
                    for (int e = 0; e < 8; ++e) d += qs[lr][c8 * 8 + e] * (float)kb[e];
                }
                sc = d;
            } else if (pos == t) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 64; ++e) d += qs[lr][e] * ks[lr][e];
                sc = d;
            }
            float cm = sc;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) cm = fmaxf(cm, __shfl_xor(cm, o, 64));
            const float m_new = fmaxf(m_run, cm);
            const float pr = pos <= t ? __expf(sc - m_new) : 0.f;
            float ps = pr;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) ps += __shfl_xor(ps, o, 64);
            const float resc = __expf(m_run - m_new);                   // 0 on the first chunk (m_run = -inf)
            l_run = l_run * resc + ps;
            o_acc *= resc;
            const int prb = __builtin_bit_cast(int, pr);
            for (int j = 0; j < n_old; ++j)
                o_acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(prb, j)) * (float)Vst[wave][j][lane];
            if (t - p0 < 64) o_acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(prb, t - p0)) * vs[lr][lane];
            m_run = m_new;
            __builtin_amdgcn_wave_barrier();                             // the staging buffers are rewritten by the next chunk / row
        }
        O[lr][lane] = (bf16_t)(o_acc / l_run);
    }
    __syncthreads();
    // output projection, this head's 64 input columns: wave w -> output columns 64 w + 16 j + r16, K = 64
    f32x4 a2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a2[j] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const u32x4 ofrag = *reinterpret_cast<const u32x4*>(&O[r16][s * 32 + q4 * 8]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32x4 w = *reinterpret_cast<const u32x4*>(Wo + (int64_t)(wave * 64 + 16 * j + r16) * (h * 64) + head * 64 + s * 32 + q4 * 8);
            Mma<bf16_t>::run(w, ofrag, a2[j]);
        }
    }
    if (!add_partial(accbuf, counters + blockIdx.y, a2, row0, ROWS, M, (int)gridDim.x, last_flag)) return;
    // (the last workgroup is past the attention: the key staging buffer is free to hold its normalised rows)
    bf16_t* ylds = Wq2 ? &Kst[0][0][0] : nullptr;
    finish_rows(accbuf, counters + blockIdx.y, bo, x32, nullptr, gamma, beta, y32, y16, row0, ROWS, M, eps, red, ylds);
    if (!Wq2) return;
    // the NEXT sub-layer's query projection (the decoder's cross attention, attention.py:41 for its one new position per row) on the
    // rows just normalised: q2 = (y Wq2^T + bq2) * q2_scale, stored head-major [M / Lq2, h2, Lq2, 64] like asr_proj_heads does
    __syncthreads();
    f32x4 qa[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qa[j] = f32x4{0, 0, 0, 0};
        u32x4 wr[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) wr[s] = *reinterpret_cast<const u32x4*>(Wq2 + (int64_t)(wave * 64 + 16 * j + r16) * DM + s * 32 + q4 * 8);
#pragma unroll
        for (int s = 0; s < 8; ++s) Mma<bf16_t>::run(wr[s], *reinterpret_cast<const u32x4*>(ylds + r16 * (DM + 8) + s * 32 + q4 * 8), qa[j]);
    }
    if (r16 < ROWS && row0 + r16 < M) {
        const int row = row0 + r16, b = row / Lq2, jq = row - b * Lq2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wave * 64 + 16 * j + 4 * q4;              // column = head2 * 64 + d
            const f32x4 v = (qa[j] + *reinterpret_cast<const f32x4*>(bq2 + c)) * q2_scale;
            bf16_t* dst = q2_out + (((int64_t)b * h2 + (c >> 6)) * Lq2 + jq) * 64 + (c & 63);
            *reinterpret_cast<bf16x4*>(dst) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        }
    }
}

}  // namespace

extern "C" int64_t asr_decode_block_workspace_bytes(int M) {     // accumulator rows (padded to 16) + one counter per 4 rows
    const int64_t rb = (M + 15) / 16;
    return rb * 16 * DM * (int64_t)sizeof(float) + (int64_t)((M + 3) / 4) * (int64_t)sizeof(int);
}

extern "C" int asr_decode_ffn(void* stream, const void* x16, const float* x32, const void* W1, const float* b1, const void* W2, const float* b2,
                              const float* gamma, const float* beta, void* workspace, float* y32, void* y16, int M, int d_model, int d_ff,
                              float eps, const void* pre_Wo, const float* pre_bo, const float* pre_gamma, const float* pre_beta, float pre_eps) {
    ASR_REQUIRE(x16 && x32 && W1 && b1 && W2 && gamma && beta && workspace && y32 && M > 0, ASR_ERR_ARG, "decode_ffn: null pointer / bad sizes");
    ASR_REQUIRE(d_model == DM && d_ff > 0 && d_ff % 128 == 0, ASR_ERR_UNSUPPORTED, "decode_ffn: d_model %d / d_ff %d (256 and a multiple of 128)",
                d_model, d_ff);
    ASR_REQUIRE(!pre_Wo || (pre_gamma && pre_beta), ASR_ERR_ARG, "decode_ffn: the attention-output prologue needs its LayerNorm weights");
    ASR_REQUIRE(asr_aligned(x16, 16) && asr_aligned(x32, 16) && asr_aligned(W1, 16) && asr_aligned(W2, 16) && asr_aligned(b1, 16) &&
                    (!b2 || asr_aligned(b2, 16)) && asr_aligned(gamma, 16) && asr_aligned(beta, 16) && asr_aligned(workspace, 16) &&
                    asr_aligned(y32, 16) && (!y16 || asr_aligned(y16, 8)) && asr_aligned(pre_Wo, 16) && asr_aligned(pre_bo, 16) &&
                    asr_aligned(pre_gamma, 16) && asr_aligned(pre_beta, 16), ASR_ERR_ALIGN, "decode_ffn: 16-byte alignment");
    const int rb = (M + 15) / 16;
    float* accbuf = static_cast<float*>(workspace);
    int* counters = reinterpret_cast<int*>(accbuf + (int64_t)rb * 16 * DM);
    if (pre_Wo)
        hipLaunchKernelGGL(decode_ffn_kernel<true>, dim3(d_ff / 128, rb), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)x16, x32,
                           (const bf16_t*)pre_Wo, pre_bo, pre_gamma, pre_beta, pre_eps, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, gamma, beta,
                           accbuf, counters, y32, (bf16_t*)y16, M, d_ff, eps);
    else
        hipLaunchKernelGGL(decode_ffn_kernel<false>, dim3(d_ff / 128, rb), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)x16, x32,
                           nullptr, nullptr, nullptr, nullptr, 0.f, (const bf16_t*)W1, b1, (const bf16_t*)W2, b2, gamma, beta, accbuf, counters,
                           y32, (bf16_t*)y16, M, d_ff, eps);
    ASR_LAUNCH_CHECK("decode_ffn");
    return 0;
}

extern "C" int asr_decode_self_attn(void* stream, const void* x16, const float* x32, const void* Wqkv, const float* bqkv, const void* Wo,
                                    const float* bo, const float* gamma, const float* beta, void* k_cache, void* v_cache, const int32_t* state,
                                    void* workspace, float* y32, void* y16, int M, int d_model, int h, int Tmax, float eps, const void* next_Wq,
                                    const float* next_bq, void* next_q, int next_Lq, float next_scale) {
    ASR_REQUIRE(x16 && x32 && Wqkv && bqkv && Wo && gamma && beta && k_cache && v_cache && state && workspace && y32 && M > 0 && Tmax > 0,
                ASR_ERR_ARG, "decode_self_attn: null pointer / bad sizes");
    ASR_REQUIRE(d_model == DM && h >= 1 && h <= 16, ASR_ERR_UNSUPPORTED, "decode_self_attn: d_model %d / %d heads (256, 1..16 heads of 64)", d_model, h);
    ASR_REQUIRE(!next_Wq || (next_bq && next_q && next_Lq > 0 && M % next_Lq == 0 && asr_aligned(next_Wq, 16) && asr_aligned(next_bq, 16) &&
                             asr_aligned(next_q, 8)), ASR_ERR_ARG, "decode_self_attn: the next query projection needs weight, bias, output and Lq | M");
    ASR_REQUIRE(asr_aligned(x16, 16) && asr_aligned(x32, 16) && asr_aligned(Wqkv, 16) && asr_aligned(Wo, 16) && asr_aligned(bqkv, 16) &&
                    (!bo || asr_aligned(bo, 16)) && asr_aligned(gamma, 16) && asr_aligned(beta, 16) && asr_aligned(workspace, 16) &&
                    asr_aligned(k_cache, 16) && asr_aligned(v_cache, 16) && asr_aligned(y32, 16) && (!y16 || asr_aligned(y16, 8)),
                ASR_ERR_ALIGN, "decode_self_attn: 16-byte alignment");
    const int rb = (M + 15) / 16;
    float* accbuf = static_cast<float*>(workspace);
    int* counters = reinterpret_cast<int*>(accbuf + (int64_t)rb * 16 * DM);
    if (M <= 64)      // few rows: 4-row blocks, one row per wave
        hipLaunchKernelGGL(decode_self_attn_kernel<4>, dim3(h, (M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)x16,
                           x32, (const bf16_t*)Wqkv, bqkv, (const bf16_t*)Wo, bo, gamma, beta, (bf16_t*)k_cache, (bf16_t*)v_cache, state, accbuf,
                           counters, y32, (bf16_t*)y16, M, h, Tmax, eps, (const bf16_t*)next_Wq, next_bq, (bf16_t*)next_q, next_Lq, DM / 64,
                           next_scale);
    else
        hipLaunchKernelGGL(decode_self_attn_kernel<16>, dim3(h, rb), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)x16, x32,
                           (const bf16_t*)Wqkv, bqkv, (const bf16_t*)Wo, bo, gamma, beta, (bf16_t*)k_cache, (bf16_t*)v_cache, state, accbuf,
                           counters, y32, (bf16_t*)y16, M, h, Tmax, eps, (const bf16_t*)next_Wq, next_bq, (bf16_t*)next_q, next_Lq, DM / 64,
                           next_scale);
    ASR_LAUNCH_CHECK("decode_self_attn");
    return 0;
}
