// Conv2dSubsample (conv_encoder.py:101-108) for gfx950.
//
// Layout: activations between conv layers are channel-LAST [B,T,F,32] so the 32 input channels of one tap are a
// contiguous 64-byte (bf16) / 128-byte (f32) MFMA K-slice.  Only the region of each layer that the final crop
// (freq[:ceil(D/2)], time[:ceil(T/2^n)]) can see is produced: the reference computes 86 frequency columns in the
// last layer and keeps 40.
//   layer 0 (1 -> 32 ch, K = 9): VALU, one thread per output position, weights in LDS, implicit zero right-pad.
//   layer i >= 1 (32 -> 32 ch): implicit GEMM on MFMA, K = 9 taps x 32 channels.  Weights (A operand, rows =
//     c_out) are gathered once per wave into registers; the B operand (16 consecutive output positions x 32 c_in)
//     is read straight from global/L2 per tap - neighbouring taps/positions hit the same lines in L1.
//     D[c_out][position] puts 4 consecutive output channels of one position in each lane.
#include "asr_common.h"

namespace {

template <typename CT>
__global__ __launch_bounds__(256) void conv_sub0_kernel(const float* __restrict__ feats, const float* __restrict__ w0,
                                                        const float* __restrict__ b0, CT* __restrict__ y, int B, int T, int D,
                                                        int T1, int F1) {
    __shared__ float ws[9][32];
    __shared__ float bs[32];
    for (int i = threadIdx.x; i < 288; i += 256) ws[i % 9][i / 9] = w0[i];  // w0[c][0][kh][kw] -> ws[tap][c]
    if (threadIdx.x < 32) bs[threadIdx.x] = b0[threadIdx.x];
    __syncthreads();
    const int64_t total = (int64_t)B * T1 * F1;
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= total) return;
    const int f = (int)(pos % F1);
    const int t1 = (int)((pos / F1) % T1);
    const int b = (int)(pos / ((int64_t)F1 * T1));
    float x[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int tt = 2 * t1 + kh, ff = f + kw;
            x[kh * 3 + kw] = (tt < T && ff < D) ? feats[((int64_t)b * T + tt) * D + ff] : 0.f;
        }
    CT* yp = y + pos * 32;
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += 4) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = bs[c0 + i];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc = fmaf(ws[tap][c0 + i], x[tap], acc);
            o[i] = fmaxf(acc, 0.f);
        }
        if constexpr (sizeof(CT) == 4) {
            *reinterpret_cast<f32x4*>(yp + c0) = o;
        } else {
            bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            *reinterpret_cast<bf16x4*>(yp + c0) = ob;
        }
    }
}

template <typename CT> struct ConvK {  // 16-byte chunks per 32-channel tap slice
    static constexpr int CH = 16 / sizeof(CT);      // elements per chunk
    static constexpr int GROUPS = 32 / (4 * CH);    // 4-chunk MFMA groups per tap: bf16 1, f32 2
};

template <typename CT>
__global__ __launch_bounds__(256) void conv_sub1_kernel(const CT* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, CT* __restrict__ y, int B, int Tin,
                                                        int Fin, int Tout, int Fout, int last, int n_tiles) {
    constexpr int CH = ConvK<CT>::CH, G = ConvK<CT>::GROUPS;
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;

    // weights as the MFMA row operand: wf[tap][ct][g] = chunk (g*4+q4) of w[c_out = ct*16 + r16][c_in][tap]
    u32x4 wf[9][2][G];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int co = ct * 16 + r16;
                const int ci0 = (g * 4 + q4) * CH;
                if constexpr (sizeof(CT) == 4) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = w[(co * 32 + ci0 + j) * 9 + tap];
                    wf[tap][ct][g] = __builtin_bit_cast(u32x4, v);
                } else {
                    bf16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)w[(co * 32 + ci0 + j) * 9 + tap];
                    wf[tap][ct][g] = __builtin_bit_cast(u32x4, v);
                }
            }
    f32x4 bv[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(bias + ct * 16 + q4 * 4);

    const int64_t total = (int64_t)B * Tout * Fout;
    for (int tile = gwave; tile < n_tiles; tile += nwaves) {
        int64_t pos = (int64_t)tile * 16 + r16;
        const bool ok = pos < total;
        if (!ok) pos = total - 1;
        const int f = (int)(pos % Fout);
        const int t = (int)((pos / Fout) % Tout);
        const int b = (int)(pos / ((int64_t)Fout * Tout));
        f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const CT* xp = x + (((int64_t)b * Tin + 2 * t + kh) * Fin + f + kw) * 32;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const u32x4 xv = *reinterpret_cast<const u32x4*>(xp + (g * 4 + q4) * CH);
                    Mma<CT>::run(wf[kh * 3 + kw][0][g], xv, acc[0]);
                    Mma<CT>::run(wf[kh * 3 + kw][1][g], xv, acc[1]);
                }
            }
        if (!ok) continue;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            f32x4 o = acc[ct] + bv[ct];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], 0.f);
            const int c0 = ct * 16 + q4 * 4;
            if (last) {
                CT* yp = y + ((int64_t)b * Tout + t) * (32 * Fout) + f;
#pragma unroll
                for (int i = 0; i < 4; ++i) yp[(int64_t)(c0 + i) * Fout] = from_f32<CT>(o[i]);
            } else {
                CT* yp = y + pos * 32 + c0;
                if constexpr (sizeof(CT) == 4) {
                    *reinterpret_cast<f32x4*>(yp) = o;
                } else {
                    bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                    *reinterpret_cast<bf16x4*>(yp) = ob;
                }
            }
        }
    }
}

// ---- backward helpers: the conv layers' gradients are expressed as GEMMs over an explicit patch matrix ------------------
// im2col: col[(b,t,f), tap*C + c] = x[b, 2t+kh, f+kw, c] (zero outside [Tin) x [Fin)); columns 9*C..ldc-1 are zeroed.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void conv_im2col_kernel(const TI* __restrict__ x, TO* __restrict__ col, int C, int Tin, int Fin,
                                                          int Tout, int Fout, int ldc, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int j = (int)(i % ldc);
        const int64_t pos = i / ldc;
        float v = 0.f;
        if (j < 9 * C) {
            const int tap = j / C, c = j - tap * C, kh = tap / 3, kw = tap - kh * 3;
            const int f = (int)(pos % Fout), t = (int)((pos / Fout) % Tout), b = (int)(pos / ((int64_t)Fout * Tout));
            const int tt = 2 * t + kh, ff = f + kw;
            if (tt < Tin && ff < Fin) v = to_f32(x[(((int64_t)b * Tin + tt) * Fin + ff) * C + c]);
        }
        col[i] = from_f32<TO>(v);
    }
}

// bf16 -> bf16 with C % 8 == 0 and ldc % 8 == 0: one thread per 16-byte chunk (8 channels of one tap) - the element-wise kernel above
// spent its time in 64-bit div / mod per 2-byte store (0.6 TB/s on the 184 MB patch matrix of the second conv layer)
__global__ __launch_bounds__(256) void conv_im2col_vec8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int C, int Tin, int Fin,
                                                               int Tout, int Fout, int ldc8, int64_t npos) {
    const int cpt = C >> 3;                                        // chunks per tap
    const int j8 = threadIdx.x % ldc8, prow = threadIdx.x / ldc8, rows_per_block = 256 / ldc8;
    const int tap = j8 / cpt, c8 = (j8 - tap * cpt) * 8, kh = tap / 3, kw = tap - kh * 3;
    const bool live = prow < rows_per_block && tap < 9;
    for (int64_t pos = (int64_t)blockIdx.x * rows_per_block + prow; pos < npos && prow < rows_per_block; pos += (int64_t)gridDim.x * rows_per_block) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (live) {
            const int f = (int)(pos % Fout);
            const int64_t bt = pos / Fout;
            const int t = (int)(bt % Tout), b = (int)(bt / Tout);
            const int tt = 2 * t + kh, ff = f + kw;
            if (tt < Tin && ff < Fin) v = *reinterpret_cast<const u32x4*>(x + (((int64_t)b * Tin + tt) * Fin + ff) * C + c8);
        }
        *reinterpret_cast<u32x4*>(col + pos * (int64_t)ldc8 * 8 + j8 * 8) = v;
    }
}

// col2im (gather form) + ReLU mask: dx[b,ti,fi,c] = (y[b,ti,fi,c] > 0) * sum_{kh,kw} dcol[(b,(ti-kh)/2,fi-kw), tap*32 + c]
// over taps with (ti-kh) even, 0 <= (ti-kh)/2 < Tout, 0 <= fi-kw < Fout.  One thread per (position, 4 channels).
template <typename T>
__global__ __launch_bounds__(256) void conv_col2im_kernel(const T* __restrict__ dcol, int ldc, const T* __restrict__ y,
                                                          T* __restrict__ dx, int Tin, int Fin, int Tout, int Fout, int64_t total) {
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i & 7) * 4;
    const int64_t pos = i >> 3;
    const int fi = (int)(pos % Fin), ti = (int)((pos / Fin) % Tin), b = (int)(pos / ((int64_t)Fin * Tin));
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int t2 = ti - kh;
        if (t2 < 0 || (t2 & 1)) continue;
        const int t = t2 >> 1;
        if (t >= Tout) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int f = fi - kw;
            if (f < 0 || f >= Fout) continue;
            const vec4 v = *reinterpret_cast<const vec4*>(dcol + (((int64_t)b * Tout + t) * Fout + f) * ldc + (kh * 3 + kw) * 32 + c4);
            acc += f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        }
    }
    const vec4 yv = *reinterpret_cast<const vec4*>(y + pos * 32 + c4);
    vec4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (T)(((float)yv[e] > 0.f) ? acc[e] : 0.f);
    *reinterpret_cast<vec4*>(dx + pos * 32 + c4) = o;
}


// ---- direct backward of a 32 -> 32 channel layer (bf16 path): no patch matrix --------------------------------------------------
// The patch-matrix formulation above (im2col, two GEMMs, col2im) moves a [B*Tout*Fout, 288] matrix through HBM four times (184 MB
// each way at S2) for 5.9 GFLOP.  These two kernels read dy and x once.
typedef short s16x4 __attribute__((ext_vector_type(4)));
constexpr int CONV_WS_SLAB = 9216 + 32;      // one workgroup's partial dw [32 x 288] + db [32]
constexpr int CONV_WS_WGS = 256;

// (a) data gradient: the transposed convolution, as an implicit GEMM like the forward kernel.
//   dx[b,ti,fi,ci] = (xin[b,ti,fi,ci] > 0) * sum_{kh,kw,co} dy[b,(ti-kh)/2,fi-kw,co] * w[co][ci][kh][kw]   ((ti-kh) even, in range)
// A tile = 16 consecutive fi of one (b, ti): the row's parity picks the taps (kh = 0, 2 or kh = 1) for the whole tile.
__global__ __launch_bounds__(256) void conv_sub1_bwd_x_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ w,
                                                              const bf16_t* __restrict__ xin, bf16_t* __restrict__ dx, int B, int Tin, int Fin,
                                                              int Tout, int Fout, int fblocks, int n_tiles) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, q4 = lane >> 4;
    const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    // weights as the MFMA row operand: rows = c_in, reduction = c_out: wt[tap][cit] = w[co = q4*8 .. +8][ci = cit*16 + r16][tap]
    u32x4 wt[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cit = 0; cit < 2; ++cit) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16_t)w[((q4 * 8 + j) * 32 + cit * 16 + r16) * 9 + tap];
            wt[tap][cit] = __builtin_bit_cast(u32x4, v);
        }
    for (int tile = gwave; tile < n_tiles; tile += nwaves) {
        const int fb = tile % fblocks, ti = (tile / fblocks) % Tin, b = tile / (fblocks * Tin);
        const int fi = fb * 16 + r16;
        const bool lane_ok = fi < Fin;
        f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t2 = ti - kh;
            if (t2 < 0 || (t2 & 1) || (t2 >> 1) >= Tout) continue;          // (tile-uniform)
            const int t = t2 >> 1;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int f = fi - kw;
                u32x4 g = {0u, 0u, 0u, 0u};
                if (lane_ok && f >= 0 && f < Fout) g = *reinterpret_cast<const u32x4*>(dy + (((int64_t)b * Tout + t) * Fout + f) * 32 + q4 * 8);
                Mma<bf16_t>::run(wt[kh * 3 + kw][0], g, acc[0]);
                Mma<bf16_t>::run(wt[kh * 3 + kw][1], g, acc[1]);
            }
        }
        if (!lane_ok) continue;
        const int64_t pos = ((int64_t)b * Tin + ti) * Fin + fi;
#pragma unroll
        for (int cit = 0; cit < 2; ++cit) {
            const int c0 = cit * 16 + q4 * 4;
            const bf16x4 yv = *reinterpret_cast<const bf16x4*>(xin + pos * 32 + c0);
            bf16x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(((float)yv[i] > 0.f) ? acc[cit][i] : 0.f);
            *reinterpret_cast<bf16x4*>(dx + pos * 32 + c0) = o;
        }
    }
}

// (b) weight and bias gradients: dw[co][tap*32 + ci] += sum_{b,t,f} dy[b,t,f,co] * x[b,2t+kh,f+kw,ci], db[co] += sum dy.
// The reduction runs over positions, the slow index of both operands: a chunk of TT output rows of dy and the 2 TT + 1 input rows
// under it are staged in LDS as they lie in memory (zero-padded to 48 / 50 columns), and the MFMA 16x16x16 fragments - 4 consecutive
// positions of one channel per lane - come from ds_read_b64_tr_b16: a 16-lane group reads a block of 4 positions x 16 channels and
// every lane receives one channel's 4 positions.  A position is 64 bytes, so positions f and f + 4 would share banks: the 16-byte
// chunks of a position are stored at chunk ^ 2 when (f >> 2) is odd (conflict-free for any 8 consecutive positions).  The four waves split the nine taps (3 / 2 / 2 / 2), keep
// their [32 x 32] accumulators across the chunks a persistent workgroup walks, and add them once at the end.
template <int TT>
__global__ __launch_bounds__(256) void conv_sub1_bwd_w_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* __restrict__ ws,
                                                              int B, int Tin, int Fin, int Tout, int Fout, int chunks_per_b, int n_chunks) {
    constexpr int FP = 48, XP = 50;
    __shared__ __attribute__((aligned(16))) unsigned short dys[TT * FP * 32];
    __shared__ __attribute__((aligned(16))) unsigned short xs[(2 * TT + 1) * XP * 32];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntap = wave == 0 ? 3 : 2;
    f32x4 acc[3][2][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int d = 0; d < 2; ++d) acc[a][c][d] = f32x4{0, 0, 0, 0};
    float bsum = 0.f;                                         // thread (co = tid & 31, part = tid >> 5): column sums of dy
    // staging: every thread's 16-byte pieces of a chunk are loaded into registers in ONE batch (3 of dy, 8 of x), and the NEXT
    // chunk's batch is issued before the current chunk is multiplied - a load / store loop per piece serialised ~10 global-load
    // latencies per chunk and was 90 % of this kernel's time
    constexpr int ND = (TT * FP * 4 + 255) / 256, NX = ((2 * TT + 1) * XP * 4 + 255) / 256;
    u32x4 gd[ND], gx[NX];
    auto gload = [&](int chunk) {
        const int b = chunk / chunks_per_b, t0 = (chunk - b * chunks_per_b) * TT, rows = min(TT, Tout - t0);
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, f = (idx >> 2) % FP, r = idx / (4 * FP);
            gd[k] = u32x4{0u, 0u, 0u, 0u};
            if (r < rows && f < Fout) gd[k] = *reinterpret_cast<const u32x4*>(dy + (((int64_t)b * Tout + t0 + r) * Fout + f) * 32 + c * 8);
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, f = (idx >> 2) % XP, r = idx / (4 * XP);
            const int tin = 2 * t0 + r;
            gx[k] = u32x4{0u, 0u, 0u, 0u};
            if (r <= 2 * rows && r < 2 * TT + 1 && tin < Tin && f < Fin)
                gx[k] = *reinterpret_cast<const u32x4*>(x + (((int64_t)b * Tin + tin) * Fin + f) * 32 + c * 8);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, f = (idx >> 2) % FP, r = idx / (4 * FP);
            if (idx < TT * FP * 4) *reinterpret_cast<u32x4*>(dys + (r * FP + f) * 32 + ((c ^ (((f >> 2) & 1) << 1)) * 8)) = gd[k];
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, f = (idx >> 2) % XP, r = idx / (4 * XP);
            if (idx < (2 * TT + 1) * XP * 4) *reinterpret_cast<u32x4*>(xs + (r * XP + f) * 32 + ((c ^ (((f >> 2) & 1) << 1)) * 8)) = gx[k];
        }
    };
    if ((int)blockIdx.x < n_chunks) gload(blockIdx.x);
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int b = chunk / chunks_per_b, t0 = (chunk - b * chunks_per_b) * TT, rows = min(TT, Tout - t0);
        __syncthreads();                                      // the previous chunk's fragments have been read
        lstore();
        __syncthreads();
        if (chunk + (int)gridDim.x < n_chunks) gload(chunk + gridDim.x);
        for (int p = tid >> 5; p < rows * FP; p += 8) {
            const int co = tid & 31, f = p % FP;
            bsum += (float)__builtin_bit_cast(bf16_t, dys[p * 32 + ((((co >> 3) ^ (((f >> 2) & 1) << 1)) << 3) | (co & 7))]);
        }
        for (int r = 0; r < rows; ++r)
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) {
                const int fk = kc * 16 + q4 * 4;                               // this lane's 4 consecutive positions of the K = 16 slice
                const int tq = r16 >> 2, tp = r16 & 3;                        // transposed read: this lane addresses row fk + tq, channels 4 tp .. 4 tp + 3
                auto tr = [&](const unsigned short* base, int f, int ct) {    // (base: the LDS row of positions; f: position; ct: channel tile)
                    const int chunk = (ct * 2 + (tp >> 1)) ^ (((f >> 2) & 1) << 1);
                    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + f * 32 + chunk * 8 + (tp & 1) * 4));
                };
                s16x4 a[2];
#pragma unroll
                for (int cot = 0; cot < 2; ++cot) a[cot] = tr(dys + r * FP * 32, fk + tq, cot);
#pragma unroll
                for (int ta = 0; ta < 3; ++ta) {
                    if (ta >= ntap) break;
                    const int tap = wave + 4 * ta, kh = tap / 3, kw = tap - kh * 3;
                    s16x4 bx[2];
#pragma unroll
                    for (int cit = 0; cit < 2; ++cit) bx[cit] = tr(xs + (2 * r + kh) * XP * 32, fk + tq + kw, cit);
#pragma unroll
                    for (int cot = 0; cot < 2; ++cot)
#pragma unroll
                        for (int cit = 0; cit < 2; ++cit)
                            acc[ta][cot][cit] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[cot], bx[cit], acc[ta][cot][cit], 0, 0, 0);
                }
            }
    }
    // this workgroup's partial sums -> its slab of the workspace (plain stores; conv_sub1_bwd_w_reduce_kernel adds the slabs up:
    // 256 workgroups adding 9 216 floats each into the same 36 KB with float atomics serialised on 576 cache lines)
    float* slab = ws + (int64_t)blockIdx.x * CONV_WS_SLAB;
#pragma unroll
    for (int ta = 0; ta < 3; ++ta) {
        if (ta >= ntap) break;
        const int tap = wave + 4 * ta;
#pragma unroll
        for (int cot = 0; cot < 2; ++cot)
#pragma unroll
            for (int cit = 0; cit < 2; ++cit)
#pragma unroll
                for (int i = 0; i < 4; ++i) slab[(cot * 16 + q4 * 4 + i) * 288 + tap * 32 + cit * 16 + r16] = acc[ta][cot][cit][i];
    }
    __shared__ float bred[8][32];
    __syncthreads();
    bred[tid >> 5][tid & 31] = bsum;
    __syncthreads();
    if (tid < 32) {
        float t = 0.f;
#pragma unroll
        for (int p = 0; p < 8; ++p) t += bred[p][tid];
        slab[9216 + tid] = t;
    }
}

// dw[e] += sum over slabs, e < 9216; db[e - 9216] += ... for the 32 bias sums (db optional)
__global__ __launch_bounds__(256) void conv_sub1_bwd_w_reduce_kernel(const float* __restrict__ ws, int nslabs, float* __restrict__ dw,
                                                                     float* __restrict__ db, int slab = CONV_WS_SLAB, int nw = 9216) {
    // blockIdx.y: one of 8 groups of slabs (8 independent loads in flight per thread; one thread summing all 256 slabs was a chain
    // of 256 load latencies = 61 us); the 8 partial sums of an element meet in a float atomic
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= slab) return;
    const int w0 = (int)((int64_t)blockIdx.y * nslabs / gridDim.y), w1 = (int)((int64_t)(blockIdx.y + 1) * nslabs / gridDim.y);
    float t[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int w = w0;
    for (; w + 8 <= w1; w += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] += ws[(int64_t)(w + u) * slab + e];
    }
    for (; w < w1; ++w) t[0] += ws[(int64_t)w * slab + e];
    const float tot = ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    if (e < nw) atomicAdd(dw + e, tot);
    else if (db) atomicAdd(db + (e - nw), tot);
}


// (c) the first layer (1 -> 32 channels): dw0[co][tap] += sum_pos dy[pos][co] * feats[b, 2t+kh, f+kw], db0[co] += sum dy.  Also an
// MFMA 16x16x16 reduction over positions: A = dy^T as in (b); B[k = position][n] = the patch value of tap n (n < 9, bf16 like the
// patch-matrix route's operands), 1 for n = 9 (that column is the bias gradient), 0 beyond.  The four waves split a chunk's rows.
constexpr int CONV0_SLAB = 288 + 32;
template <int TT>
__global__ __launch_bounds__(256) void conv_sub0_bwd_w_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ feats, float* __restrict__ ws,
                                                              int B, int T, int D, int T1, int F1, int chunks_per_b, int n_chunks) {
    constexpr int FP = 48, XP = 52;
    __shared__ __attribute__((aligned(16))) unsigned short dys[TT * FP * 32];
    __shared__ float xs[(2 * TT + 1) * XP];
    __shared__ float red[4][32][10];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = r16 / 3, kw = r16 - kh * 3;                // this lane's output column = tap r16 (r16 < 9)
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    constexpr int ND = (TT * FP * 4 + 255) / 256, NX = ((2 * TT + 1) * XP + 255) / 256;
    u32x4 gd[ND];
    float gx[NX];
    auto gload = [&](int chunk) {
        const int b = chunk / chunks_per_b, t0 = (chunk - b * chunks_per_b) * TT, rows = min(TT, T1 - t0);
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, f = (idx >> 2) % FP, r = idx / (4 * FP);
            gd[k] = u32x4{0u, 0u, 0u, 0u};
            if (r < rows && f < F1) gd[k] = *reinterpret_cast<const u32x4*>(dy + (((int64_t)b * T1 + t0 + r) * F1 + f) * 32 + c * 8);
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + 256 * k, f = idx % XP, r = idx / XP, tin = 2 * t0 + r;
            gx[k] = (r <= 2 * rows && r < 2 * TT + 1 && tin < T && f < D) ? feats[((int64_t)b * T + tin) * D + f] : 0.f;
        }
    };
    if ((int)blockIdx.x < n_chunks) gload(blockIdx.x);
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int b = chunk / chunks_per_b, t0 = (chunk - b * chunks_per_b) * TT, rows = min(TT, T1 - t0);
        (void)b;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, f = (idx >> 2) % FP, r = idx / (4 * FP);
            if (idx < TT * FP * 4) *reinterpret_cast<u32x4*>(dys + (r * FP + f) * 32 + ((c ^ (((f >> 2) & 1) << 1)) * 8)) = gd[k];
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + 256 * k;
            if (idx < (2 * TT + 1) * XP) xs[idx] = gx[k];
        }
        __syncthreads();
        if (chunk + (int)gridDim.x < n_chunks) gload(chunk + gridDim.x);
        for (int r = wave; r < rows; r += 4)
#pragma unroll
            for (int kc = 0; kc < 3; ++kc) {
                const int fk = kc * 16 + q4 * 4, tq = r16 >> 2, tp = r16 & 3;
                s16x4 a[2];
#pragma unroll
                for (int cot = 0; cot < 2; ++cot) {
                    const int f = fk + tq, chunkc = (cot * 2 + (tp >> 1)) ^ (((f >> 2) & 1) << 1);
                    a[cot] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) s16x4*)(dys + (r * FP + f) * 32 + chunkc * 8 + (tp & 1) * 4));
                }
                bf16x4 pv = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
                if (r16 < 9) {
                    const float* xr = xs + (2 * r + kh) * XP + fk + kw;
                    pv = bf16x4{(bf16_t)xr[0], (bf16_t)xr[1], (bf16_t)xr[2], (bf16_t)xr[3]};
                } else if (r16 == 9) {
                    pv = bf16x4{(bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f, (bf16_t)1.f};
                }
                const s16x4 bx = __builtin_bit_cast(s16x4, pv);
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[0], bx, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a[1], bx, acc[1], 0, 0, 0);
            }
    }
    // D[co = cot*16 + q4*4 + i][n = r16]: the four waves' partials meet in LDS, one slab per workgroup
    if (r16 < 10) {
#pragma unroll
        for (int cot = 0; cot < 2; ++cot)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][cot * 16 + q4 * 4 + i][r16] = acc[cot][i];
    }
    __syncthreads();
    float* slab = ws + (int64_t)blockIdx.x * CONV0_SLAB;
    for (int e = tid; e < 320; e += 256) {
        const int co = e / 10, n = e - co * 10;
        const float v = (red[0][co][n] + red[1][co][n]) + (red[2][co][n] + red[3][co][n]);
        if (n < 9) slab[co * 9 + n] = v;
        else slab[288 + co] = v;
    }
}


// ---- LDS-staged forward and data-gradient kernels of a 32 -> 32 channel layer (bf16) --------------------------------------------
// A workgroup stages the input rows of TT output rows once (28.8 KB: every element feeds up to 9 taps x 2 channel tiles), in the
// position-major layout of (b) - 64 bytes per position, 16-byte chunks swizzled by ((f >> 2) & 1) << 1, which is conflict-free for
// ds_read_b128 at any column offset - and its four waves take the 16-position tiles; each tap's B operand is then one LDS read
// instead of a dependent global load.
template <int TT>
__global__ __launch_bounds__(256) void conv_sub1_lds_kernel(const bf16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            bf16_t* __restrict__ y, int B, int Tin, int Fin, int Tout, int Fout, int last,
                                                            int chunks_per_b, int n_chunks) {
    constexpr int XP = 50;
    __shared__ __attribute__((aligned(16))) unsigned short xs[(2 * TT + 1) * XP * 32];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32x4 wf[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16_t)w[((ct * 16 + r16) * 32 + q4 * 8 + j) * 9 + tap];
            wf[tap][ct] = __builtin_bit_cast(u32x4, v);
        }
    f32x4 bv[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(bias + ct * 16 + q4 * 4);
    constexpr int NX = ((2 * TT + 1) * XP * 4 + 255) / 256;
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int b = chunk / chunks_per_b, t0 = (chunk - b * chunks_per_b) * TT, rows = min(TT, Tout - t0);
        u32x4 gx[NX];
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, f = (idx >> 2) % XP, r = idx / (4 * XP);
            const int tin = 2 * t0 + r;
            gx[k] = u32x4{0u, 0u, 0u, 0u};
            if (r <= 2 * rows && r < 2 * TT + 1 && tin < Tin && f < Fin)
                gx[k] = *reinterpret_cast<const u32x4*>(x + (((int64_t)b * Tin + tin) * Fin + f) * 32 + c * 8);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, f = (idx >> 2) % XP, r = idx / (4 * XP);
            if (idx < (2 * TT + 1) * XP * 4) *reinterpret_cast<u32x4*>(xs + (r * XP + f) * 32 + ((c ^ (((f >> 2) & 1) << 1)) * 8)) = gx[k];
        }
        __syncthreads();
        for (int tile = wave; tile < rows * 3; tile += 4) {
            const int r = tile / 3, f0 = (tile - r * 3) * 16, f = f0 + r16;
            f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int fp = min(f + kw, XP - 1);
                    const u32x4 xv = *reinterpret_cast<const u32x4*>(xs + ((2 * r + kh) * XP + fp) * 32 + ((q4 ^ (((fp >> 2) & 1) << 1)) * 8));
                    Mma<bf16_t>::run(wf[kh * 3 + kw][0], xv, acc[0]);
                    Mma<bf16_t>::run(wf[kh * 3 + kw][1], xv, acc[1]);
                }
            if (f >= Fout) continue;
            const int t = t0 + r;
            const int64_t pos = ((int64_t)b * Tout + t) * Fout + f;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x4 o = acc[ct] + bv[ct];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], 0.f);
                const int c0 = ct * 16 + q4 * 4;
                if (last) {
                    bf16_t* yp = y + ((int64_t)b * Tout + t) * (32 * Fout) + f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) yp[(int64_t)(c0 + i) * Fout] = (bf16_t)o[i];
                } else {
                    const bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                    *reinterpret_cast<bf16x4*>(y + pos * 32 + c0) = ob;
                }
            }
        }
    }
}

// data gradient: a chunk = 2 TT input rows [2 TT c, 2 TT c + 2 TT) of one utterance; they draw on the TT + 1 output rows
// t = TT c - 1 .. TT c + TT - 1 of dy, staged with two zero columns on the left (f - kw runs from -2).
template <int TT>
__global__ __launch_bounds__(256) void conv_sub1_bwd_x_lds_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ w,
                                                                  const bf16_t* __restrict__ xin, bf16_t* __restrict__ dx, int B, int Tin, int Fin,
                                                                  int Tout, int Fout, int chunks_per_b, int n_chunks) {
    constexpr int FP = 52;
    __shared__ __attribute__((aligned(16))) unsigned short dys[(TT + 1) * FP * 32];
    const int tid = threadIdx.x, lane = tid & 63, r16 = lane & 15, q4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u32x4 wt[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cit = 0; cit < 2; ++cit) {
            bf16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (bf16_t)w[((q4 * 8 + j) * 32 + cit * 16 + r16) * 9 + tap];
            wt[tap][cit] = __builtin_bit_cast(u32x4, v);
        }
    constexpr int ND = ((TT + 1) * FP * 4 + 255) / 256;
    const int fblocks = (Fin + 15) / 16;
    for (int chunk = blockIdx.x; chunk < n_chunks; chunk += gridDim.x) {
        const int b = chunk / chunks_per_b, cb = chunk - b * chunks_per_b, ti0 = 2 * TT * cb, tb = TT * cb - 1;
        u32x4 gd[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, fc = (idx >> 2) % FP, r = idx / (4 * FP);
            const int t = tb + r, f = fc - 2;
            gd[k] = u32x4{0u, 0u, 0u, 0u};
            if (r <= TT && t >= 0 && t < Tout && f >= 0 && f < Fout)
                gd[k] = *reinterpret_cast<const u32x4*>(dy + (((int64_t)b * Tout + t) * Fout + f) * 32 + c * 8);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const int idx = tid + 256 * k, c = idx & 3, fc = (idx >> 2) % FP, r = idx / (4 * FP);
            if (idx < (TT + 1) * FP * 4) *reinterpret_cast<u32x4*>(dys + (r * FP + fc) * 32 + ((c ^ (((fc >> 2) & 1) << 1)) * 8)) = gd[k];
        }
        __syncthreads();
        const int nrows = min(2 * TT, Tin - ti0);
        for (int tile = wave; tile < nrows * fblocks; tile += 4) {
            const int ri = tile / fblocks, f0 = (tile - ri * fblocks) * 16, ti = ti0 + ri, fi = f0 + r16;
            f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int t2 = ti - kh;
                if (t2 & 1) continue;                                   // (tile-uniform; rows outside [0, Tout) are zero in the staged image)
                const int rr = (t2 >> 1) - tb;                          // 0 .. TT
                if (rr < 0 || rr > TT) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int fc = min(fi - kw + 2, FP - 1);
                    const u32x4 g = *reinterpret_cast<const u32x4*>(dys + (rr * FP + fc) * 32 + ((q4 ^ (((fc >> 2) & 1) << 1)) * 8));
                    Mma<bf16_t>::run(wt[kh * 3 + kw][0], g, acc[0]);
                    Mma<bf16_t>::run(wt[kh * 3 + kw][1], g, acc[1]);
                }
            }
            if (fi >= Fin) continue;
            const int64_t pos = ((int64_t)b * Tin + ti) * Fin + fi;
#pragma unroll
            for (int cit = 0; cit < 2; ++cit) {
                const int c0 = cit * 16 + q4 * 4;
                const bf16x4 yv = *reinterpret_cast<const bf16x4*>(xin + pos * 32 + c0);
                bf16x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16_t)(((float)yv[i] > 0.f) ? acc[cit][i] : 0.f);
                *reinterpret_cast<bf16x4*>(dx + pos * 32 + c0) = o;
            }
        }
    }
}

}  // namespace

extern "C" int asr_conv_im2col(void* stream, const void* x, int x_dtype, int C, void* col, int col_dtype, int ldc, int B, int Tin, int Fin,
                               int Tout, int Fout) {
    ASR_REQUIRE(x && col && B > 0 && C > 0 && ldc >= 9 * C && Tout > 0 && Fout > 0, ASR_ERR_ARG, "im2col: bad args");
    const int64_t total = (int64_t)B * Tout * Fout * ldc;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 g((unsigned)blocks), b(256);
    if (x_dtype == ASR_F32 && col_dtype == ASR_F32)
        hipLaunchKernelGGL((conv_im2col_kernel<float, float>), g, b, 0, s, (const float*)x, (float*)col, C, Tin, Fin, Tout, Fout, ldc, total);
    else if (x_dtype == ASR_F32 && col_dtype == ASR_BF16)
        hipLaunchKernelGGL((conv_im2col_kernel<float, bf16_t>), g, b, 0, s, (const float*)x, (bf16_t*)col, C, Tin, Fin, Tout, Fout, ldc, total);
    else if (x_dtype == ASR_BF16 && col_dtype == ASR_BF16 && C % 8 == 0 && ldc % 8 == 0 && ldc / 8 <= 256 && asr_aligned(x, 16) && asr_aligned(col, 16)) {
        const int ldc8 = ldc / 8, rpb = 256 / ldc8;
        const int64_t npos = (int64_t)B * Tout * Fout;
        int64_t nb = (npos + rpb - 1) / rpb;
        if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL(conv_im2col_vec8_kernel, dim3((unsigned)nb), b, 0, s, (const bf16_t*)x, (bf16_t*)col, C, Tin, Fin, Tout, Fout, ldc8, npos);
    } else if (x_dtype == ASR_BF16 && col_dtype == ASR_BF16)
        hipLaunchKernelGGL((conv_im2col_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)x, (bf16_t*)col, C, Tin, Fin, Tout, Fout, ldc, total);
    else
        ASR_REQUIRE(false, ASR_ERR_UNSUPPORTED, "im2col: dtype combination");
    ASR_LAUNCH_CHECK("conv_im2col");
    return 0;
}

extern "C" int asr_conv_col2im_relu(void* stream, const void* dcol, int ldc, const void* y, void* dx, int B, int Tin, int Fin, int Tout,
                                    int Fout) {
    ASR_REQUIRE(dcol && y && dx && B > 0 && ldc >= 288 && ldc % 4 == 0, ASR_ERR_ARG, "col2im: bad args");
    const int64_t total = (int64_t)B * Tin * Fin * 8;
    hipLaunchKernelGGL(conv_col2im_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const bf16_t*)dcol, ldc, (const bf16_t*)y, (bf16_t*)dx, Tin, Fin, Tout, Fout, total);
    ASR_LAUNCH_CHECK("conv_col2im");
    return 0;
}

extern "C" int asr_conv_col2im_relu_f32(void* stream, const float* dcol, int ldc, const float* y, float* dx, int B, int Tin, int Fin, int Tout,
                                        int Fout) {
    ASR_REQUIRE(dcol && y && dx && B > 0 && ldc >= 288 && ldc % 4 == 0, ASR_ERR_ARG, "col2im_f32: bad args");
    const int64_t total = (int64_t)B * Tin * Fin * 8;
    hipLaunchKernelGGL(conv_col2im_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dcol,
                       ldc, y, dx, Tin, Fin, Tout, Fout, total);
    ASR_LAUNCH_CHECK("conv_col2im_f32");
    return 0;
}

extern "C" int asr_conv_sub0_fwd(void* stream, const float* feats, const float* w0, const float* b0, void* y, int dtype, int B, int T,
                                 int D, int T1, int F1) {
    ASR_REQUIRE(feats && w0 && b0 && y, ASR_ERR_ARG, "conv_sub0: null pointer");
    ASR_REQUIRE(B > 0 && T > 0 && D > 0 && T1 > 0 && F1 > 0, ASR_ERR_ARG, "conv_sub0: bad sizes");
    ASR_REQUIRE(asr_aligned(y, 16), ASR_ERR_ALIGN, "conv_sub0: y alignment");
    const int64_t total = (int64_t)B * T1 * F1;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ASR_F32)
        hipLaunchKernelGGL(conv_sub0_kernel<float>, grid, block, 0, s, feats, w0, b0, (float*)y, B, T, D, T1, F1);
    else
        hipLaunchKernelGGL(conv_sub0_kernel<bf16_t>, grid, block, 0, s, feats, w0, b0, (bf16_t*)y, B, T, D, T1, F1);
    ASR_LAUNCH_CHECK("conv_sub0");
    return 0;
}

extern "C" int asr_conv_sub1_fwd(void* stream, const void* x, const float* w, const float* b, void* y, int dtype, int B, int Tin,
                                 int Fin, int Tout, int Fout, int last) {
    ASR_REQUIRE(x && w && b && y, ASR_ERR_ARG, "conv_sub1: null pointer");
    ASR_REQUIRE(B > 0 && Tout > 0 && Fout > 0 && Tin >= 2 * Tout + 1 && Fin >= Fout + 2, ASR_ERR_ARG,
                "conv_sub1: input region [%d,%d] too small for output [%d,%d]", Tin, Fin, Tout, Fout);
    ASR_REQUIRE(asr_aligned(x, 16) && asr_aligned(y, 16) && asr_aligned(b, 16), ASR_ERR_ALIGN, "conv_sub1: alignment");
    const int64_t total = (int64_t)B * Tout * Fout;
    const int n_tiles = (int)((total + 15) / 16);
    int blocks = (n_tiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ASR_BF16 && Fout <= 48 && Fin <= 50) {
        constexpr int TT = 4;
        const int chunks_per_b = (Tout + TT - 1) / TT, n_chunks = B * chunks_per_b;
        hipLaunchKernelGGL(conv_sub1_lds_kernel<TT>, dim3(n_chunks < 1280 ? n_chunks : 1280), dim3(256), 0, s, (const bf16_t*)x, w, b, (bf16_t*)y,
                           B, Tin, Fin, Tout, Fout, last, chunks_per_b, n_chunks);
        ASR_LAUNCH_CHECK("conv_sub1_lds");
        return 0;
    }
    if (dtype == ASR_F32)
        hipLaunchKernelGGL(conv_sub1_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, w, b, (float*)y, B, Tin, Fin,
                           Tout, Fout, last, n_tiles);
    else
        hipLaunchKernelGGL(conv_sub1_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, w, b, (bf16_t*)y, B, Tin, Fin,
                           Tout, Fout, last, n_tiles);
    ASR_LAUNCH_CHECK("conv_sub1");
    return 0;
}

extern "C" int asr_conv_sub1_bwd_x(void* stream, const void* dy, const float* w, const void* xin, void* dx, int B, int Tin, int Fin, int Tout,
                                   int Fout) {
    ASR_REQUIRE(dy && w && xin && dx && B > 0 && Tin > 0 && Fin > 0 && Tout > 0 && Fout > 0, ASR_ERR_ARG, "conv_sub1_bwd_x: bad args");
    ASR_REQUIRE(asr_aligned(dy, 16) && asr_aligned(xin, 8) && asr_aligned(dx, 8), ASR_ERR_ALIGN, "conv_sub1_bwd_x: alignment");
    const int fblocks = (Fin + 15) / 16;
    const int64_t tiles = (int64_t)B * Tin * fblocks;
    ASR_REQUIRE(tiles < (int64_t)1 << 31, ASR_ERR_UNSUPPORTED, "conv_sub1_bwd_x: too many tiles");
    if (Fout <= 48 && Fin <= 50 && Tin <= 2 * Tout + 1) {
        constexpr int TT = 4;
        const int chunks_per_b = (Tin + 2 * TT - 1) / (2 * TT), n_chunks = B * chunks_per_b;
        hipLaunchKernelGGL(conv_sub1_bwd_x_lds_kernel<TT>, dim3(n_chunks < 1280 ? n_chunks : 1280), dim3(256), 0, static_cast<hipStream_t>(stream),
                           (const bf16_t*)dy, w, (const bf16_t*)xin, (bf16_t*)dx, B, Tin, Fin, Tout, Fout, chunks_per_b, n_chunks);
        ASR_LAUNCH_CHECK("conv_sub1_bwd_x_lds");
        return 0;
    }
    int blocks = (int)((tiles + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_sub1_bwd_x_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)dy, w,
                       (const bf16_t*)xin, (bf16_t*)dx, B, Tin, Fin, Tout, Fout, fblocks, (int)tiles);
    ASR_LAUNCH_CHECK("conv_sub1_bwd_x");
    return 0;
}

extern "C" int64_t asr_conv_sub1_bwd_w_workspace_floats(void) { return (int64_t)CONV_WS_WGS * CONV_WS_SLAB; }

extern "C" int asr_conv_sub1_bwd_w(void* stream, const void* dy, const void* x, float* dw, float* db, float* workspace, int B, int Tin, int Fin,
                                   int Tout, int Fout) {
    ASR_REQUIRE(dy && x && dw && workspace && B > 0 && Tout > 0, ASR_ERR_ARG, "conv_sub1_bwd_w: bad args");
    ASR_REQUIRE(Fout <= 48 && Fin <= 50 && Fin >= Fout + 2 && Tin >= 2 * Tout + 1, ASR_ERR_UNSUPPORTED,
                "conv_sub1_bwd_w: region [%d,%d] -> [%d,%d] outside the staged tile (Fout <= 48, Fin <= 50)", Tin, Fin, Tout, Fout);
    ASR_REQUIRE(asr_aligned(dy, 16) && asr_aligned(x, 16), ASR_ERR_ALIGN, "conv_sub1_bwd_w: alignment");
    constexpr int TT = 4;
    const int chunks_per_b = (Tout + TT - 1) / TT, n_chunks = B * chunks_per_b;
    const int blocks = n_chunks < CONV_WS_WGS ? n_chunks : CONV_WS_WGS;
    hipLaunchKernelGGL(conv_sub1_bwd_w_kernel<TT>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)dy,
                       (const bf16_t*)x, workspace, B, Tin, Fin, Tout, Fout, chunks_per_b, n_chunks);
    ASR_LAUNCH_CHECK("conv_sub1_bwd_w");
    hipLaunchKernelGGL(conv_sub1_bwd_w_reduce_kernel, dim3((CONV_WS_SLAB + 255) / 256, 8), dim3(256), 0, static_cast<hipStream_t>(stream), workspace,
                       blocks, dw, db);
    ASR_LAUNCH_CHECK("conv_sub1_bwd_w_reduce");
    return 0;
}

extern "C" int asr_conv_sub0_bwd_w(void* stream, const void* dy, const float* feats, float* dw, float* db, float* workspace, int B, int T, int D,
                                   int T1, int F1) {
    ASR_REQUIRE(dy && feats && dw && workspace && B > 0 && T > 0 && D > 0 && T1 > 0 && F1 > 0, ASR_ERR_ARG, "conv_sub0_bwd_w: bad args");
    ASR_REQUIRE(F1 <= 48 && F1 + 2 <= 52, ASR_ERR_UNSUPPORTED, "conv_sub0_bwd_w: F1 = %d outside the staged tile (<= 48)", F1);
    ASR_REQUIRE(asr_aligned(dy, 16), ASR_ERR_ALIGN, "conv_sub0_bwd_w: alignment");
    constexpr int TT = 8;
    const int chunks_per_b = (T1 + TT - 1) / TT, n_chunks = B * chunks_per_b;
    const int blocks = n_chunks < CONV_WS_WGS ? n_chunks : CONV_WS_WGS;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(conv_sub0_bwd_w_kernel<TT>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)dy, feats, workspace, B, T, D, T1, F1,
                       chunks_per_b, n_chunks);
    ASR_LAUNCH_CHECK("conv_sub0_bwd_w");
    hipLaunchKernelGGL(conv_sub1_bwd_w_reduce_kernel, dim3((CONV0_SLAB + 255) / 256, 8), dim3(256), 0, s, workspace, blocks, dw, db, CONV0_SLAB, 288);
    ASR_LAUNCH_CHECK("conv_sub0_bwd_w_reduce");
    return 0;
}
