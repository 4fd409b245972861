// Greedy decoding helpers: what sits between the logits and the token ids in
//   Decoder.step / batch_decode   src/transformer/decoder.py:98-164   F.log_softmax, torch.argmax over the vocabulary
//   GreedyDecoder.decode          src/ctcModel/ctc_infer.py:28-46,69-80   torch.max over the vocabulary, collapse repeats, drop blanks
// All HBM-bound row kernels: one wavefront per row / per utterance.
#include "asr_common.h"

namespace {

// index of the row maximum; ties go to the lowest index (torch.argmax / torch.max on CPU)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int64_t ld, int M, int V, int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
        const float v = xr[c];
        if (v > best || bi == 0x7fffffff) { best = v; bi = c; }      // strictly greater: the first maximum of this lane's columns
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64);
        const int i2 = __shfl_xor(bi, o, 64);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

// y = x - logsumexp(x) per row (F.log_softmax, decoder.py:118)
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, int64_t ldx, int M, int V, float* __restrict__ y,
                                                               int64_t ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * ldx;
    float m = -INFINITY, s = 0.f;
    for (int c = lane; c < V; c += 64) lse_combine(m, s, xr[c], 1.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        lse_combine(m, s, m2, s2);
    }
    const float l = m + logf(s);
    for (int c = lane; c < V; c += 64) y[row * ldy + c] = xr[c] - l;
}

// CTC collapse (ctc_infer.py:37-46): walk the first len[b] frame labels, keep a label that is not blank and differs from the
// previous FRAME's label.  One wavefront per utterance, 64 frames per step, compaction by ballot + popcount.
__global__ __launch_bounds__(64) void ctc_greedy_reduce_kernel(const int64_t* __restrict__ frames, const int32_t* __restrict__ len, int L,
                                                               int blank, int64_t* __restrict__ out, int32_t* __restrict__ out_len) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t* f = frames + (int64_t)b * L;
    int64_t* o = out + (int64_t)b * L;
    const int n = min(len[b], L);
    int kept = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        const int64_t tok = t < n ? f[t] : blank;
        const int64_t prev = (t > 0 && t < n) ? f[t - 1] : -1;
        const bool keep = t < n && tok != blank && tok != prev;
        const unsigned long long mask = __ballot(keep);
        if (keep) o[kept + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = tok;
        kept += __builtin_popcountll(mask);
    }
    for (int t = kept + lane; t < L; t += 64) o[t] = 0;      // zero padding (padding_list_seqs pad=0)
    if (lane == 0) out_len[b] = kept;
}


// ---- greedy decoding with the step position in DEVICE memory: the per-token step (embed, 6 layers, projection, argmax, advance)
// has no host-visible argument that changes from token to token, so it is captured once as a hipGraph and replayed.
// state[0] = t (tokens decoded so far = position of the token being fed), state[1] = the step count at which every row had
// produced <eos> (-1 until then; once set, advance changes nothing any more).

// x[b] = emb[cur[b]] + pe[t]   (decoder.py:104-105 for the one new position)
__global__ __launch_bounds__(256) void decode_embed_kernel(const int64_t* __restrict__ cur, const float* __restrict__ emb,
                                                           const float* __restrict__ pe, const int32_t* __restrict__ state,
                                                           float* __restrict__ y32, bf16_t* __restrict__ y16, int B, int D, int V, int max_pos) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B) return;
    const int t = min(max(state[0], 0), max_pos - 1);
    int64_t id = cur[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    for (int c = lane * 4; c < D; c += 256) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(emb + id * D + c) + *reinterpret_cast<const f32x4*>(pe + (int64_t)t * D + c);
        *reinterpret_cast<f32x4*>(y32 + (int64_t)row * D + c) = o;
        if (y16) {
            const bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            *reinterpret_cast<bf16x4*>(y16 + (int64_t)row * D + c) = ob;
        }
    }
}

// the new position's key / value heads [B*h, 64] into slot t of the caches [B*h, Tmax, 64] (elements of `esz` bytes)
__global__ __launch_bounds__(256) void kv_cache_put_kernel(const unsigned char* __restrict__ k_new, const unsigned char* __restrict__ v_new,
                                                           unsigned char* __restrict__ k_cache, unsigned char* __restrict__ v_cache,
                                                           const int32_t* __restrict__ state, int BH, int Tmax, int esz) {
    const int t = state[0];
    if (t < 0 || t >= Tmax) return;
    const int rowb = 64 * esz, words = rowb / 4;                  // bytes / dwords per head row
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;    // one dword each
    if (i >= (int64_t)BH * words) return;
    const int bh = (int)(i / words), w = (int)(i - (int64_t)bh * words);
    const int64_t dst = ((int64_t)bh * Tmax + t) * rowb + w * 4, src = (int64_t)bh * rowb + w * 4;
    *reinterpret_cast<uint32_t*>(k_cache + dst) = *reinterpret_cast<const uint32_t*>(k_new + src);
    *reinterpret_cast<uint32_t*>(v_cache + dst) = *reinterpret_cast<const uint32_t*>(v_new + src);
}

// decoder.py:151-158 for one step: append the argmax tokens, update `finished` and the decoded lengths, move to the next position
__global__ __launch_bounds__(256) void decode_advance_kernel(const int64_t* __restrict__ cur, int64_t* __restrict__ preds, int32_t* __restrict__ state,
                                                             int32_t* __restrict__ k_len, unsigned char* __restrict__ finished,
                                                             int64_t* __restrict__ len_decoded, int eos, int B, int Tp1) {
    const int t = state[0];
    if (state[1] >= 0 || t + 1 >= Tp1) return;                    // (uniform: every thread reads the same words)
    int all = 1;
    for (int b = threadIdx.x; b < B; b += 256) {
        const int64_t c = cur[b];
        preds[(int64_t)b * Tp1 + t + 1] = c;
        const unsigned char f = finished[b] | (c == eos ? 1 : 0);
        finished[b] = f;
        len_decoded[b] += f ? 0 : 1;
        k_len[b] = t + 2;
        all &= f;
    }
    all = __syncthreads_and(all);
    if (threadIdx.x == 0) {
        state[0] = t + 1;
        if (all) state[1] = t + 1;
    }
}


// ---- beam search (decoder.py:166-234) ---------------------------------------------------------------------------------------------
// torch.topk(x, k, sorted=True) over the rows of x [M, V]: k passes of "largest (value, lowest index) strictly after the previous
// pick in (value desc, index asc) order" - one wavefront per row; V is the vocabulary (or beam * beam), k the beam size.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, int64_t ld, int M, int V, int k, float* __restrict__ vals,
                                                        int64_t* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * ld;
    float lastv = INFINITY;
    int lasti = -1;
    for (int j = 0; j < k; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = lane; c < V; c += 64) {
            const float v = xr[c];
            const bool after = (v < lastv) || (v == lastv && c > lasti);          // not picked yet
            if (after && (bi == 0x7fffffff || v > best)) { best = v; bi = c; }     // first maximum of this lane's columns
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > best || (v2 == best && i2 < bi))) { best = v2; bi = i2; }
        }
        if (lane == 0) {
            vals[row * k + j] = bi == 0x7fffffff ? -INFINITY : best;
            idx[row * k + j] = bi == 0x7fffffff ? 0 : bi;
        }
        lastv = best;
        lasti = bi;
    }
}

// top-k of log_softmax(x) per row in one pass over the row (decoder.py:418-440: F.log_softmax then torch.topk): the row lives in
// registers (NPL values per lane, one wavefront per row), the logsumexp is reduced in log_softmax_rows_kernel's order (same values),
// then k rounds of topk_rows_kernel's selection on the registers
template <int NPL, bool TWICE>
__global__ __launch_bounds__(256) void lsm_topk_rows_kernel(const float* __restrict__ x, int64_t ld, int M, int V, int k,
                                                            float* __restrict__ vals, int64_t* __restrict__ idx) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * ld;
    float y[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) y[i] = lane + 64 * i < V ? xr[lane + 64 * i] : -INFINITY;
#pragma unroll
    for (int pass = 0; pass < (TWICE ? 2 : 1); ++pass) {
        float m = -INFINITY, s = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            if (lane + 64 * i < V) lse_combine(m, s, y[i], 1.f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
            lse_combine(m, s, m2, s2);
        }
        const float l = m + logf(s);
#pragma unroll
        for (int i = 0; i < NPL; ++i) y[i] -= l;
    }
    float lastv = INFINITY;
    int lasti = -1;
    for (int j = 0; j < k; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = lane + 64 * i;
            const float v = y[i];
            const bool after = c < V && ((v < lastv) || (v == lastv && c > lasti));
            if (after && (bi == 0x7fffffff || v > best)) { best = v; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > best || (v2 == best && i2 < bi))) { best = v2; bi = i2; }
        }
        if (lane == 0) {
            vals[row * k + j] = bi == 0x7fffffff ? -INFINITY : best;
            idx[row * k + j] = bi == 0x7fffffff ? 0 : bi;
        }
        lastv = best;
        lasti = bi;
    }
}

// The same for long rows (the vocabulary), one WORKGROUP per row: 256 threads hold the row (NPL values each), the logsumexp and the k
// selection rounds reduce over waves through LDS - the wave-per-row kernel above walks 67 values per lane and takes 35 us for 160 rows
// of 4234.  TWICE: log_softmax applied twice before the top-k (Decoder.batch_beam_decode re-normalises Decoder.step's
// log-probabilities, decoder.py:118 + :191).  Values agree with the two-kernel path to f32 rounding (another summation order).
template <int NPL, bool TWICE>
__global__ __launch_bounds__(256) void lsm_topk_rows_wg_kernel(const float* __restrict__ x, int64_t ld, int M, int V, int k,
                                                               float* __restrict__ vals, int64_t* __restrict__ idx) {
    __shared__ float sm[4], ss[4];
    __shared__ float bv[4];
    __shared__ int bi_s[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const float* xr = x + row * ld;
    float y[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) y[i] = tid + 256 * i < V ? xr[tid + 256 * i] : -INFINITY;
#pragma unroll
    for (int pass = 0; pass < (TWICE ? 2 : 1); ++pass) {
        float m = -INFINITY, s = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i)
            if (tid + 256 * i < V) lse_combine(m, s, y[i], 1.f);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
            lse_combine(m, s, m2, s2);
        }
        if (lane == 0) { sm[wave] = m; ss[wave] = s; }
        __syncthreads();
        m = sm[0];
        s = ss[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) lse_combine(m, s, sm[w], ss[w]);
        const float l = m + logf(s);
#pragma unroll
        for (int i = 0; i < NPL; ++i) y[i] -= l;
        __syncthreads();
    }
    float lastv = INFINITY;
    int lasti = -1;
    for (int j = 0; j < k; ++j) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = tid + 256 * i;
            const float v = y[i];
            const bool after = c < V && ((v < lastv) || (v == lastv && c > lasti));
            if (after && (bi == 0x7fffffff || v > best)) { best = v; bi = c; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > best || (v2 == best && i2 < bi))) { best = v2; bi = i2; }
        }
        if (lane == 0) { bv[wave] = best; bi_s[wave] = bi; }
        __syncthreads();
        best = bv[0];
        bi = bi_s[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float v2 = bv[w];
            const int i2 = bi_s[w];
            if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > best || (v2 == best && i2 < bi))) { best = v2; bi = i2; }
        }
        if (tid == 0) {
            vals[row * k + j] = bi == 0x7fffffff ? -INFINITY : best;
            idx[row * k + j] = bi == 0x7fffffff ? 0 : bi;
        }
        lastv = best;
        lasti = bi;
        __syncthreads();
    }
}

// One pruning step (decoder.py:196-209) per utterance: candidates c = parent * beam + j with score scores[parent] + next_scores[parent][j]
// (an f32 add, like the reference's broadcast add), the best `beam` of the beam * beam in sorted order -> their scores, parent ROWS
// (k_indices // beam_size, global row ids) and tokens.  One wavefront per utterance, lane = candidate (beam * beam <= 64).
__global__ __launch_bounds__(64) void beam_prune_kernel(const float* __restrict__ scores, const float* __restrict__ next_scores,
                                                        const int64_t* __restrict__ next_preds, int beam, float* __restrict__ new_scores,
                                                        int64_t* __restrict__ parent, int64_t* __restrict__ new_tok) {
    const int b = blockIdx.x, lane = threadIdx.x, nc = beam * beam;
    const int row = b * beam + lane / beam;
    const float v = lane < nc ? scores[row] + next_scores[(int64_t)b * nc + lane] : -INFINITY;
    const int64_t tok = lane < nc ? next_preds[(int64_t)b * nc + lane] : 0;
    float lastv = INFINITY;
    int lasti = -1;
    for (int j = 0; j < beam; ++j) {
        const bool after = lane < nc && ((v < lastv) || (v == lastv && lane > lasti));
        float best = after ? v : -INFINITY;
        int bi = after ? lane : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > best || (v2 == best && i2 < bi))) { best = v2; bi = i2; }
        }
        if (bi == 0x7fffffff) bi = 0;                               // (fewer than `beam` candidates cannot happen: nc >= beam)
        if (lane == bi) {
            new_scores[b * beam + j] = v;
            parent[b * beam + j] = row;
            new_tok[b * beam + j] = tok;
        }
        lastv = best;
        lasti = bi;
    }
}

// ---- batched beam search over integrated frames (Decoder_CIF.recognize_beam, decoder.py:425-475, for B utterances at once) ----------
// Rows r = b * beam + j are the hypotheses; the step position t lives in state[0]; utterance b is live while t < n_steps[b].

// out[r] = [frames[b, t, :D] | other] (f32), other = other32[r, :D2] or, with `cur`, emb[cur[r]] + pe[t] (D2 = D): the decoder's input
// rows (decoder.py:407-408) and the rows of the output projection (decoder.py:416)
__global__ __launch_bounds__(256) void beam_cat_frames_kernel(const float* __restrict__ frames, const int32_t* __restrict__ state,
                                                              const float* __restrict__ other32, const int64_t* __restrict__ cur,
                                                              const float* __restrict__ emb, const float* __restrict__ pe,
                                                              float* __restrict__ out, int N, int beam, int Tmax, int D, int D2, int V,
                                                              int max_pos) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int t = min(max(state[0], 0), Tmax - 1);
    const float* fr = frames + ((int64_t)(row / beam) * Tmax + t) * D;
    float* o = out + (int64_t)row * (D + D2);
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<f32x4*>(o + c) = *reinterpret_cast<const f32x4*>(fr + c);
    if (cur) {
        int64_t id = cur[row];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        const int tp = min(t, max_pos - 1);
        for (int c = lane * 4; c < D2; c += 256)
            *reinterpret_cast<f32x4*>(o + D + c) =
                *reinterpret_cast<const f32x4*>(emb + id * D2 + c) + *reinterpret_cast<const f32x4*>(pe + (int64_t)tp * D2 + c);
    } else {
        for (int c = lane * 4; c < D2; c += 256)
            *reinterpret_cast<f32x4*>(o + D + c) = *reinterpret_cast<const f32x4*>(other32 + (int64_t)row * D2 + c);
    }
}

// One search step for utterance b = blockIdx.x (decoder.py:445-462): candidates c = j * beam + k (hypothesis-major, rank-minor) with
// score scores[row j] + next_scores[row j][k]; the best `beam` by score, ties in candidate order (Python's stable sort), become the new
// hypotheses IN PLACE: scores, token rows preds[., 0..t] gathered from the parents plus the new token at column t + 1, `cur`, and the
// parent row ids (for the K/V caches).  A finished utterance (t >= n_steps[b]) keeps everything and reports identity parents.
constexpr int BEAM_MAX_W = 512;
__global__ __launch_bounds__(64) void beam_step_kernel(float* __restrict__ scores, const float* __restrict__ next_scores,
                                                       const int64_t* __restrict__ next_preds, int64_t* __restrict__ preds,
                                                       const int32_t* __restrict__ state, const int32_t* __restrict__ n_steps,
                                                       int64_t* __restrict__ parent, int64_t* __restrict__ cur, int beam, int W) {
    __shared__ int sel_row[8];
    __shared__ int64_t sel_tok[8];
    __shared__ float sel_score[8];
    __shared__ int64_t rows[8 * BEAM_MAX_W];
    const int b = blockIdx.x, lane = threadIdx.x, nc = beam * beam;
    const int t = state[0];
    if (t >= n_steps[b] || t + 1 >= W || state[1] >= 0) {      // (state[1]: the step count at which the search stopped, -1 while it runs)
        if (lane < beam) parent[b * beam + lane] = b * beam + lane;
        return;
    }
    const int row = b * beam + lane / beam;
    const float v = lane < nc ? scores[row] + next_scores[(int64_t)b * nc + lane] : -INFINITY;
    const int64_t tok = lane < nc ? next_preds[(int64_t)b * nc + lane] : 0;
    float lastv = INFINITY;
    int lasti = -1;
    for (int j = 0; j < beam; ++j) {
        const bool after = lane < nc && ((v < lastv) || (v == lastv && lane > lasti));
        float best = after ? v : -INFINITY;
        int bi = after ? lane : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(best, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (i2 != 0x7fffffff && (bi == 0x7fffffff || v2 > best || (v2 == best && i2 < bi))) { best = v2; bi = i2; }
        }
        if (bi == 0x7fffffff) bi = 0;
        if (lane == bi) { sel_row[j] = row; sel_tok[j] = tok; sel_score[j] = v; }
        lastv = best;
        lasti = bi;
    }
    __syncthreads();
    for (int j = 0; j < beam; ++j)
        for (int c = lane; c <= t; c += 64) rows[j * BEAM_MAX_W + c] = preds[(int64_t)sel_row[j] * W + c];
    __syncthreads();
    for (int j = 0; j < beam; ++j) {
        int64_t* pr = preds + (int64_t)(b * beam + j) * W;
        for (int c = lane; c <= t; c += 64) pr[c] = rows[j * BEAM_MAX_W + c];
    }
    if (lane < beam) {
        const int r = b * beam + lane;
        preds[(int64_t)r * W + t + 1] = sel_tok[lane];
        scores[r] = sel_score[lane];
        cur[r] = sel_tok[lane];
        parent[r] = sel_row[lane];
    }
}

// K/V caches [n_kv, N, h, Tmax, 64] re-gathered by parent row, in place: a thread owns one 16-byte piece of one cache position for all
// `beam` rows of its utterance (reads every source, then writes), positions <= t only
__global__ __launch_bounds__(256) void beam_reorder_cache_kernel(uint4* __restrict__ cache, const int64_t* __restrict__ parent,
                                                                 const int32_t* __restrict__ state, int n_kv, int B, int beam, int h,
                                                                 int Tmax, int segs) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int seg = (int)(i % segs);
    const int pos = (int)((i / segs) % Tmax);
    const int hd = (int)((i / ((int64_t)segs * Tmax)) % h);
    const int b = (int)((i / ((int64_t)segs * Tmax * h)) % B);
    const int kv = (int)(i / ((int64_t)segs * Tmax * h * B));
    if (kv >= n_kv || pos > state[0]) return;
    bool same = true;
    int src[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        src[r] = r < beam ? (int)(parent[b * beam + r] - (int64_t)b * beam) : 0;
        same = same && (r >= beam || src[r] == r);
    }
    if (same) return;
    const int64_t row_stride = (int64_t)h * Tmax * segs;
    uint4* base = cache + (((int64_t)kv * B * beam + (int64_t)b * beam) * h + hd) * Tmax * segs + (int64_t)pos * segs + seg;
    uint4 v[8];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (r < beam) v[r] = base[(int64_t)min(max(src[r], 0), beam - 1) * row_stride];
#pragma unroll
    for (int r = 0; r < 8; ++r)
        if (r < beam) base[(int64_t)r * row_stride] = v[r];
}

// End of a search step: t += 1, every row's key length += 1; with `finished` given also decoder.py:211-216 - finished[r] |= cur[r] == eos,
// len_decoded[r] += !finished[r] (both stay with the beam SLOT, like the reference's), and once every row is finished state[1] = the
// number of steps taken: from then on asr_beam_step and this kernel change nothing (the host reads the flag every few steps)
__global__ __launch_bounds__(256) void beam_advance_kernel(int32_t* __restrict__ state, int32_t* __restrict__ k_len, int N,
                                                           const int64_t* __restrict__ cur, int eos, unsigned char* __restrict__ finished,
                                                           int64_t* __restrict__ len_decoded) {
    if (state[1] >= 0) return;
    const int t = state[0];
    int all = 1;
    for (int r = threadIdx.x; r < N; r += 256) {
        k_len[r] += 1;
        if (finished) {
            const unsigned char f = finished[r] | (unsigned char)(cur[r] == eos);
            finished[r] = f;
            len_decoded[r] += f ? 0 : 1;
            all &= f;
        }
    }
    if (finished) all = __syncthreads_and(all);
    __syncthreads();                                   // every thread has read state[] before it changes
    if (threadIdx.x == 0) {
        state[0] = t + 1;
        if (finished && all) state[1] = t + 1;
    }
}

}  // namespace

extern "C" int asr_argmax_rows(void* stream, const float* x, int64_t ld, int M, int V, int64_t* out) {
    ASR_REQUIRE(x && out && M > 0 && V > 0 && ld >= V, ASR_ERR_ARG, "argmax_rows: bad args");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, M, V, out);
    ASR_LAUNCH_CHECK("argmax_rows");
    return 0;
}

extern "C" int asr_log_softmax_rows(void* stream, const float* x, int64_t ldx, int M, int V, float* y, int64_t ldy) {
    ASR_REQUIRE(x && y && M > 0 && V > 0 && ldx >= V && ldy >= V, ASR_ERR_ARG, "log_softmax_rows: bad args");
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, M, V, y, ldy);
    ASR_LAUNCH_CHECK("log_softmax_rows");
    return 0;
}

extern "C" int asr_ctc_greedy_reduce(void* stream, const int64_t* frames, const int32_t* len, int B, int L, int blank, int64_t* out,
                                     int32_t* out_len) {
    ASR_REQUIRE(frames && len && out && out_len && B > 0 && L > 0, ASR_ERR_ARG, "ctc_greedy_reduce: bad args");
    hipLaunchKernelGGL(ctc_greedy_reduce_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), frames, len, L, blank, out, out_len);
    ASR_LAUNCH_CHECK("ctc_greedy_reduce");
    return 0;
}

extern "C" int asr_decode_embed(void* stream, const int64_t* cur, const float* emb, const float* pe, const int32_t* state, float* y32,
                                void* y16, int B, int D, int V, int max_pos) {
    ASR_REQUIRE(cur && emb && pe && state && y32 && B > 0 && D > 0 && D % 4 == 0 && V > 0 && max_pos > 0, ASR_ERR_ARG, "decode_embed: bad args");
    ASR_REQUIRE(asr_aligned(emb, 16) && asr_aligned(pe, 16) && asr_aligned(y32, 16), ASR_ERR_ALIGN, "decode_embed: alignment");
    hipLaunchKernelGGL(decode_embed_kernel, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), cur, emb, pe, state, y32,
                       reinterpret_cast<bf16_t*>(y16), B, D, V, max_pos);
    ASR_LAUNCH_CHECK("decode_embed");
    return 0;
}

extern "C" int asr_kv_cache_put(void* stream, const void* k_new, const void* v_new, void* k_cache, void* v_cache, const int32_t* state,
                                int BH, int Tmax, int dtype) {
    ASR_REQUIRE(k_new && v_new && k_cache && v_cache && state && BH > 0 && Tmax > 0, ASR_ERR_ARG, "kv_cache_put: bad args");
    ASR_REQUIRE(dtype == ASR_F32 || dtype == ASR_BF16, ASR_ERR_ARG, "kv_cache_put: dtype");
    const int esz = dtype == ASR_F32 ? 4 : 2;
    const int64_t n = (int64_t)BH * 16 * esz;
    hipLaunchKernelGGL(kv_cache_put_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const unsigned char*>(k_new), static_cast<const unsigned char*>(v_new), static_cast<unsigned char*>(k_cache),
                       static_cast<unsigned char*>(v_cache), state, BH, Tmax, esz);
    ASR_LAUNCH_CHECK("kv_cache_put");
    return 0;
}

extern "C" int asr_decode_advance(void* stream, const int64_t* cur, int64_t* preds, int32_t* state, int32_t* k_len, unsigned char* finished,
                                  int64_t* len_decoded, int eos, int B, int Tp1) {
    ASR_REQUIRE(cur && preds && state && k_len && finished && len_decoded && B > 0 && Tp1 > 1, ASR_ERR_ARG, "decode_advance: bad args");
    hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), cur, preds, state, k_len, finished,
                       len_decoded, eos, B, Tp1);
    ASR_LAUNCH_CHECK("decode_advance");
    return 0;
}

extern "C" int asr_topk_rows(void* stream, const float* x, int64_t ld, int M, int V, int k, float* vals, int64_t* idx) {
    ASR_REQUIRE(x && vals && idx && M > 0 && V > 0 && k > 0 && k <= V, ASR_ERR_ARG, "topk_rows: bad args (M=%d V=%d k=%d)", M, V, k);
    hipLaunchKernelGGL(topk_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, M, V, k, vals, idx);
    ASR_LAUNCH_CHECK("topk_rows");
    return 0;
}

extern "C" int asr_beam_prune(void* stream, const float* scores, const float* next_scores, const int64_t* next_preds, int B, int beam,
                              float* new_scores, int64_t* parent, int64_t* new_tok) {
    ASR_REQUIRE(scores && next_scores && next_preds && new_scores && parent && new_tok && B > 0, ASR_ERR_ARG, "beam_prune: bad args");
    ASR_REQUIRE(beam >= 1 && beam * beam <= 64, ASR_ERR_UNSUPPORTED, "beam_prune: beam_size %d (beam * beam must fit one wavefront)", beam);
    hipLaunchKernelGGL(beam_prune_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), scores, next_scores, next_preds, beam,
                       new_scores, parent, new_tok);
    ASR_LAUNCH_CHECK("beam_prune");
    return 0;
}

extern "C" int asr_beam_cat_frames(void* stream, const float* frames, const int32_t* state, const float* other32, const int64_t* cur,
                                   const float* emb, const float* pe, float* out, int N, int beam, int Tmax, int D, int D2, int V, int max_pos) {
    ASR_REQUIRE(frames && state && out && N > 0 && beam > 0 && N % beam == 0 && Tmax > 0 && D > 0 && D2 > 0 && D % 4 == 0 && D2 % 4 == 0,
                ASR_ERR_ARG, "beam_cat_frames: bad args");
    ASR_REQUIRE((cur && emb && pe && V > 0 && max_pos > 0) || (!cur && other32), ASR_ERR_ARG, "beam_cat_frames: needs `other32` or cur + emb + pe");
    ASR_REQUIRE(asr_aligned(frames, 16) && asr_aligned(out, 16) && asr_aligned(other32, 16) && asr_aligned(emb, 16) && asr_aligned(pe, 16),
                ASR_ERR_ALIGN, "beam_cat_frames: alignment");
    hipLaunchKernelGGL(beam_cat_frames_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), frames, state, other32, cur,
                       emb, pe, out, N, beam, Tmax, D, D2, V, max_pos);
    ASR_LAUNCH_CHECK("beam_cat_frames");
    return 0;
}

extern "C" int asr_beam_step(void* stream, float* scores, const float* next_scores, const int64_t* next_preds, int64_t* preds,
                             const int32_t* state, const int32_t* n_steps, int64_t* parent, int64_t* cur, int B, int beam, int W) {
    ASR_REQUIRE(scores && next_scores && next_preds && preds && state && n_steps && parent && cur && B > 0 && W > 1, ASR_ERR_ARG,
                "beam_step: bad args");
    ASR_REQUIRE(beam >= 1 && beam * beam <= 64, ASR_ERR_UNSUPPORTED, "beam_step: beam_size %d (beam * beam must fit one wavefront)", beam);
    ASR_REQUIRE(W <= BEAM_MAX_W, ASR_ERR_UNSUPPORTED, "beam_step: %d token columns (at most %d)", W, BEAM_MAX_W);
    hipLaunchKernelGGL(beam_step_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), scores, next_scores, next_preds, preds, state,
                       n_steps, parent, cur, beam, W);
    ASR_LAUNCH_CHECK("beam_step");
    return 0;
}

extern "C" int asr_beam_reorder_cache(void* stream, void* cache, const int64_t* parent, const int32_t* state, int n_kv, int B, int beam, int h,
                                      int Tmax, int dtype) {
    ASR_REQUIRE(cache && parent && state && n_kv > 0 && B > 0 && h > 0 && Tmax > 0, ASR_ERR_ARG, "beam_reorder_cache: bad args");
    ASR_REQUIRE(beam >= 1 && beam <= 8, ASR_ERR_UNSUPPORTED, "beam_reorder_cache: beam_size %d (at most 8)", beam);
    ASR_REQUIRE(dtype == ASR_F32 || dtype == ASR_BF16, ASR_ERR_ARG, "beam_reorder_cache: dtype");
    ASR_REQUIRE(asr_aligned(cache, 16), ASR_ERR_ALIGN, "beam_reorder_cache: alignment");
    const int segs = dtype == ASR_F32 ? 16 : 8;
    const int64_t n = (int64_t)n_kv * B * h * Tmax * segs;
    hipLaunchKernelGGL(beam_reorder_cache_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint4*>(cache), parent, state, n_kv, B, beam, h, Tmax, segs);
    ASR_LAUNCH_CHECK("beam_reorder_cache");
    return 0;
}

extern "C" int asr_beam_advance(void* stream, int32_t* state, int32_t* k_len, int N, const int64_t* cur, int eos, unsigned char* finished,
                                int64_t* len_decoded) {
    ASR_REQUIRE(state && k_len && N > 0, ASR_ERR_ARG, "beam_advance: bad args");
    ASR_REQUIRE(!finished || (cur && len_decoded), ASR_ERR_ARG, "beam_advance: `finished` needs cur and len_decoded");
    hipLaunchKernelGGL(beam_advance_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), state, k_len, N, cur, eos, finished,
                       len_decoded);
    ASR_LAUNCH_CHECK("beam_advance");
    return 0;
}

extern "C" int asr_lsm_topk_rows(void* stream, const float* x, int64_t ld, int M, int V, int k, int twice, float* vals, int64_t* idx) {
    ASR_REQUIRE(x && vals && idx && M > 0 && V > 0 && k > 0 && k <= V && ld >= V, ASR_ERR_ARG, "lsm_topk_rows: bad args (M=%d V=%d k=%d)", M, V, k);
    ASR_REQUIRE(V <= 256 * 18, ASR_ERR_UNSUPPORTED, "lsm_topk_rows: V = %d (at most 4608; use asr_log_softmax_rows + asr_topk_rows)", V);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (V <= 64 * 16) {       // short rows: one wavefront per row
        if (twice) hipLaunchKernelGGL((lsm_topk_rows_kernel<16, true>), dim3((M + 3) / 4), dim3(256), 0, s, x, ld, M, V, k, vals, idx);
        else hipLaunchKernelGGL((lsm_topk_rows_kernel<16, false>), dim3((M + 3) / 4), dim3(256), 0, s, x, ld, M, V, k, vals, idx);
    } else {
        if (twice) hipLaunchKernelGGL((lsm_topk_rows_wg_kernel<18, true>), dim3(M), dim3(256), 0, s, x, ld, M, V, k, vals, idx);
        else hipLaunchKernelGGL((lsm_topk_rows_wg_kernel<18, false>), dim3(M), dim3(256), 0, s, x, ld, M, V, k, vals, idx);
    }
    ASR_LAUNCH_CHECK("lsm_topk_rows");
    return 0;
}
