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
