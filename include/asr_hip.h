/*
 * asr_hip.h — C-ABI of libasr_hip.so: the MI355X (gfx950) Speech-Transformer / CTC / CIF forward+loss path.
 *
 * The reference (eastonYi/end-to-end_asr_pytorch) has no FFI / plugin boundary: it is 100 % Python calling stock
 * aten ops.  Each entry point below replaces one aten call site (cited as `src/...:line`, paths
 * relative to the reference tree) so that the Python host code in end-to-end_asr_pytorch_amd/ can mirror the
 * reference's nn.Module classes one-to-one.  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no torch / C++ types cross the boundary.
 *   - every pointer is a DEVICE pointer owned by the caller (torch); the library never allocates or frees device
 *     memory and keeps no state besides its loaded code objects and a thread-local error string.
 *   - `stream` is a hipStream_t (torch's current stream).  Calls are asynchronous, never synchronise, and are
 *     capturable into a hipGraph.
 *   - return value: 0 = ok, <0 = invalid argument (ASR_ERR_*), >0 = hipError_t.  asr_last_error() gives text.
 *   - dtype codes: ASR_F32 = 0 (exact-fp32 MFMA path), ASR_BF16 = 1 (bf16 MFMA inputs, fp32 accumulate).
 *   - lengths / indices on device: int32 lengths, int64 token ids (the reference's dtypes).
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_F32 0
#define ASR_BF16 1
#define ASR_F16 2      /* IEEE half: only the CTC branch's logits image (asr_vocab_proj_ctc -> asr_ctc_loss_bwd_ex) */

#define ASR_ERR_ARG (-1)       /* null pointer / non-positive size */
#define ASR_ERR_ALIGN (-2)     /* pointer or leading dimension not aligned as documented */
#define ASR_ERR_UNSUPPORTED (-3)

/* Dropout (nn.Dropout in training mode: attention.py:59,83, module.py:51, encoder.py:48, decoder.py:83,385,
 * attentionAssigner.py:35).  No mask tensor exists: the keep decision is a counter-based hash of the element's index, regenerated
 * by the backward kernels from the same descriptor.  For a tensor viewed as [N0, N1, N2] (attention probabilities
 * [h*B (index head*B + b, attention.py:43-49), Lq, Lk]; activations [B, L, D]):
 *     lowbias32(x): x ^= x>>16; x *= 0x7feb352d; x ^= x>>15; x *= 0x846ca68b; x ^= x>>16          (uint32 arithmetic)
 *     sub  = lowbias32(n0 * 0x9E3779B9 + key0)
 *     word = lowbias32((n1 * ceil(N2/2) + (n2 >> 1)) ^ sub) ^ key1
 *     r16  = (n2 & 1) ? word >> 16 : word & 0xFFFF
 *     element kept  <=>  r16 >= thr16;   kept elements are multiplied by 65536 / (65536 - thr16)
 * thr16 = round(p * 65536) (p quantised to 2^-16); thr16 == 0 disables dropout.  oracle/asr_oracle.py restates this in numpy.
 * `salt` (optional DEVICE pointer to one uint32, read when the kernel runs): a launch captured into a hipGraph keeps its by-value
 * arguments for every replay, so a captured training step points `salt` at the step counter of asr_step_tick and the kernels use
 *     key0' = key0 ^ lowbias32(*salt ^ 0x5bd1e995),   key1' = key1 ^ lowbias32(*salt + 0x27d4eb2f)
 * in place of (key0, key1): a fresh mask per replay, still a pure function of (descriptor, step).  NULL = keys as given. */
typedef struct asr_dropout {
    uint32_t thr16;
    uint32_t key0;
    uint32_t key1;
    const uint32_t* salt;
} asr_dropout_t;

/* y = dropout(x) over f32 [N0,N1,N2] (y may alias x); also the backward of itself (apply to the gradient). */
int asr_dropout_apply(void* stream, const float* x, float* y, int N0, int N1, int N2, asr_dropout_t drop);

/* GEMM epilogue flags */
#define ASR_GEMM_RELU 1u
#define ASR_GEMM_C_IS_ZERO 4u   /* the caller's C is already all zeros: a split-K launch (few output tiles, long K) skips its zeroing kernel */

/* ABI revision: 100 = rounds 1-4; 101 = asr_vocab_proj_lse and asr_ctc_loss_fwd_lse removed (asr_vocab_proj_ctc + asr_ctc_loss_fwd_table
 * replace them), asr_launch_budget_current added; 102 = asr_attn_ffn_fwd added.  A binder checks this before it resolves symbols. */
int asr_version(void);
/* Deterministic mode (also ASR_AMD_DETERMINISTIC=1 in the environment): the forward GEMMs stop splitting K across workgroups (float
 * atomics in arrival order) and the weight gradient's bias side product takes its single-writer form; the weight gradient itself
 * (asr_gemm_tn_ws) is order-fixed in either mode.  Returns the previous setting. */
int asr_set_deterministic(int on);
/* Launch budget of the calling host thread (0 = none; returns the previous value).  A launch queued on a side stream beside a chain
 * of small, latency-bound kernels must not fill the chip: a persistent GEMM workgroup holds its CU's registers / LDS for the whole
 * launch, and a streaming kernel at eight waves per SIMD leaves no wave slot - the small kernels' workgroups then find no CU to
 * start on until the side launch has drained (measured: the decoder's 12-18 us kernels took 100-150 us beside the CTC branch's
 * gradient pass and ctc_fc's two backward GEMMs).  With a budget of `cus`, asr_gemm_nn's persistent grid is at most `cus`
 * workgroups and asr_ctc_loss_bwd(_ex)'s gradient pass at most 3 * cus (three waves per SIMD when spread over the chip); asr_gemm_tn
 * takes its share as its own max_workgroups argument.  Results do not depend on the budget. */
int asr_launch_budget(int cus);
/* Ordering between two streams of ONE device without the system-scope fence a default HIP event performs at every record (cache
 * write-back and invalidation: ~1-3 us of the recording stream's time per event, ~100 events per eagerly queued training step):
 * asr_stream_order_after records `event` (from asr_event_create) on earlier_stream and makes later_stream wait for it - everything
 * queued on earlier_stream so far happens before anything queued on later_stream from now on.  The event can be reused at once. */
int asr_event_create(void** out_event);
int asr_stream_order_after(void* later_stream, void* earlier_stream, void* event);
int asr_event_destroy(void* event);
/* Timing events without the system-scope fence of a default HIP event (which writes back and invalidates the caches at every record: the
 * op behind it starts cold and the bracket measures the fence too).  asr_timer_elapsed_ms needs both events complete (synchronise the
 * stream or device first); destroy with asr_event_destroy. */
int asr_timer_create(void** out_event);
int asr_timer_record(void* event, void* stream);
int asr_timer_elapsed_ms(void* start, void* stop, float* ms);
/* Set-up time probe: do two streams share a hardware queue (their kernels then never overlap)?  The runtime multiplexes all streams
 * of the process onto a few queues in creation order; the trainer picks its side streams with this.  Synchronises both streams. */
int asr_streams_share_queue(void* stream_a, void* stream_b, int* shared);
/* Test support: one launch that leaves all 160 KiB of every CU's LDS filled with NaN bit patterns (0x7fc07fc0), so that a kernel
 * reading LDS bytes it has not yet been handed shows up deterministically (tests/test_gpu_lds_poison.py).  scratch4: 4 device bytes. */
int asr_debug_poison_lds(void* stream, void* scratch4);
const char* asr_last_error(void);

/* ------------------------------------------------------------------------------------------------------------
 * Dense projection  C[M,N] = act(A[M,K] . W[N,K]^T + bias[N])          (nn.Linear; W is [out,in] row-major)
 * replaces: encoder.py:49 linear_in, conv_encoder.py:124 affine, module.py:50 w_1 / w_2, attention.py:59 fc,
 *           transformer.py:148 ctc_fc, decoder.py:94 tgt_word_prj, decoder.py:387,397 (CIF), ctcModel/decoder.py:32.
 * A: a_dtype (f32 is converted on load when w_dtype is bf16);  W: w_dtype = the MFMA compute type;
 * bias: f32 or NULL;  C: c_dtype.  lda/ldw/ldc in elements.  16-byte alignment of A/W rows required
 * (K, lda, ldw multiples of 8 for bf16, 4 for f32); C may have any ldc.
 */
int asr_gemm_nt(void* stream, const void* A, int a_dtype, int64_t lda, const void* W, int w_dtype, int64_t ldw,
                const float* bias, void* C, int c_dtype, int64_t ldc, int M, int N, int K, unsigned flags);

/* asr_gemm_nt with more epilogue terms: C += addend (f32 [M,N], ld_add) and, when relu_mask (bf16 [M,N], ld_mask: the forward's
 * post-ReLU activations) is given, C = relu_mask > 0 ? C : 0.  Either may be NULL.
 * relu_bits_out (optional; with ASR_GEMM_RELU, bf16 operands and output, N % 128 == 0; ld_bits = N/8): the ReLU mask of
 * module.py:50 at 1 bit per element, for asr_gemm_nn(mask_is_bits).  A buffer of roundup(M, 128) * N/8 bytes in the epilogues'
 * own order - per 64 x 64 block (m/64, n/64; blocks row-major) 64 slots of 8 bytes, slot 8 (m & 7) + (n/8 & 7), byte (m/8 & 7)
 * of it, bit (n & 7) = (C[m,n] > 0) - so that a wave writes / reads one 8-byte word per lane and block. */
int asr_gemm_nt_ex(void* stream, const void* A, int a_dtype, int64_t lda, const void* W, int w_dtype, int64_t ldw,
                   const float* bias, void* C, int c_dtype, int64_t ldc, int M, int N, int K, unsigned flags,
                   const float* addend, int64_t ld_add, const void* relu_mask, int64_t ld_mask, void* relu_bits_out,
                   int64_t ld_bits);

/* Data gradient  C[M,N] = A[M,K] . Bm[K,N] (+bias) (+addend) (masked by relu_mask > 0): for nn.Linear with weight W [out,in],
 * dX = dY . W is A = dY [M,out], Bm = W (bf16, as stored: no transposed copy), K = out, N = in.  A f32 or bf16 with lda % 8 == 0
 * covering K rounded up to 8 (columns K..lda of A must be zero).  mask_is_bits != 0: relu_mask is the uint8 sign-bit image written
 * by asr_gemm_nt_ex(relu_bits_out) and ld_mask = N/8 (bf16 A, N % 128 == 0 only).  mask_is_bits & 2: C is already all zeros
 * (see ASR_GEMM_C_IS_ZERO). */
int asr_gemm_nn(void* stream, const void* A, int a_dtype, int64_t lda, const void* Bm, int64_t ldb, const float* bias, void* C,
                int c_dtype, int64_t ldc, int M, int N, int K, const float* addend, int64_t ld_add, const void* relu_mask,
                int64_t ld_mask, int mask_is_bits);

/* Head-major projection (attention.py:43-49: w_qs/w_ks/w_vs + view/permute/contiguous fused).
 * X[M = B*L, K] . W[n_proj*h*64, K]^T + bias -> out[p][B][h][L][64] for p < n_proj, out_dtype = w_dtype.
 * `proj_stride` = elements between consecutive projections' buffers.  If scale_first != 1, projection 0 (Q) is
 * multiplied by it: the 1/sqrt(d_k) of attention.py:77 and the log2(e) asr_attention_* expect are folded in here
 * (one rounding to the output dtype either way).
 */
int asr_proj_heads(void* stream, const void* X, int x_dtype, int64_t ldx, const void* W, int w_dtype, int64_t ldw,
                   const float* bias, void* out, int64_t proj_stride, int n_proj, int B, int L, int h, int K,
                   float scale_first);

/* Fused scaled-dot-product attention (attention.py:76-84 bmm -> /sqrt(dk) -> masked_fill(-inf) -> softmax -> bmm,
 * plus the un-permute of attention.py:56-57).  q [B,h,Lq,64] PRE-MULTIPLIED by log2(e)/sqrt(d_k), k,v [B,h,Lk,64] in `dtype`;
 * ctx [B,Lq,h*64] in `dtype`.  The scores q.k are therefore base-2 logits and the softmax is 2^s / sum 2^s - the same
 * probabilities as attention.py's exp(q.k/sqrt(d_k)), one v_exp_f32 each with no multiply in front.  Masks are expressed by
 * lengths, not tensors: key j of batch b is masked iff j >= k_len[b] (utils.py:157-165 / :146-154) or (causal && j > i)
 * (utils.py:135-143).  k_len may be NULL.  lse (f32 [B,h,Lq], BASE-2 log of the softmax denominator incl. max) is written
 * when non-NULL (for backward).
 * Dropout of the probabilities (attention.py:83; drop.thr16 != 0, bf16 only): the keep decisions are read from `drop_bits`, the
 * bit images asr_attention_dropmask() wrote for this (drop, B, h, Lq, Lk); the caller keeps the buffer for the backward.
 */
int asr_attention_fwd(void* stream, const void* q, const void* k, const void* v, int dtype, void* ctx, float* lse,
                      int B, int h, int Lq, int Lk, const int32_t* k_len, int causal, asr_dropout_t drop, const uint32_t* drop_bits);

/* Keep bits of one attention call's dropout, element (n0 = head*B + b, n1 = query, n2 = key) of asr_dropout_t's definition.
 * One hash pass instead of one per attention kernel (forward, dQ, dK/dV).  `bits`: asr_attention_dropmask_words() uint32 words,
 * holding the mask twice (Lq, Lk rounded up to 128 = Lqp, Lkp; BH = B*h; words of all-padding tiles unspecified):
 *   words [0, BH*Lkp/32*Lqp):     Mk[bh][key/32][q]  bit (key & 31)    (query-stationary kernels read along q)
 *   then  BH*Lqp/32*Lkp words:    Mq[bh][q/32][key]  bit (q & 31)      (the key-stationary dK/dV kernel reads along key) */
int64_t asr_attention_dropmask_words(int B, int h, int Lq, int Lk);
int asr_attention_dropmask(void* stream, asr_dropout_t drop, int B, int h, int Lq, int Lk, uint32_t* bits);
/* ... of n <= 8 dropout sites of one shape in one launch (host arrays drops[n], bits[n]: one buffer per site). */
int asr_attention_dropmask_multi(void* stream, int n, const asr_dropout_t* drops, uint32_t* const* bits, int B, int h, int Lq, int Lk);

/* Backward of asr_attention_fwd (bf16 only).  q,k,v as in the forward; o = the forward's ctx and d_o = its gradient, both
 * token-major bf16 [B,Lq,h*64]; lse from the forward.  delta: f32 workspace of asr_attention_bwd_workspace_floats(B, h, Lq) floats (the dQ
 * kernel leaves -rowsum(dO o O) and -lse there, padded to whole 64-query tiles, for the dK / dV kernel).  Outputs are token-major bf16:
 * dq[(b*Lq+i)*ldq + head*64 + d] (gradient wrt the UNSCALED query times `scale` = 1/sqrt(d_k), i.e. what the Q projection's
 * backward consumes; the log2(e) in q cancels against the base-2 softmax), dk / dv at
 * [(b*Lk+j)*ldkv + head*64 + d] - i.e. directly the A operands of the projection GEMMs' backward. */
int64_t asr_attention_bwd_workspace_floats(int B, int h, int Lq);
int asr_attention_bwd(void* stream, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                      const float* lse, float* delta, void* dq, int64_t ldq, void* dk, void* dv, int64_t ldkv, int B, int h,
                      int Lq, int Lk, const int32_t* k_len, int causal, float scale, asr_dropout_t drop, const uint32_t* drop_bits);
/* The two kernels of asr_attention_bwd as separate calls (dq first: it also produces delta, which dkv consumes). */
int asr_attention_bwd_dq(void* stream, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                         const float* lse, float* delta, void* dq, int64_t ldq, int B, int h, int Lq, int Lk,
                         const int32_t* k_len, int causal, float scale, asr_dropout_t drop, const uint32_t* drop_bits);
int asr_attention_bwd_dkv(void* stream, const void* q, const void* k, const void* v, const void* d_o, const float* lse,
                          const float* delta, void* dk, void* dv, int64_t ldkv, int B, int h, int Lq, int Lk,
                          const int32_t* k_len, int causal, asr_dropout_t drop, const uint32_t* drop_bits);

/* The fp32 parity-mode backward (partner of asr_attention_fwd with f32 operands): q (scaled as for the forward), k, v f32
 * [B,h,L,64]; o, d_o f32 token-major [B,Lq,h*64]; dq / dk / dv f32 token-major with row strides ldq / ldkv, same meaning as
 * asr_attention_bwd's outputs.  No dropout (the f32 mode runs in eval mode / with dropout 0).  A slow VALU kernel with float
 * atomics: it lets the whole backward tape run in fp32 so that gradients compare with the reference's at 1e-4. */
int asr_attention_bwd_f32(void* stream, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                          const float* lse, float* dq, int64_t ldq, float* dk, float* dv, int64_t ldkv, int B, int h, int Lq, int Lk,
                          const int32_t* k_len, int causal, float scale);

/* Decoder-sized rows (M = B*L of a few thousand): the projection and the residual + LayerNorm after it in ONE launch,
 *   s = dropout_x(A . W^T + bias) + residual;  y = LayerNorm(s) * gamma + beta, rows t >= row_len[b] zeroed
 * (attention.py:58-60: fc -> dropout -> + residual -> layer_norm; module.py:50-52 likewise for w_2) for exactly 256 outputs.
 * A bf16 [M,K] (lda), W bf16 [256,K] as stored, K % 32 == 0.  Writes what asr_add_layernorm_fwd(save) writes: s_out (the pre-norm sum),
 * y32, y16 (optional), mean / rstd (optional).  One workgroup owns 16 complete rows and reads the whole weight: meant for M <~ 4096.
 * (The encoder-sized variant of round 2, asr_gemm_add_layernorm, was slower than the GEMM + LayerNorm pair and is gone; the
 * encoder's feed-forward sub-layer is asr_ffn_fwd.) */
int asr_gemm_add_layernorm_small(void* stream, const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                                 const float* gamma, const float* beta, const int32_t* row_len, float* s_out, float* y32, void* y16,
                                 float* mean, float* rstd, int B, int L, int K, float eps, asr_dropout_t drop_x);

/* Multi-stream executor for a captured HIP graph (csrc/graph_exec.hip; host runtime of the training / decoding step - the reference
 * queues every op from Python and has no counterpart).  `hip_graph` is a hipGraph_t that stays owned by the caller and must
 * outlive the executor (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph(); the caller also keeps the capture's memory pool).
 * asr_graphx_create analyses it once: kernel, memset, memcpy and empty nodes, each placed on one of at most `max_streams` HIP streams
 * (a node continues the stream of one of its predecessors when it can), cross-stream edges become events.  asr_graphx_launch queues
 * one execution behind everything already queued on `stream` and joins the side streams back into it: the device sees free-running
 * queues that overlap like the eager step's, the host pays ~1-2 us per node.  Other node types (host nodes, child graphs, kernels
 * launched with an `extra` buffer) make create fail with a message - the caller then replays the graph with hipGraphLaunch. */
int asr_graphx_create(void* hip_graph, int max_streams, void** out_handle);
int asr_graphx_launch(void* handle, void* stream);
/* Rotate the plan's logical side streams over the physical ones (which streams share a hardware queue with the launch stream is the
 * runtime's choice; a caller times the rotations 0 .. n_streams - 2 once and keeps the best). */
int asr_graphx_set_rotation(void* handle, int rotation);
/* Place the plan's side streams by probing which physical streams share a hardware queue (asr_streams_share_queue on the pairs, a few
 * ms, synchronising): busiest logical stream first, onto the least loaded queue, the launch stream's queue last.  clear != 0 returns to
 * the rotation.  launch_stream: the stream asr_graphx_launch will be given. */
int asr_graphx_place_streams(void* handle, void* launch_stream, int clear);
int asr_graphx_info(void* handle, int* n_nodes, int* n_kernels, int* n_streams, int* n_events);
int asr_graphx_destroy(void* handle);

/* Gradient all-reduce of the data-parallel step (csrc/collective.hip; SURVEY §8(e) - the reference trains on one GPU, solver.py:80-96,
 * and has no counterpart: each rank runs the whole step on its shard of the global batch and the flat f32 gradient is summed bucket
 * by bucket).  RCCL is bound at run time: asr_rccl_load(path) (NULL: the librccl already mapped into the process, e.g. torch's).
 * asr_rccl_unique_id fills ASR_RCCL_ID_BYTES bytes on ONE rank; the caller carries them to the others (bootstrap only) and every rank
 * calls asr_rccl_comm_create(id, nranks, rank) with its device current.  asr_rccl_all_reduce_f32 queues buf <- sum over ranks, in
 * place, on `stream`.  asr_rccl_comm_check returns the communicator's asynchronous error state (0 = healthy).  asr_rccl_comm_abort tears
 * a communicator down without waiting for outstanding collectives (peers see an asynchronous error instead of waiting forever);
 * when asr_graphx_launch fails part-way through a step with collective nodes - fatal for the job - it returns ASR_ERR_COLLECTIVE_STEP and
 * forgets the (borrowed) communicator: the OWNER then calls asr_rccl_comm_abort on it and drops its handle (ops.RcclComm.abort). */
#define ASR_RCCL_ID_BYTES 128
#define ASR_ERR_COLLECTIVE_STEP (-6)
int asr_rccl_load(const char* path);
int asr_rccl_version(int* version);
int asr_rccl_unique_id(void* out_id);
int asr_rccl_comm_create(const void* id, int nranks, int rank, void** out_comm);
int asr_rccl_comm_destroy(void* comm);
int asr_rccl_comm_abort(void* comm);
int asr_rccl_all_reduce_f32(void* comm, float* buf, long long count, void* stream);
int asr_rccl_comm_check(void* comm);
/* The ready point of a gradient bucket: a no-op kernel node on `stream` carrying (buf, count, tag).  Launched while a step is being
 * captured it becomes a COLLECTIVE node of the executor: asr_graphx_launch queues the all-reduce of buf[0..count) on that node's stream
 * from its own C loop, ordered by the node's edges like any kernel - through the RCCL communicator given to asr_graphx_set_collective,
 * or through `fn(ctx, buf, count, tag, stream)` when one is given (a rig whose ranks are not RCCL peers; 0 = queued).  A plan with
 * collective nodes and neither fails at launch.  asr_graphx_collectives: how many such nodes the plan holds, and their total count. */
typedef int (*asr_collective_fn)(void* ctx, float* buf, long long count, int tag, void* stream);
int asr_collective_mark(float* buf, long long count, int tag, void* stream);
int asr_graphx_set_collective(void* handle, void* rccl_comm, asr_collective_fn fn, void* ctx);
int asr_graphx_collectives(void* handle, int* n_collective, long long* total_count);

/* The position-wise feed-forward sub-layer of an encoder layer in ONE launch (module.py:48-53 `layer_norm(dropout(w_2(relu(w_1(x)))) +
 * residual)` followed by encoder.py:77 `enc_output *= non_pad_mask`), for d_model = 256 and d_ff a multiple of 64 (<= 2048):
 *   h = relu(x16 . W1^T + b1);  s = dropout_x(h . W2^T + b2) + x32;  y = LayerNorm(s) * gamma + beta, rows t >= row_len[b] zeroed.
 * x16 bf16 [M = B*L, 256] (the MFMA operand), x32 f32 [M, 256] (the residual); w1 bf16 [d_ff, 256], w2 bf16 [256, d_ff] as nn.Linear
 * stores them.  The [M, d_ff] hidden activation stays on the chip between the two products.  Outputs as asr_add_layernorm_fwd
 * leaves them: s_out (pre-norm sum, optional), y32, y16 (optional), mean / rstd (optional).  Training passes hid_out (bf16 [M, d_ff]:
 * the activation, written once for the weight gradient dW2 = ds^T h) AND bits_out (asr_ffn_bits_words(M, d_ff) 32-bit words: 1 bit per
 * hidden unit, set where h > 0; the layout is private to asr_ffn_fwd / asr_ffn_bwd); both NULL in inference. */
int64_t asr_ffn_bits_words(int M, int d_ff);
int asr_ffn_fwd(void* stream, const void* x16, const float* x32, const void* w1, const float* b1, const void* w2, const float* b2,
                const float* gamma, const float* beta, const int32_t* row_len, void* hid_out, void* bits_out, float* s_out,
                float* y32, void* y16, float* mean_out, float* rstd_out, int B, int L, int d_model, int d_ff, float eps,
                asr_dropout_t drop_x);

/* The attention sub-layer's tail at encoder size in one launch (attention.py:58-60: fc -> dropout -> + residual -> layer_norm;
 * encoder.py:77's row mask): y = LayerNorm(dropout_x(ctx . W^T + bias) + residual) * gamma + beta, rows t >= row_len[b] zeroed.
 * ctx16 bf16 [B*L, 256] (the attention context, h * d_v = 256), w bf16 [256, 256] as stored, residual f32 [B*L, 256].  Same workgroup
 * shape, prologue and epilogue as asr_ffn_fwd (the projection is that kernel's first product over four 64-row chunks of w); writes
 * what asr_add_layernorm_fwd(save) writes: s_out (pre-norm sum, training; may be NULL with mean_out / rstd_out given - the backward then
 * takes x^ from y32, asr_add_layernorm_bwd_y) / mean_out / rstd_out (training), y32, y16 (optional). */
int asr_proj_ln_fwd(void* stream, const void* ctx16, const float* residual, const void* w, const float* bias, const float* gamma,
                    const float* beta, const int32_t* row_len, float* s_out, float* y32, void* y16, float* mean_out, float* rstd_out,
                    int B, int L, int d_model, float eps, asr_dropout_t drop_x);
/* asr_proj_ln_fwd AND asr_ffn_fwd as ONE launch - the two row-wise sub-layer tails of an encoder layer behind its attention
 * (encoder.py:74-77: `slf_attn`'s fc / dropout / residual / layer_norm, the row mask, then `pos_ffn`), d_model = 256:
 *   x = LayerNorm0(dropout0(ctx16 . Wo^T + bo) + residual) [rows >= row_len zeroed]      -> x32 / x16 (OUTPUTS here), s0 / mean0 / rstd0
 *   y = LayerNorm(dropout_x(relu(x . W1^T + b1) . W2^T + b2) + x) [rows >= row_len zeroed] -> everything asr_ffn_fwd writes
 * Arguments as in the two entry points it replaces (s0_out may be NULL with mean0_out / rstd0_out given, like asr_proj_ln_fwd's);
 * both dropout sites active or neither.  Results are those of the two launches, bit for bit: a workgroup owns the same 128 tokens in
 * both phases and reads its own x rows back between them. */
int asr_attn_ffn_fwd(void* stream, const void* ctx16, const float* residual, const void* wo, const float* bo, const float* gamma0,
                     const float* beta0, float eps0, asr_dropout_t drop0, float* s0_out, float* x32, void* x16, float* mean0_out,
                     float* rstd0_out, const void* w1, const float* b1, const void* w2, const float* b2, const float* gamma, const float* beta,
                     const int32_t* row_len, void* hid_out, void* bits_out, float* s_out, float* y32, void* y16, float* mean_out,
                     float* rstd_out, int B, int L, int d_model, int d_ff, float eps, asr_dropout_t drop_x);
/* Data gradient of the same sub-layer in ONE launch (autograd of module.py:50-52 below the LayerNorm):
 *   dH = (ds16 . W2) * [h > 0]  (bits from asr_ffn_fwd);   dx = dH . W1 + ds32.
 * ds16 bf16 [M, 256]: the gradient wrt w_2's output (asr_add_layernorm_bwd's ds16); ds32 f32 [M, 256]: the gradient wrt the residual.
 * dhid_out bf16 [M, d_ff] (written once, for dW1 = dH^T x and db1 = colsum dH), dx_out f32 [M, 256]. */
int asr_ffn_bwd(void* stream, const void* ds16, const float* ds32, const void* w1, const void* w2, const void* bits, void* dhid_out,
                float* dx_out, int M, int d_model, int d_ff);
/* asr_ffn_bwd with the backward of the LayerNorm that produced the sub-layer's input folded into its epilogue (encoder.py:74-76: the
 * feed-forward sub-layer is the only reader of the attention sub-layer's output, attention.py:60): dx never goes to memory; the launch
 * leaves what asr_add_layernorm_bwd(dy = dx, s = ln_s, ln_mean, ln_rstd, ln_gamma, row_len, drop_x) leaves - ds_out (f32, gradient wrt
 * that LayerNorm's residual), ds16_out (bf16, gradient wrt the normalised projection's output, dropout-masked), and dgamma / dbeta /
 * dbias (optional) ACCUMULATED into - with the same arithmetic per row (a row lies across a wave in both kernels).  M = B * L. */
/* Data gradient of an nn.Linear whose input is d_model = 256 wide, as a row-block kernel (csrc/dgrad_rows.hip): a workgroup owns 128
 * complete rows of  dX[M, 256] = dY[M, K] . W[K, 256] (+ addend),  W bf16 as nn.Linear stores it ([out = K][in = 256]), dY bf16 with row
 * stride ldy elements (a column slice of a wider buffer is fine), K a multiple of 64.  Autograd of attention.py:43-49 (w_qs / w_ks / w_vs),
 * attention.py:58 (fc) and the decoder's cross K / V projections at encoder size, where a tiled GEMM is all prologue and epilogue.
 * asr_dgrad_rows: out f32 (ASR_F32, addend optional) or bf16 (ASR_BF16) [M, 256].
 * asr_dgrad_rows_ln: dX + addend is the dy of the LayerNorm that produced the projection's input and nothing else reads it: the launch
 * leaves what asr_add_layernorm_bwd(dy, ln_s, ln_mean, ln_rstd, ln_gamma, row_len, drop_x) leaves - ds_out f32, ds16_out bf16,
 * dgamma / dbeta / dbias (optional) accumulated into - and dx never goes to memory (M = B * L).  ln_beta non-null (here and in
 * asr_ffn_bwd_ln): ln_s holds that LayerNorm's OUTPUT y instead of its pre-norm sum, x^ = (y - beta) / gamma, ln_mean is not read -
 * asr_add_layernorm_bwd_y's convention, for a forward that did not store the pre-norm sum. */
int asr_dgrad_rows(void* stream, const void* dy, int64_t ldy, const void* w, const float* addend, void* out, int out_dtype, int M, int K,
                   int d_model);
int asr_dgrad_rows_ln(void* stream, const void* dy, int64_t ldy, const void* w, const float* addend, int B, int L, int K, int d_model,
                      const float* ln_s, const float* ln_mean, const float* ln_rstd, const float* ln_gamma, const float* ln_beta,
                      const int32_t* row_len, float* ds_out, void* ds16_out, float* dgamma, float* dbeta, float* dbias,
                      asr_dropout_t drop_x);
int asr_ffn_bwd_ln(void* stream, const void* ds16, const float* ds32, const void* w1, const void* w2, const void* bits, void* dhid_out,
                   int B, int L, int d_model, int d_ff, const float* ln_s, const float* ln_mean, const float* ln_rstd,
                   const float* ln_gamma, const float* ln_beta, const int32_t* row_len, float* ds_out, void* ds16_out, float* dgamma,
                   float* dbeta, float* dbias, asr_dropout_t drop_x);

/* y = LayerNorm(x [+ residual]) * gamma + beta [+ pe[t]] ; rows with t >= row_len[b] are zeroed when row_len given.
 * (attention.py:60, module.py:52, encoder.py:48-50,74,77).  x, residual, y32 f32 [M = B*L, D]; y16 optional bf16 copy.
 * mean/rstd (f32 [M]) and s_out (f32 [M,D], the pre-norm sum x+residual; may alias x) are optional saves for backward.
 * drop_x: dropout applied to x before the residual add (attention.py:59-60, module.py:51-52: LN(dropout(fc(..)) + residual));
 * drop_y: dropout applied to the output after the PE add (encoder.py:48-50: dropout(LN(linear(x)) + PE)).  Both over [B,L,D].
 */
int asr_add_layernorm_fwd(void* stream, const float* x, const float* residual, const float* gamma, const float* beta,
                          const float* pe, const int32_t* row_len, float* y32, void* y16, float* mean, float* rstd,
                          float* s_out, int B, int L, int D, float eps, asr_dropout_t drop_x, asr_dropout_t drop_y);

/* Backward of asr_add_layernorm_fwd: s = the pre-norm sum x+residual (f32 [M,D]), mean/rstd from the forward.
 * ds (f32, optional bf16 copy ds16) = gradient wrt s (= wrt x and wrt residual); rows t >= row_len[b] get zero and do
 * not contribute.  dgamma/dbeta (f32 [D]) are ACCUMULATED into (caller zeroes them); dbias (optional, f32 [D]) likewise
 * receives colsum(ds) = the bias gradient of the projection whose output was normalised.
 * drop_y (the forward's) masks dy first.  With drop_x, ds (f32) stays the gradient wrt the residual while ds16 and dbias carry
 * the gradient wrt x = dropout-masked ds (the operand of the projection's backward GEMMs). */
/* asr_add_layernorm_bwd for a forward that kept no pre-norm sum: y is the LayerNorm's OUTPUT (alive anyway as the next sub-layer's input
 * and residual), x^ = (y - beta) / gamma (0 where gamma is 0 and in masked rows); rstd from the forward.  No drop_y form. */
int asr_add_layernorm_bwd_y(void* stream, const float* dy, const float* y, const float* rstd, const float* gamma, const float* beta,
                            const int32_t* row_len, float* ds, void* ds16, float* dgamma, float* dbeta, float* dbias, int B, int L,
                            int D, asr_dropout_t drop_x);
int asr_add_layernorm_bwd(void* stream, const float* dy, const float* s, const float* mean, const float* rstd,
                          const float* gamma, const int32_t* row_len, float* ds, void* ds16, float* dgamma, float* dbeta,
                          float* dbias, int B, int L, int D, asr_dropout_t drop_x, asr_dropout_t drop_y);

/* Weight gradient  C[N,K] (+)= sum_m A[m,n] * B[m,k]  (nn.Linear: A = dY [M,N], B = X [M,K] -> dW).  A, B f32 or bf16
 * (converted to bf16 MFMA operands on load); C f32.  zero_first != 0 clears C first (the kernel accumulates with
 * float atomics across M-splits).  lda / ldb multiples of 4 and >= N / K rounded up to 4 (A and B may live in padded buffers).
 * colsum (optional, f32 [N]): += sum_m A[m,n], the bias gradient, accumulated in the same pass (caller zeroes it).
 * max_workgroups: 0 = the kernel's own choice (2 workgroups per CU: the launch has the chip to itself); a caller that queues the
 * GEMM beside other kernels (a weight-gradient side stream) passes 256 = one per CU, half the M-splits and atomics. */
int asr_gemm_tn(void* stream, const void* A, int a_dtype, int64_t lda, const void* Bm, int b_dtype, int64_t ldb, float* C,
                int64_t ldc, int M, int N, int K, int zero_first, float* colsum, int max_workgroups);
/* The same weight gradient with a caller-owned workspace (csrc/wgrad.hip; solver.py:119 loss.backward() -> every nn.Linear's
 * weight.grad): bf16 A and B.  The M-splits' partial tiles go to the workspace with plain stores and a second small launch sums
 * them in a fixed order - no float atomics on C (they retire at ~1.2 TB/s on MI355X, plain stores at > 4), no pre-zeroing, and a
 * result that does not depend on timing.  accumulate != 0: C += the product.  colsum as above (float atomics over the K/128
 * workgroups that share a column block); deterministic != 0: one writer per colsum element as well.
 * workspace: asr_gemm_tn_ws_bytes(M, N, K, max_workgroups) bytes, 16-byte aligned, contents irrelevant, not shared by two launches
 * that may run concurrently.  Shapes the kernel does not take (K % 128, rows not 16-byte aligned, M < 64, M * ld * 2 >= 2^31,
 * workspace null or short) fall through to asr_gemm_tn. */
int64_t asr_gemm_tn_ws_bytes(int M, int N, int K, int max_workgroups);
int asr_gemm_tn_ws(void* stream, const void* A, int64_t lda, const void* Bm, int64_t ldb, float* C, int64_t ldc, int M, int N, int K,
                   int accumulate, float* colsum, int max_workgroups, void* workspace, int64_t workspace_bytes, int deterministic);
/* Up to 16 of those weight gradients in ONE pair of launches (the decoder's weights: 1632 rows each, a single launch is 13-17 us of
 * which ~2 are work; or all weight gradients of one or two encoder layers).  Every problem must satisfy asr_gemm_tn_ws's own
 * conditions for the slab kernel (else ASR_ERR_UNSUPPORTED and nothing is launched: issue them one by one) and brings a workspace of
 * asr_gemm_tn_ws_bytes(M, N, K, 0) bytes.  _wgs: with an explicit workgroup budget for the launch, shared out by output tiles (0 = the
 * default, half the CUs); a problem never gets fewer workgroups than tiles, so a budget <= the tile total runs every problem
 * unsplit over M - no partial tiles, no reduce launch. */
typedef struct {
    const void* A; int64_t lda;      /* dY [M, lda] bf16 */
    const void* B; int64_t ldb;      /* X  [M, ldb] bf16 */
    float* C; int64_t ldc;           /* dW [N, ldc] */
    int M, N, K, accumulate;
    float* colsum;                   /* optional bias gradient [N], += */
    void* workspace; int64_t workspace_bytes;
} asr_tn_problem_t;
int asr_gemm_tn_ws_group(void* stream, int n, const asr_tn_problem_t* problems, int deterministic);
int asr_gemm_tn_ws_group_wgs(void* stream, int n, const asr_tn_problem_t* problems, int deterministic, int group_workgroups);
/* Bias gradient out[n] (+)= sum_m A[m,n]. */
int asr_colsum(void* stream, const void* A, int a_dtype, int64_t lda, int M, int N, float* out, int zero_first);
/* Embedding backward: demb[ids[r], :] += dropout_mask(dy[r, :])  (f32 atomics; caller zeroes demb).  M = B*U rows. */
int asr_embed_bwd(void* stream, const int64_t* ids, const float* dy, int B, int U, int D, int V, float* demb, asr_dropout_t drop);
/* Fused Adam over a flat buffer (torch.optim.Adam semantics, no weight decay): g is multiplied by grad_scale first;
 * p16 (optional) receives the bf16 copy of the updated parameters (the MFMA operand shadow). */
int asr_adam_step(void* stream, float* p, const float* g, float* m, float* v, void* p16, int64_t n, float lr, float beta1,
                  float beta2, float eps, int step, float grad_scale);
/* Step state on the device, for a training step replayed from a hipGraph (scalars passed by value would be frozen at capture):
 * state is 8 x 32-bit words {uint32 step, float lr, float 1-beta1^step, float sqrt(1-beta2^step), 4 spare}.  asr_step_tick adds 1 to
 * `step` and recomputes the others - lr = k * init_lr * min(step^-0.5, step * warmup^-1.5), the Noam schedule of
 * src/transformer/optimizer.py:24-29, evaluated in f64 - and asr_adam_step_dev is asr_adam_step reading (lr, bias corrections) from it.
 * &state[0] is also what asr_dropout_t.salt points at. */
int asr_step_tick(void* stream, uint32_t* state, float k, float init_lr, float warmup, float beta1, float beta2);
int asr_adam_step_dev(void* stream, float* p, const float* g, float* m, float* v, void* p16, int64_t n, const uint32_t* state,
                      float beta1, float beta2, float eps, float grad_scale);

/* Decoder.preprocess (src/transformer/decoder.py:42-58) in one launch: per utterance strip the pad (0) entries of targets[b, 0..U)
 * in order, ys_in[b] = <sos> tokens 0.. and ys_out[b] = tokens <eos> 0.., both [B, W] with W = longest target + 1 (the caller knows
 * it - the loader does - or reads it back once).  Optional outputs: in_len[b] = number of positive entries of ys_in[b] (the decoder's
 * non-pad length, decoder.py:83), n_out[b] = tokens kept, *overflow set to 1 if a row held more than W - 1 tokens (the rest dropped). */
int asr_decoder_targets(void* stream, const int64_t* targets, int64_t* ys_in, int64_t* ys_out, int32_t* in_len, int64_t* n_out,
                        int32_t* overflow, int B, int U, int W, int64_t sos_id, int64_t eos_id);

/* Decoder_CIF.preprocess (src/transformer/decoder.py:356-366): ys_in[b, u] = (<sos>, target[b, 0..U-1))[u] * (target[b, u] > 0), and
 * (optional) in_len[b] = number of positive targets of row b - the decoder's non-pad length. */
int asr_decoder_cif_targets(void* stream, const int64_t* target, int64_t* ys_in, int32_t* in_len, int B, int U, int64_t sos_id);

/* Embedding gather + positional encoding (decoder.py:83): out[b,u,:] = dropout(emb[ids[b,u],:] + pe[u,:]). */
int asr_embed_pe_fwd(void* stream, const int64_t* ids, const float* emb, const float* pe, float* y32, void* y16,
                     int B, int U, int D, int V, asr_dropout_t drop);

/* Conv2dSubsample (conv_encoder.py:101-108).  Layer 0: feats f32 [B,T,D] with the implicit zero right-pad of
 * conv_encoder.py:104 -> relu(conv 1->32, 3x3, stride (2,1)) -> y [B,T1,F1,32] channel-last in `dtype`; only the
 * T1 x F1 region later layers need is produced.  w0 f32 [32,1,3,3], b0 f32 [32].
 */
int asr_conv_sub0_fwd(void* stream, const float* feats, const float* w0, const float* b0, void* y, int dtype,
                      int B, int T, int D, int T1, int F1);
/* Layer i>=1: x [B,Tin,Fin,32] channel-last -> relu(conv 32->32, 3x3, stride (2,1)).  If last != 0 the output is
 * written as [B,Tout,32*Fout] with column c*Fout+f (the permute/view of conv_encoder.py:108), else channel-last.
 * w f32 [32,32,3,3], b f32 [32].
 */
int asr_conv_sub1_fwd(void* stream, const void* x, const float* w, const float* b, void* y, int dtype,
                      int B, int Tin, int Fin, int Tout, int Fout, int last);

/* Backward helpers for the conv layers (their gradients run on asr_gemm_tn / asr_gemm_nn over an explicit patch matrix):
 * im2col: col[(b,t,f), tap*C + c] = x[b, 2t+kh, f+kw, c] for the 3x3 stride-(2,1) conv (zero outside x; pad columns zeroed),
 * x channel-last [B,Tin,Fin,C];  col2im_relu: dx[b,ti,fi,:] = (y > 0) * gather-sum of dcol (bf16, 32 channels). */
int asr_conv_im2col(void* stream, const void* x, int x_dtype, int C, void* col, int col_dtype, int ldc, int B, int Tin, int Fin,
                    int Tout, int Fout);
int asr_conv_col2im_relu(void* stream, const void* dcol, int ldc, const void* y, void* dx, int B, int Tin, int Fin, int Tout,
                         int Fout);
/* Direct backward of a 32 -> 32 channel conv layer (conv_encoder.py:104, bf16), without the patch matrix:
 * asr_conv_sub1_bwd_x: dx[b,ti,fi,ci] = (xin > 0) * sum_{kh,kw,co} dy[b,(ti-kh)/2,fi-kw,co] * w[co][ci][kh][kw]  (dy bf16 [B,Tout,Fout,32],
 *   w f32 [32,32,3,3] as stored, xin = the layer's input activation bf16 [B,Tin,Fin,32] = the ReLU mask, dx like xin).
 * asr_conv_sub1_bwd_w: dw f32 [32, 288] (+=, column tap*32 + ci: the conv-as-GEMM weight matrix, see asr_add_transposed) and
 *   db f32 [32] (+=, optional) from dy and the layer's input x; Fout <= 48, Fin <= 50 (the staged tile).  workspace: f32
 *   [asr_conv_sub1_bwd_w_workspace_floats()] - the workgroups' partial sums, added up by a second small launch. */
int asr_conv_sub1_bwd_x(void* stream, const void* dy, const float* w, const void* xin, void* dx, int B, int Tin, int Fin, int Tout, int Fout);
int64_t asr_conv_sub1_bwd_w_workspace_floats(void);
/* ... and of the first layer (1 -> 32 channels, conv_encoder.py:103): dw f32 [32, 9] (+=, the nn.Conv2d weight [32,1,3,3] as stored),
 * db f32 [32] (+=) from dy bf16 [B,T1,F1,32] and the features f32 [B,T,D]; same workspace; F1 <= 48. */
int asr_conv_sub0_bwd_w(void* stream, const void* dy, const float* feats, float* dw, float* db, float* workspace, int B, int T, int D, int T1,
                        int F1);
int asr_conv_sub1_bwd_w(void* stream, const void* dy, const void* x, float* dw, float* db, float* workspace, int B, int Tin, int Fin,
                        int Tout, int Fout);
/* ... in the fp32 parity mode (dcol, y, dx f32). */
int asr_conv_col2im_relu_f32(void* stream, const float* dcol, int ldc, const float* y, float* dx, int B, int Tin, int Fin, int Tout,
                             int Fout);

/* ------------------------------------------------------------------------------------------------------------
 * CTC loss (loss.py:41-43 / ctcModel/loss.py:9-11: F.log_softmax(dim=-1) -> F.ctc_loss(blank=V-1)).
 * logits f32 [B,L,V] with row stride ldl (elements) and batch stride L*ldl; in_len int32 [B]; targets int64
 * [B,Umax] zero-padded; target length = number of non-zero ids per row (loss.py:40).
 * Workspaces (caller-owned, opaque layout): lse f32 [B,L]; lp_ext f32 [B,L,S]; alpha f32 [B,L+2,S] with
 * S = asr_ctc_workspace_stride(Umax) (>= 2*Umax+1; one 512-byte wave store per row and 64 labels).
 * Outputs: nll f32 [B] (inf for infeasible rows, zero_infinity=False), tgt_len int32 [B].
 * n_chunks (forward only): > 1 (and U + 1 <= 64, L >= 64) selects the FUSED form - one launch in which persistent pass workgroups
 * walk the logits chunk-major (frames cut outside-in into n_chunks pieces per direction), publish the table rows (write-through
 * stores + per-chunk arrival counters kept in the alpha workspace) and the two recursion wavefronts of every utterance, resident
 * in the same grid, consume each piece as it arrives through an LDS ring: the HBM-bound pass and the latency-bound alpha / beta
 * recursion overlap.  <= 1: two launches (pass, then recursion).  Results are bit-identical.
 * zero_counters: NULL, or an int32 buffer of asr_ctc_counter_words(B, L, n_chunks) words that are ZERO when the kernel starts (a slice of
 * a per-step zero arena, or a buffer zeroed once): the arrival counters, the pass workgroups' queue heads and the finished-utterance
 * count then live there and the call queues no memset in front of its launch.  The kernel leaves the words zero again when it ends
 * (the last utterance to finish clears them), so ONE buffer per stream serves every call on that stream.
 * asr_ctc_loss_mean_fwd is the same call with the batch reduction of asr_ctc_mean folded in (loss.py:41-43 in one launch).
 */
/* The CTC branch's vocabulary projection in the training step (transformer.py:119,148 `ctc_fc`: Linear d_model -> V without bias,
 * d_model = 256; loss.py:41-43 behind it), with the logits never read back for the forward: logits = x16 [B*L, 256] . w16 [V, 256]^T, logits16 IEEE fp16 [B*L, ldl] (ldl % 8 == 0, pad columns zero) - their only later reader is asr_ctc_loss_bwd_ex, which
 * takes softmax = exp(logit - lse) from them: fp16's 11-bit significand puts <= 0.2 % on a probability at |logit| <= 8 where bf16's 8
 * bits put 1.6 %, more than the bf16 gradient image's own rounding -, lse f32 [B*L], and lp_ext f32 [B*L, 128]: the CTC table rows asr_ctc_loss_fwd builds by a pass over
 * the logits ((x[blank or label] - lse) log2 e per state of the utterance's extended label sequence, -inf beyond), taken here from
 * the fp32 accumulators as each 64-column chunk passes through LDS - bit-identical to that pass on f32 logits.  targets int64
 * [B, Umax] (0 = pad); L >= 128, Umax + 1 <= 64, d_model = 256.  asr_ctc_loss_fwd_table then runs the alpha / beta recursion on the
 * finished table (workspaces and outputs of asr_ctc_loss_fwd; loss: optional f32 [1] batch mean). */
int asr_vocab_proj_ctc(void* stream, const void* x16, const void* w16, void* logits16, int64_t ldl, float* lse, float* lp_ext,
                       const int64_t* targets, int B, int L, int V, int Umax, int blank, int d_model);
int asr_ctc_loss_fwd_table(void* stream, const float* lp_ext, const int32_t* in_len, const int64_t* targets, int B, int L, int Umax,
                           float* alpha, float* nll, int32_t* tgt_len, float* loss);
int asr_ctc_workspace_stride(int Umax);
int asr_ctc_loss_fwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets,
                     int B, int L, int V, int Umax, int blank, float* lse, float* lp_ext, float* alpha, float* nll,
                     int32_t* tgt_len, void* zero_counters, int n_chunks);
int asr_ctc_loss_mean_fwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets,
                          int B, int L, int V, int Umax, int blank, float* lse, float* lp_ext, float* alpha, float* nll,
                          int32_t* tgt_len, void* zero_counters, int n_chunks, float* loss);
/* number of 32-bit words of the optional caller-zeroed counter buffer of the fused forward (0: the call would not take the fused form) */
int64_t asr_ctc_counter_words(int B, int L, int n_chunks);
/* mean_b(nll_b / max(tgt_len_b,1))  — reduction='mean' of F.ctc_loss.  loss: f32 [1]. */
int asr_ctc_mean(void* stream, const float* nll, const int32_t* tgt_len, int B, float* loss);
/* Gradient wrt logits of gout * mean-reduced loss: g[b,t,v] = gout/(B*max(tgt_len_b,1)) * (softmax - occupancy),
 * zero for t >= in_len[b].  Consumes (and overwrites) alpha with the occupancies.  grad: `grad_dtype` f32 or bf16, row stride
 * ldg elements.  bf16 (a gradient that only feeds bf16 MFMA GEMMs - those round an f32 image to the same values on load): rows
 * 16-byte aligned (ldg % 8 == 0, ldl % 4 == 0) and columns V .. ldg-1 are WRITTEN as zeros, so ldg can be the GEMM kernels'
 * padded row width (asr_gemm_nn / asr_gemm_tn below).
 * alpha2 (optional, a second workspace of alpha's size): the recursion's second half then stores its raw rows there instead of
 * turning alpha's rows into occupancies in place - a chain step without the second row load and the exp2, about half as long -
 * and the gradient pass forms each occupancy from the two rows itself; alpha is left as the forward wrote it.
 */
int asr_ctc_loss_bwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets,
                     int B, int L, int V, int Umax, int blank, const float* lse, const float* lp_ext, float* alpha,
                     const float* nll, const int32_t* tgt_len, const float* gout, void* grad, int grad_dtype, int64_t ldg,
                     float* alpha2);
/* ... with the logits' element type given: ASR_F16 = the image asr_vocab_proj_ctc wrote (ldl % 8 == 0, bf16 gradient only). */
int asr_ctc_loss_bwd_ex(void* stream, const void* logits, int logits_dtype, int64_t ldl, const int32_t* in_len, const int64_t* targets,
                        int B, int L, int V, int Umax, int blank, const float* lse, const float* lp_ext, float* alpha,
                        const float* nll, const int32_t* tgt_len, const float* gout, void* grad, int grad_dtype, int64_t ldg,
                        float* alpha2);

/* Label-smoothed cross entropy (loss.py:5-31).  logits f32 [N,V] (row stride ldl), targets int64 [N] (0 = pad).
 * row_loss f32 [N] (0 on pad rows), lse f32 [N];  asr_ce_mean: loss[0] = sum(row_loss) / n_word, loss[1] = n_word
 * = count(target != 0) (f32 [2]);  asr_ce_loss_bwd: grad of gout * loss wrt logits, n_word = &loss[1]; grad f32 or bf16
 * (`grad_dtype`) with row stride ldg - as bf16 the whole row is written, zeros in columns V .. ldg-1 (see asr_ctc_loss_bwd).
 */
int asr_ce_loss_fwd(void* stream, const float* logits, int64_t ldl, const int64_t* targets, int N, int V,
                    float smoothing, float* row_loss, float* lse);
int asr_ce_mean(void* stream, const float* row_loss, const int64_t* targets, int N, float* loss);
int asr_ce_loss_bwd(void* stream, const float* logits, int64_t ldl, const int64_t* targets, int N, int V,
                    float smoothing, const float* lse, const float* n_word, const float* gout, void* grad, int grad_dtype, int64_t ldg);
/* mask_lm's cal_ce_mask_loss (src/mask_lm/loss.py:5-32): loss[0] = sum over non-pad rows of row_loss / n_word with
 * loss[1] = n_word = count(target != 0 AND counted != 0) - the reference sums every non-pad row but counts only the masked ones.
 * counted: one byte per row.  The backward is asr_ce_loss_bwd with n_word = &loss[1]. */
int asr_ce_mean_masked(void* stream, const float* row_loss, const int64_t* targets, const unsigned char* counted, int N, float* loss);
/* mask_lm token masking (src/mask_lm/Mask_LM.py:19-41): ids int64 [B,T], rand01 f32 [B,T] (the reference's torch.rand((B,T))):
 * position t is kept iff rand01[b,(t+j) mod T] > p for every j in 0..M; out = kept ? ids : 0, masked (one byte) = !kept. */
int asr_token_mask(void* stream, const int64_t* ids, const float* rand01, int B, int T, float p, int M, int64_t* out,
                   unsigned char* masked);

/* ------------------------------------------------------------------------------------------------------------
 * CIF (cif_model.py:57-106).  asr_cif_scan_fwd runs the integrate-and-fire recurrence in the reference's exact fp32
 * operation order (one wavefront per utterance): cur/rem f32 [B,L] weights, fire_idx int32 [B,L] (frame index of
 * each fire, in order), n_fire int32 [B], n_label int32 [B] = round(sum alpha) (cif_model.py:95), tok optional (see backward).
 * asr_cif_gather_fwd forms out[b,u,:] (f32 [B,Umax,H], zero-padded rows u >= n_fire[b]) as the same ordered
 * fp32 sum the reference accumulates.
 */
int asr_cif_scan_fwd(void* stream, const float* alpha, int B, int L, float threshold, float* cur, float* rem,
                     int32_t* fire_idx, int32_t* n_fire, int32_t* n_label, int32_t* tok);
/* Backward of the two CIF kernels.  tok (int32 [B,L], optional output of asr_cif_scan_fwd): index of the token frame t accumulates
 * into, bit 30 set when the frame fires.  gather_bwd: d_out f32 [B,Umax,H] -> d_hidden [B,L,H], d_cur / d_rem [B,L];
 * scan_bwd: d_cur / d_rem -> d_alpha [B,L] (the autograd of cif_model.py:67-87 through the running `integrate`). */
int asr_cif_gather_bwd(void* stream, const float* hidden, const float* cur, const float* rem, const int32_t* tok,
                       const int32_t* n_fire, const float* d_out, int B, int L, int H, int Umax, float* d_hidden, float* d_cur,
                       float* d_rem);
int asr_cif_scan_bwd(void* stream, const float* d_cur, const float* d_rem, const int32_t* tok, int B, int L, float* d_alpha);
int asr_cif_gather_fwd(void* stream, const float* hidden, const float* cur, const float* rem, const int32_t* fire_idx,
                       const int32_t* n_fire, int B, int L, int H, int Umax, float* out);

/* Attention_Assigner tail (attentionAssigner.py:37-40): alpha = sigmoid(x . w + b) * (t < len). x f32 [B,L,Dh]. */
int asr_assigner_tail_fwd(void* stream, const float* x, const float* w, const float* b, const int32_t* len,
                          int B, int L, int Dh, float* alpha);
/* Conv1d k=w stride 1 valid + ReLU over time with implicit zero right-pad (conv_encoder.py:33-43) is expressed by
 * the caller as w_context shifted GEMMs through asr_gemm_nt; no dedicated entry point. */

/* ---- input step in front of the path ----------------------------------------------------------------------
 * asr_lfr_stack: low-frame-rate stacking of a padded batch (utils/data.py:191-218 build_LFR_features): x f32 [B,T,D], len int32 [B]
 * -> y f32 [B, ceil(T/n), m*D] with y[b,i,j*D:(j+1)*D] = x[b, min(i*n + j, len_b - 1)] for i < ceil(len_b / n) and zeros beyond;
 * len_out[b] = ceil(len_b / n).
 * asr_spec_aug: utils/utils.py:168-194 in place on x f32 [B,T,V].  rand01: the uniform [0,1) draws of the reference's loop in its
 * order, f32 [(n_freq + n_time) * 2, B]: per mask rand(B) for the width, rand(B) for the start; width = (long)(max_width * r),
 * start = (long)((extent - width) * r) with extent = V (frequency) or len_b (time).  A frequency band takes each frame's mean over
 * frequency, a time span the utterance's mean over time (sum over the padded T rows / len_b), both of the UNmasked features; time
 * masks are written last.  (The reference's frequency loop runs `time_mask_num` times - utils.py:177 - so its callers pass
 * n_freq = time_mask_num.)  fmean f32 [B,T], tsum f32 [B,V]: workspaces. */
int asr_lfr_stack(void* stream, const float* x, const int32_t* len, int B, int T, int D, int m, int n, float* y, int32_t* len_out);
int asr_spec_aug(void* stream, float* x, const int32_t* len, int B, int T, int V, const float* rand01, int n_freq, int freq_width,
                 int n_time, int time_width, float* fmean, float* tsum);

/* ---- greedy decoding ---------------------------------------------------------------------------------------
 * asr_argmax_rows: out[m] = argmax_v x[m, v] (ties -> lowest index, like torch.argmax / torch.max on CPU): decoder.py:151
 * (`torch.argmax(cur_score, -1)`), ctc_infer.py:77 (`torch.max(prob_tensor, 2)`).  x f32 [M, V] with row stride ld.
 * asr_log_softmax_rows: y = x - logsumexp(x) per row (decoder.py:118 `F.log_softmax(logits, -1)`).
 * asr_ctc_greedy_reduce: the collapse of ctc_infer.py:37-46 - of the first len[b] frame labels keep those that are not `blank` and
 * differ from the previous frame's label; out int64 [B, L] zero-padded, out_len int32 [B]. */
int asr_argmax_rows(void* stream, const float* x, int64_t ld, int M, int V, int64_t* out);
int asr_log_softmax_rows(void* stream, const float* x, int64_t ldx, int M, int V, float* y, int64_t ldy);
int asr_ctc_greedy_reduce(void* stream, const int64_t* frames, const int32_t* len, int B, int L, int blank, int64_t* out, int32_t* out_len);
/* Beam search (decoder.py:166-234).  asr_topk_rows: torch.topk(x, k, sorted=True) over the rows of x f32 [M, V] (row stride ld):
 * vals f32 [M, k], idx int64 [M, k], equal values in index order.  asr_beam_prune: the pruning of decoder.py:196-209 - per
 * utterance the best `beam` of the beam * beam candidates scores[parent] + next_scores[parent][j] -> their scores, parent rows
 * (k_indices // beam_size as global row ids into [B * beam]) and tokens next_preds[parent][j]; beam * beam <= 64. */
int asr_topk_rows(void* stream, const float* x, int64_t ld, int M, int V, int k, float* vals, int64_t* idx);
int asr_beam_prune(void* stream, const float* scores, const float* next_scores, const int64_t* next_preds, int B, int beam,
                   float* new_scores, int64_t* parent, int64_t* new_tok);
/* asr_lsm_topk_rows: asr_topk_rows of log_softmax(x) without materialising it (decoder.py:418-440, F.log_softmax then torch.topk);
 * twice != 0: of log_softmax(log_softmax(x)) (Decoder.batch_beam_decode re-normalises Decoder.step's log-probabilities,
 * decoder.py:118 + :191).  Same order as the separate calls, values to f32 rounding; V <= 4608 (ASR_ERR_UNSUPPORTED beyond). */
int asr_lsm_topk_rows(void* stream, const float* x, int64_t ld, int M, int V, int k, int twice, float* vals, int64_t* idx);
/* The per-token step of Decoder.batch_decode (decoder.py:138-164) with its position in device memory, so that the whole step is
 * one capturable launch sequence (hipGraph replay per token).  state int32[2]: [0] = t, the position of the token being fed
 * (0 = <sos>), [1] = the number of steps after which every row had produced <eos>, -1 until then.
 * asr_decode_embed:   x[b] = emb[cur[b]] + pe[t]                       (decoder.py:104-105, the one new position)
 * asr_kv_cache_put:   k_cache[b, h, t] = k_new[b, h], same for v       (caches [B*h, Tmax, 64], dtype ASR_F32 | ASR_BF16)
 * asr_decode_advance: preds[b, t+1] = cur[b]; finished[b] |= cur[b] == eos; len_decoded[b] += !finished[b]; k_len[b] = t + 2;
 *                     t += 1; state[1] = t once all rows are finished (and nothing changes after that)   (decoder.py:151-158) */
int asr_decode_embed(void* stream, const int64_t* cur, const float* emb, const float* pe, const int32_t* state, float* y32, void* y16,
                     int B, int D, int V, int max_pos);
int asr_kv_cache_put(void* stream, const void* k_new, const void* v_new, void* k_cache, void* v_cache, const int32_t* state, int BH,
                     int Tmax, int dtype);
int asr_decode_advance(void* stream, const int64_t* cur, int64_t* preds, int32_t* state, int32_t* k_len, unsigned char* finished,
                       int64_t* len_decoded, int eos, int B, int Tp1);
/* Whole sub-layers of the per-token decode step in one launch each (decode_blocks.hip): the step's M = B or B * beam rows form
 * row blocks whose workgroups split a sub-layer by hidden units / heads, add partial output rows with float atomics into a
 * zeroed accumulator and let the last arrival apply bias + residual + LayerNorm (and re-zero accumulator and counter).
 * d_model = 256, bf16 operands.  `workspace`: asr_decode_block_workspace_bytes(M) bytes, zeroed ONCE by the caller, reusable by
 * every call on the same stream.
 * asr_decode_ffn:       y = LayerNorm(relu(x W1^T + b1) W2^T + b2 + x)            PositionwiseFeedForward, module.py:48-53;
 *                       with pre_Wo: x = LayerNorm0(ctx pre_Wo^T + pre_bo + res) first (the end of the attention sub-layer in front,
 *                       attention.py:58-60), x16 then holds the attention output ctx [M, 256] and x32 that sub-layer's input res
 * asr_decode_self_attn: MultiheadAttention.forward (attention.py:33-62) for ONE new position per row against that row's K / V cache
 *                       [M, h, Tmax, 64] bf16: q / k / v projections (Wqkv rows [q | k | v] x (h * 64)), k and v written into slot
 *                       t = state[0], softmax(q . K[0..t] / 8) V, output projection + bias + residual + LayerNorm; with next_Wq
 *                       also the NEXT sub-layer's query projection of the normalised rows, next_q = (y next_Wq^T + next_bq) *
 *                       next_scale as bf16 head-major [M / next_Lq, 4, next_Lq, 64] (the decoder's cross attention, whose
 *                       queries are the beams of an utterance: what asr_proj_heads would write) */
int64_t asr_decode_block_workspace_bytes(int M);
int asr_decode_ffn(void* stream, const void* x16, const float* x32, const void* W1, const float* b1, const void* W2, const float* b2,
                   const float* gamma, const float* beta, void* workspace, float* y32, void* y16, int M, int d_model, int d_ff, float eps,
                   const void* pre_Wo, const float* pre_bo, const float* pre_gamma, const float* pre_beta, float pre_eps);
int asr_decode_self_attn(void* stream, const void* x16, const float* x32, const void* Wqkv, const float* bqkv, const void* Wo, const float* bo,
                         const float* gamma, const float* beta, void* k_cache, void* v_cache, const int32_t* state, void* workspace, float* y32,
                         void* y16, int M, int d_model, int h, int Tmax, float eps, const void* next_Wq, const float* next_bq, void* next_q,
                         int next_Lq, float next_scale);
/* Beam search over integrated frames for B utterances at once (Decoder_CIF.recognize_beam, decoder.py:425-475, which decodes ONE
 * utterance with a Python loop over hypotheses; Decoder_CIF.step_forward_cache, decoder.py:477-496).  Hypothesis rows r = b * beam + j;
 * the step position t is state[0]; utterance b is live while t < n_steps[b] (its number of integrated frames).
 * asr_beam_cat_frames:    out[r] = [frames[b, t, :D] | other] f32 [N, D + D2]; other = other32[r, :D2] or, when `cur` is given,
 *                         emb[cur[r]] + pe[t] (D2 = row length of emb): decoder.py:407-408 (input_affine's rows) and :416 (tgt_word_prj's)
 * asr_beam_step:          decoder.py:445-462 for every live utterance, in place: the best `beam` of the beam * beam candidates
 *                         scores[row j] + next_scores[row j][k] (ties in candidate order = Python's stable sort) -> scores, token rows
 *                         preds[r, 0..t] gathered from the parents + the new token at column t + 1 (preds int64 [N, W], W <= 512),
 *                         cur[r], parent[r] (global row ids); a finished utterance is left as it is and reports identity parents
 * asr_beam_reorder_cache: caches [n_kv, N, h, Tmax, 64] (ASR_F32 | ASR_BF16) re-gathered by parent row in place, positions <= t
 * asr_beam_advance:       t += 1; k_len[r] += 1; with `finished` (Decoder.batch_beam_decode, decoder.py:211-216) also finished[r] |=
 *                         cur[r] == eos, len_decoded[r] += !finished[r], and state[1] = steps taken once every row is finished - from
 *                         then on asr_beam_step and asr_beam_advance change nothing (state[1] must be -1 while the search runs) */
int asr_beam_cat_frames(void* stream, const float* frames, const int32_t* state, const float* other32, const int64_t* cur, const float* emb,
                        const float* pe, float* out, int N, int beam, int Tmax, int D, int D2, int V, int max_pos);
int asr_beam_step(void* stream, float* scores, const float* next_scores, const int64_t* next_preds, int64_t* preds, const int32_t* state,
                  const int32_t* n_steps, int64_t* parent, int64_t* cur, int B, int beam, int W);
int asr_beam_reorder_cache(void* stream, void* cache, const int64_t* parent, const int32_t* state, int n_kv, int B, int beam, int h, int Tmax,
                           int dtype);
int asr_beam_advance(void* stream, int32_t* state, int32_t* k_len, int N, const int64_t* cur, int eos, unsigned char* finished,
                     int64_t* len_decoded);

/* ---- CIF family, training side (autograd of cif_model.py:44-48, attentionAssigner.py:37-40, conv_encoder.py:33-49) and the tape's
 * gradient bookkeeping; all f32. */
/* dst[r, 0..cols) += src[r, 0..cols) for r < rows (row strides ldd / lds in elements): gradient accumulation where two paths join. */
int asr_add2d(void* stream, float* dst, int64_t ldd, const float* src, int64_t lds, int rows, int cols);
/* dst[o][a][b] (contiguous [O, A, B]) += src[o * lds + b * A + a]: a weight gradient the GEMM produced in [O, B, A] order
 * (conv kernels: nn.Conv1d / nn.Conv2d weights are [out, in, taps], the conv-as-GEMM weight matrix is [out, taps, in]). */
int asr_add_transposed(void* stream, float* dst, const float* src, int O, int A, int B, int64_t lds);
/* out = d * (y > 0): ReLU backward from the saved forward output y (f32 or bf16, `y_dtype`); out may alias d. */
int asr_relu_mask_mul(void* stream, const float* d, const void* y, int y_dtype, float* out, int64_t n);
/* Conv1d-as-GEMM input gradient: d_win f32 [rows, w * cin] is the gradient wrt the overlapping row windows the forward GEMM read
 * (window r = input rows r .. r+w-1); d_in f32 [rows + w, cin] receives d_in[r] = sum_j d_win[r - j][j * cin ..]. */
int asr_conv1d_overlap_add(void* stream, const float* d_win, int rows, int w, int cin, float* d_in);
/* Backward of asr_assigner_tail_fwd: g = d(loss)/d(alpha) [B*L], alpha its output, h the [B*L, Dh] input, w [Dh]:
 * d_h = dz * w with dz = g * alpha * (1 - alpha); dw [Dh] += dz^T h; db [1] += sum dz (caller-zeroed or accumulating). */
int asr_assigner_tail_bwd(void* stream, const float* g, const float* alpha, const float* h, const float* w, int B, int L, int Dh,
                          float* d_h, float* dw, float* db);
/* Quantity scaling (cif_model.py:44-48): num_pred[b] = sum_t alpha_raw[b,t], num[b] = count(targets[b,:] > 0),
 * scale[b] = (num + noise - 0.5) / num_pred, alpha = alpha_raw * scale.  noise f32 [B] (the reference draws torch.rand(B)).
 * bwd: d_raw = d_alpha * scale + (d_num_in - sum_t(d_alpha * alpha_raw) * scale / num_pred); d_num_in (optional, [B]) is the
 * quantity loss's gradient wrt num_pred. */
int asr_cif_rescale_fwd(void* stream, const float* alpha_raw, const int64_t* targets, const float* noise, int B, int L, int U,
                        float* alpha, float* num_pred, float* num, float* scale);
int asr_cif_rescale_bwd(void* stream, const float* d_alpha, const float* alpha_raw, const float* scale, const float* num_pred,
                        const float* d_num_in, int B, int L, float* d_raw);

/* Utility: dtype cast f32 -> bf16 (weights / activations entering the bf16 MFMA path). n elements. */
int asr_cast_f32_bf16(void* stream, const float* x, void* y, int64_t n);
/* logits *= (t < len)  (ctcModel/decoder.py:33-36), in place, f32 [B,L,V] with row stride ld (elements) and batch stride L*ld. */
int asr_mask_rows(void* stream, float* x, const int32_t* len, int B, int L, int V, int64_t ld);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
