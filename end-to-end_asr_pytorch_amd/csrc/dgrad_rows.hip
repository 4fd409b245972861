// Data gradient of an nn.Linear whose INPUT is d_model = 256 wide, at encoder size, as a row-block kernel:
//   dX[M, 256] = dY[M, K] . W[K, 256]  (+ addend)          autograd of y = x W^T (attention.py:43-49 w_qs / w_ks / w_vs, :58 fc; the
//   decoder's cross-attention K / V projections over the encoder output, decoder.py:216-230), W as nn.Linear stores it: [out = K][in = 256]
// A tiled GEMM sees a skinny problem here (N = 256, K = 256 ... 3072): 500 tiles with a short reduction each, every one of them
// mostly prologue and epilogue - 29.5 us for K = 768 where the bytes (dY in, addend in, dX out: 115 MB) take 18.  This kernel is
// csrc/ffn.hip's data-gradient kernel without its first product: a workgroup owns 128 rows of dX complete (4 waves x 32 rows, the
// 256 x 32 accumulator of a wave = 128 registers), W streams through LDS in 64-row chunks exactly as W1 does there (row-major image,
// transposing reads, same swizzle), and the rows' dY chunk [128][64] rides in the same ring stage (16 KiB, 16-byte slot ^ (row & 7):
// the B operand of a lane is one conflict-free ds_read_b128).  Three stages, requested two chunks ahead, one barrier per chunk.
// Because the rows are complete, the epilogue can be more than a store:
//   OUT_F32   dX (+ addend) as f32 rows                         (the encoder-output gradient of the decoder's cross K / V: addend = what
//                                                                is there already)
//   OUT_BF16  dX as bf16 rows                                   (fc's gradient wrt the attention context: the operand of attention_bwd)
//   OUT_LNB   dX + addend never leaves: it is the dy of the LayerNorm that produced x (module.py:52 / encoder.py:48-50), and the
//             launch writes what asr_add_layernorm_bwd writes - ds, ds16, dgamma / dbeta / dbias column sums (same arithmetic per row as
//             that kernel and as asr_ffn_bwd_ln; unlike there the fold pays: this kernel is bound by its bytes, and the fold removes
//             65 MB of them - dx out, dy in - together with a launch)
#include "dgrad_rows_body.h"

namespace {

template <int OUT, bool ADD>
__global__ __launch_bounds__(256, 1) void dgrad_rows_kernel(const RowsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[RSMEM];
    dgrad_rows_body<OUT, ADD>(a, smem);
}

template <int OUT>
void launch_rows(hipStream_t s, const RowsArgs& a) {
    const dim3 grid((a.M + RBM - 1) / RBM), block(256);
    if (a.addend) hipLaunchKernelGGL((dgrad_rows_kernel<OUT, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((dgrad_rows_kernel<OUT, false>), grid, block, 0, s, a);
}

int check_common(const char* who, const void* dy, long long ldy, const void* w, int M, int K, int d_model) {
    ASR_REQUIRE(d_model == RD, ASR_ERR_UNSUPPORTED, "%s: d_model = %d (the row-block data gradient is built for 256)", who, d_model);
    ASR_REQUIRE(K >= RKC && K % RKC == 0, ASR_ERR_UNSUPPORTED, "%s: K = %d (a multiple of 64)", who, K);
    ASR_REQUIRE(M > 0 && ldy >= K && ldy % 8 == 0 && ((int64_t)(M - 1) * ldy + K) * 2 < (1ll << 32), ASR_ERR_ARG, "%s: bad M / row stride", who);
    ASR_REQUIRE(dy && w, ASR_ERR_ARG, "%s: null argument", who);
    ASR_REQUIRE(asr_aligned(dy, 16) && asr_aligned(w, 16), ASR_ERR_ALIGN, "%s: 16-byte aligned operands required", who);
    return 0;
}

}  // namespace

extern "C" int asr_dgrad_rows(void* stream, const void* dy, int64_t ldy, const void* w, const float* addend, void* out, int out_dtype, int M,
                              int K, int d_model) {
    if (int rc = check_common("asr_dgrad_rows", dy, ldy, w, M, K, d_model)) return rc;
    ASR_REQUIRE(out && (out_dtype == ASR_F32 || out_dtype == ASR_BF16), ASR_ERR_ARG, "asr_dgrad_rows: bad output");
    ASR_REQUIRE(asr_aligned(out, 16) && (!addend || asr_aligned(addend, 16)), ASR_ERR_ALIGN, "asr_dgrad_rows: 16-byte aligned buffers required");
    RowsArgs a{};
    a.dy = (const bf16_t*)dy; a.w = (const bf16_t*)w; a.addend = addend; a.M = M; a.K = K; a.ldy = ldy;
    if (out_dtype == ASR_F32) { a.out32 = (float*)out; launch_rows<OUT_F32>((hipStream_t)stream, a); }
    else { a.out16 = (bf16_t*)out; launch_rows<OUT_BF16>((hipStream_t)stream, a); }
    ASR_LAUNCH_CHECK("asr_dgrad_rows");
    return 0;
}

extern "C" int asr_dgrad_rows_ln(void* stream, const void* dy, int64_t ldy, const void* w, const float* addend, int B, int L, int K,
                                 int d_model, const float* ln_s, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                 const float* ln_beta, const int32_t* row_len, float* ds_out, void* ds16_out, float* dgamma, float* dbeta, float* dbias,
                                 asr_dropout_t drop_x) {
    ASR_REQUIRE(B > 0 && L > 0, ASR_ERR_ARG, "asr_dgrad_rows_ln: bad B / L");
    const int M = B * L;
    if (int rc = check_common("asr_dgrad_rows_ln", dy, ldy, w, M, K, d_model)) return rc;
    ASR_REQUIRE(ln_s && (ln_mean || ln_beta) && ln_rstd && ln_gamma && ds_out && ds16_out && dgamma && dbeta, ASR_ERR_ARG, "asr_dgrad_rows_ln: null argument");
    ASR_REQUIRE(asr_aligned(ln_s, 16) && asr_aligned(ln_gamma, 16) && asr_aligned(ds_out, 16) && asr_aligned(ds16_out, 16) &&
                    (!addend || asr_aligned(addend, 16)), ASR_ERR_ALIGN, "asr_dgrad_rows_ln: 16-byte aligned buffers required");
    RowsArgs a{};
    a.dy = (const bf16_t*)dy; a.w = (const bf16_t*)w; a.addend = addend; a.M = M; a.K = K; a.ldy = ldy;
    a.out32 = ds_out; a.out16 = (bf16_t*)ds16_out;
    a.ln_s = ln_s; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.row_len = row_len; a.L = L;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias; a.drop_x = drop_x;
    launch_rows<OUT_LNB>((hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_dgrad_rows_ln");
    return 0;
}
