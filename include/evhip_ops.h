/*
 * evhip_ops.h -- per-kernel entry points of libevhip.so used by the parity tests (tests/test_gpu_ops.py).
 * They launch one gfx950 kernel on caller-provided DEVICE pointers and are not needed by an integrator;
 * the drop-in boundary is include/evhip.h.  Each op cites the reference op it is checked against.
 */
#ifndef EVHIP_OPS_H_
#define EVHIP_OPS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Implicit-GEMM conv / linear (emotivoice_amd/csrc/ev_gemm.hip).  Checked against torch.nn.functional
 * linear / conv1d / conv_transpose1d, i.e. the ops behind reference modules/encoder.py:50-52,72-109,
 * modules/variance.py:41-46 and models/hifigan/models.py:50-57,116-128.
 *   out[m,n] = post( scale * ( act( sum_{tap,k} pro(A[m+(tap-center)*dil,k]) * W[n][tap][k] + bias[n] )
 *                               + seq_bias[row_seq[m]][n] + res[m,n] ) + acc32[m,n] + add16_a[m,n] + add16_b[m,n] ), invalid rows -> 0
 * Field order and types mirror ev::ConvGemmParams exactly. */
typedef struct ev_conv_gemm_desc {
    int dtype;                 /* 0: fp16 operands (MFMA 16x16x32 f16), 1: fp32 operands (MFMA 16x16x4 f32),
                                  2: fp32 activations x fp16 hi/lo split weights, 3 fp16 MFMAs per product (fp32-level accuracy),
                                  3: fp32 activations; hi x hi as one fp16 MFMA + the two cross terms as block-scaled fp4 MFMAs
                                     (v_mfma_scale_f32_16x16x128_f8f6f4): needs W, W_lo, W_mx, mx_scratch */
    const void* A; int lda;
    const void* W;
    const void* W_lo;          /* dtype 2 only */
    const float* bias;
    int M, N, K, taps, dil, center;
    const uint8_t* row_valid; int valid_shift;
    const int32_t* row_seq; const float* seq_bias; int ld_seq_bias;
    int act; float act_slope;  /* 0 none, 1 relu, 2 gelu(erf), 3 leaky_relu, 4 tanh */
    int pro_lrelu; float pro_slope;
    const void* res; int res_dtype; int ldres;
    float out_scale;
    const float* acc32; int ldacc;
    int post_lrelu; float post_slope;
    void* out16; float* out32; int ldo;
    int out32_before_post;
    int reserved0;
    const void* add16_a; const void* add16_b; int ldadd;   /* two fp16 [M][N] tensors added after scaling (both or neither), or NULL */
    int ksplit;                                  // DT_F32S, N % 64 == 0, no add16: > 1 = split-K.  The K / 32 chunks are cut into `ksplit` equal ranges (K / 32 must be a
                                                 // multiple), each (tile, range) is a block of the 128 x 64-tile kernel writing fp32 partial sums to mx_scratch
                                                 // (>= ksplit * M * N * 4 bytes), and a second kernel adds them in range order and applies the epilogue.  Shortens the
                                                 // sequential step chain of the token-rate GEMMs (few tiles, K * taps up to 4608).  The summation order differs from
                                                 // ksplit <= 1, so a caller that promises batch invariance picks it by layer shape, never by M.
    const void* W_mx;          /* dtype 3: emotivoice_amd/mxfp4.py pack_weight_planes(W) on the device (NULL: the call runs as dtype 2) */
    void* mx_scratch; size_t mx_scratch_size;   /* dtype 3 with fp32 A: device scratch, >= ev_op_mx_scratch_bytes(M, K) */
    const void* mx_x4[2]; const void* mx_xs[2]; unsigned mx_xs_stride; int polyphase_cout;   /* dtype 3 with a plane-set input: A = its fp16 hi
                                  plane, these = its fp4 code planes (hi, lo), E8M0 scale planes and the chunk stride of those; else zero.
                                  polyphase_cout > 0 (dtype 3, taps 3): the call is a ConvTranspose1d(k = 2 s, stride s, pad s / 2) as a 3-tap conv with
                                  N = s * polyphase_cout (packer._convT_to_gemm): phases below s / 2 have an all-zero tap 2, the others an all-zero tap 0,
                                  and the MX conv-GEMM skips that tap's matrix instructions; 0 = every tap is multiplied */
    /* plane-set output (dtype 3): planes of lrelu(result, mxo_slope) as [rows][2^mxo_logC]: fp16 hi plane, fp4 code planes of the hi / lo
       parts, their scale planes [C/128][mxo_qs_stride/4][4]; all NULL = none.  emotivoice_amd/mxfp4.py states the contents. */
    void* mxo_h; void* mxo_q4[2]; void* mxo_qs[2]; unsigned mxo_qs_stride; int mxo_logC; float mxo_slope; int reserved3;
    /* res_dtype 3 (dtype 3 only): the residual is the plane set of lrelu(x, 1 / res_inv_slope): res = its fp16 hi plane [M][N] (ldres == N),
       res_x4 = the fp4 codes of the remainder [M][N / 2], res_xs their E8M0 scales [N / 128][res_xs_stride / 4][4]; x = hi + code * scale,
       negative values times res_inv_slope */
    const void* res_x4; const void* res_xs; unsigned res_xs_stride; float res_inv_slope;
    /* accumulate-in from a partial plane set (dtype 3 with res_dtype 3; instead of acc32): acc_h fp16 hi plane [M][N] (ldacc == N), acc_x4 fp4 codes of the
       remainder [M][N / 2], acc_xs their E8M0 scales [N / 128][acc_xs_stride / 4][4]; the addend is hi + code * scale.  mxo_partial != 0: the output plane
       set is partial as well (mxo_h, mxo_q4[1], mxo_qs[1] only; mxo_slope = 1), and may alias the acc_* planes */
    const void* acc_h; const void* acc_x4; const void* acc_xs; unsigned acc_xs_stride; int mxo_partial;
} ev_conv_gemm_desc;

int ev_op_conv_gemm(const ev_conv_gemm_desc* d, void* hip_stream);
/* scratch bytes a dtype-3 call with an [M][K] activation needs (fp16 hi plane, two fp4 code planes, two E8M0 scale planes) */
size_t ev_op_mx_scratch_bytes(int M, int K);

/* Fused HiFi-GAN ResBlock1 pair for C = 32: xt = lrelu(c1(lrelu(x)) + b1); out = epi(c2(xt) + b2 + x)
 * (reference models/hifigan/models.py:50-57).  `epi` uses the ev_conv_gemm_desc fields bias (= b2), res (= x), res_dtype,
 * ldres, row_valid, valid_shift, out_scale, acc32, ldacc, add16_a, add16_b, ldadd (acc32 and add16 are mutually exclusive here),
 * post_lrelu, post_slope (in [0, 1]), out16, out32, ldo, out32_before_post. */
typedef struct ev_res_pair_desc {
    const void* x; int ldx;
    const void* w1; const float* b1;
    const void* w2;
    int M, k, dil;
    int gmin, gmax;    /* rows of x that exist (relative to x): gmin <= g < gmax; leave both 0 for the whole tensor {0, M} */
    const void* w1_mx; const void* w2_mx;   /* ev_op_resblock_pair_c32_mx only: mxfp4.pack_pair_weight_planes(w) of the two convs */
    ev_conv_gemm_desc epi;
} ev_res_pair_desc;
int ev_op_resblock_pair_c32(const ev_res_pair_desc* d, void* hip_stream);
/* The same pair in the MX arithmetic (one fp16 MFMA + two block-scaled fp4 MFMAs per product): x and out32 fp32 [rows][32], w1 / w2 the
 * fp16 hi parts of the weights, w1_mx / w2_mx their fp4 planes; epi: bias, res (= x, fp32), row_valid, out_scale, acc32 (optional, may
 * alias out32), out32, ldo. */
int ev_op_resblock_pair_c32_mx(const ev_res_pair_desc* d, void* hip_stream);
/* The pair at C = 64, k = 3 in the MX arithmetic with plane sets in and out: x = the fp16 hi plane [rows][64] of the plane set of leaky_relu(x, 1 / res_inv_slope),
 * epi.mx_x4 / mx_xs / mx_xs_stride its code / scale planes; w1 / w2 fp16 hi parts [64][3][64], w1_mx / w2_mx = mxfp4.pack_c64_weight_planes; epi: bias (= b2),
 * res_inv_slope, out_scale, acc32 (optional), row_valid, out32 and / or the output plane set mxo_* (mxo_logC = 6).  Bit-identical to the two layer-wise
 * ev_op_conv_gemm launches it replaces. */
int ev_op_resblock_pair_c64_mx(const ev_res_pair_desc* d, void* hip_stream);
/* the same pair at C = 64 (HiFi-GAN stage 2), k = 3 only (both weight sets stay in LDS) */
int ev_op_resblock_pair_c64(const ev_res_pair_desc* d, void* hip_stream);

/* LayerNorm(eps) over channels, optional fused Linear(C,1) head (reference modules/encoder.py:112-127,
 * modules/variance.py:29-33,46). */
int ev_op_layernorm(const float* x, int rows, int C, const float* gamma, const float* beta, float eps,
                    const uint8_t* row_valid, void* out16, float* out32, const float* dot_w, float dot_b,
                    float* dot_out, void* hip_stream);

/* The same LayerNorm writing the MX plane set of its output (emotivoice_amd/mxfp4.py) instead of fp32 rows: h fp16 hi plane [rows][C], q4h / q4l the fp4
 * codes of the hi / lo parts [rows][C / 2], qsh / qsl their E8M0 block scales [C / 128][qs_stride / 4 rows][4]; C <= 512, C % 128 == 0.  What the mel
 * decoder's QKV projection / conv-FFN read in the mx mode (reference modules/encoder.py:154-200: the norm in front of each sub-layer). */
int ev_op_layernorm_planes(const float* x, int rows, int C, const float* gamma, const float* beta, float eps,
                           const uint8_t* row_valid, void* h, void* q4h, void* q4l, void* qsh, void* qsl,
                           unsigned qs_stride, void* hip_stream);

/* Multi-head self-attention restricted to each utterance's rows (reference modules/encoder.py:72-109).
 * is_f16: 1 = fp16 rows (fp16 MFMA flash kernel), 0 = fp32 rows, exact fp32 MFMA products, 2 = fp32 rows, split-precision products
 * (three fp16 MFMAs each: the mel decoder in the strict / mx modes). */
int ev_op_attention(const void* qkv, int is_f16, int C, int heads, const int32_t* seq_off, const int32_t* seq_len,
                    int B, int max_len, void* out, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* EVHIP_OPS_H_ */
