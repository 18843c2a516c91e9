// Internal launcher interface between the engine (ev_engine.cpp) and the gfx950 kernels.
// Not part of the public ABI (that is include/evhip.h); the ev_op_* C wrappers in
// include/evhip_ops.h expose these launchers to the per-kernel parity tests.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace ev {

// A/B switches of the kernel launchers (EV_GEMM_TILE, EV_ATTN_X3_NW, ...: tools/bench_*.py) exist only in tuning builds
// (`build.py --variant tune EV_TUNING`, loaded through EVHIP_LIB): the product library reads no environment variable; what an
// integrator may choose is an ev_config field.
#ifdef EV_TUNING
inline const char* tuning_env(const char* name) { return getenv(name); }
#else
inline const char* tuning_env(const char*) { return nullptr; }
#endif

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_LRELU = 3, ACT_TANH = 4 };
enum DType { DT_F16 = 0, DT_F32 = 1, DT_F32S = 2, DT_MX = 3 };   // F32S: fp32 activations, fp16 hi/lo split weights (ConvGemmParams::W / W_lo)
                                                                  // MX: fp32 activations; hi.hi as one fp16 MFMA + the two cross terms as block-scaled fp4 MFMAs (W, W_lo, W_mx)

// out[m, n] = post( scale * ( act( sum_{tap,k} pro(A[m + (tap-center)*dil, k]) * W[n][tap][k] + bias[n] )
//                              + seq_bias[row_seq[m]][n] + res[m, n] ) + acc32[m, n] + add16_a[m, n] + add16_b[m, n] )
// rows with row_valid[m >> valid_shift] == 0 are written as exact zeros.
// Channels-last activations ([rows][channels]); the conv over time is an implicit GEMM whose
// M dimension is time.  Reference ops covered: nn.Linear, Conv1d (any k / dilation, "same" pad),
// ConvTranspose1d(k = 2*stride, pad = stride/2) as a 3-tap conv with N = stride*C_out.
struct ConvGemmParams {
    int dtype;              // DT_F16: A/W fp16, MFMA f32_16x16x32_f16; DT_F32: A/W fp32, MFMA f32_16x16x4_f32 (exact fp32)
    const void* A; int lda; // activations, row pitch in elements; rows [-64, M+64) must be readable
    const void* W;          // [N][taps][K], K contiguous (DT_F32S: fp16 "hi" part)
    const void* W_lo;       // DT_F32S only: fp16((w - hi) * 2048), same layout
    const float* bias;      // [N] or null
    int M, N, K, taps, dil, center;
    const uint8_t* row_valid; int valid_shift;   // null -> all rows valid
    const int32_t* row_seq; const float* seq_bias; int ld_seq_bias;   // per-utterance additive vector, or null
    int act; float act_slope;
    int pro_lrelu; float pro_slope;              // leaky-relu applied to A while staging
    const void* res; int res_dtype; int ldres;   // residual (fp16 or fp32) or null
    float out_scale;
    const float* acc32; int ldacc;               // fp32 accumulate-in (after scaling) or null
    int post_lrelu; float post_slope;
    void* out16; float* out32; int ldo;          // either / both outputs
    int out32_before_post;                       // out32 receives the value BEFORE post_lrelu (stage taps)
    int reserved0;                               // (was: ablation switches used while tuning)
    const void* add16_a; const void* add16_b; int ldadd;   // two fp16 tensors added after scaling (the MRF sum of three ResBlocks
                                                 // with the first two branches kept in fp16: half the traffic of acc32), or null
    int ksplit;                                  // DT_F32S, N % 64 == 0, no add16: > 1 = split-K.  The K / 32 chunks are cut into `ksplit` equal ranges (K / 32 must be a
                                                 // multiple), each (tile, range) is a block of the 128 x 64-tile kernel writing fp32 partial sums to mx_scratch
                                                 // (>= ksplit * M * N * 4 bytes), and a second kernel adds them in range order and applies the epilogue.  Shortens the
                                                 // sequential step chain of the token-rate GEMMs (few tiles, K * taps up to 4608).  The summation order differs from
                                                 // ksplit <= 1, so a caller that promises batch invariance picks it by layer shape, never by M.
    // DT_MX only.  W_mx: the weight's fp4 planes (emotivoice_amd/mxfp4.py: pack_weight_planes), null -> the call runs as DT_F32S.
    // Activations, either (a) A = fp32 [M][K] + mx_scratch (>= mx_scratch_bytes(M, K) bytes): the launcher first runs mx_planes_kernel
    // (leaky-relu of pro_lrelu, then the planes) into the scratch, or (b) a plane set written by the producer's epilogue (mxo below):
    // A = the fp16 hi plane (lda == K), mx_x4 / mx_xs / mx_xs_stride = its code and scale planes; pro_lrelu must be 0.
    // polyphase_cout > 0 (DT_MX, taps == 3, conv_gemm_mx_kernel launches): the call is a ConvTranspose1d(k = 2 s, stride s, pad s / 2) run as a 3-tap conv with
    // N = s * polyphase_cout (packer._convT_to_gemm): output phase n / polyphase_cout < s / 2 has an all-zero tap 2, the other phases an all-zero tap 0, and the kernel
    // skips that tap's matrix instructions (a third of the launch's).  0: every tap is multiplied.  The weights' zero tap must really be zero.
    const void* W_mx; void* mx_scratch; size_t mx_scratch_size;
    const void* mx_x4[2]; const void* mx_xs[2]; unsigned mx_xs_stride; int polyphase_cout;
    // Plane-set OUTPUT (any dtype whose launch takes an EPI_MXP epilogue: DT_MX kernels): besides / instead of out32 the epilogue writes
    // the planes of a = lrelu(result, mxo_slope) (slope 1 = none) viewed as [rows][C], C = 2^mxo_logC, or C = N when mxo_logC == 0
    // (ldo == N; a transposed conv's [M][s * C] output is the [M * s][C] tensor): mxo_h fp16(a); mxo_q4[0] / [1] fp4 codes of fp16(a) and of a - fp16(a), C / 2 bytes
    // per row; mxo_qs[0] / [1] their E8M0 block scales, one byte per 32 channels, chunk-major [C / 128][mxo_qs_stride / 4 rows][4].
    // Invalid rows give all-zero planes.  Every plane needs 64 readable slack rows on both sides for its consumer.
    void* mxo_h; void* mxo_q4[2]; void* mxo_qs[2]; unsigned mxo_qs_stride; int mxo_logC; float mxo_slope; int reserved3;
    // Residual from a plane set (res_dtype == DT_MX; DT_MX launches only): the residual x of the launch's [M][N] output is not an fp32 tensor but the
    // plane set of a = lrelu(x, 1 / res_inv_slope) that conv1 of the pair read as its operand: res = its fp16 hi plane ([M][N], ldres == N), res_x4 = the
    // fp4 codes of the remainder a - fp16(a) ([M][N / 2]), res_xs their E8M0 scales (chunk-major [N / 128][res_xs_stride / 4 rows][4], as mx_xs);
    // the epilogue rebuilds x = a' >= 0 ? a' : a' * res_inv_slope, a' = hi + code * scale.  tools/precision_study_mx.py: the generator's waveform error
    // goes from 3.4e-4 to 4.4e-4 on the zero-mean recipe, and a conv2 launch moves 8.7 instead of 14.1 bytes per element.
    const void* res_x4; const void* res_xs; unsigned res_xs_stride; float res_inv_slope;
    // Accumulate-in from a PARTIAL plane set (conv_gemm_mx_kernel launches with a plane-set residual; excludes acc32): the addend is acc_h (fp16 hi plane,
    // [M][N], ldacc == N) + acc_x4 (fp4 codes of the remainder, [M][N / 2]) * acc_xs (their E8M0 scales, chunk-major [N / 128][acc_xs_stride / 4 rows][4]) --
    // a plane set without the hi-code plane and without an activation: 2.53 bytes per element instead of acc32's 4.  mxo_partial != 0: the OUTPUT plane set is
    // such a partial one (mxo_h, mxo_q4[1], mxo_qs[1]; mxo_q4[0] / mxo_qs[0] unused; mxo_slope must be 1); it may alias the acc_* planes (same rows, same
    // columns: each element is read before it is rewritten by the same lane).  The MRF running sum of a generator stage travels like this.
    const void* acc_h; const void* acc_x4; const void* acc_xs; unsigned acc_xs_stride; int mxo_partial;
};
// bytes of activation-plane scratch a DT_MX call with an [M][K] input needs
size_t mx_scratch_bytes(int M, int K);
// 0 if the plane-set fields of a call are consistent with its dtype, shape and epilogue (launch_conv_gemm would run it)
int mx_check(const ConvGemmParams& p);
// 0 if ConvGemmParams::ksplit is consistent with the rest of the call (<= 1, or a DT_F32S call whose shape and scratch allow the split)
int splitk_check(const ConvGemmParams& p);
void launch_conv_gemm(const ConvGemmParams& p, hipStream_t s);
// Three independent DT_MX convs of the conv_gemm_mx_kernel family (same M, N, K and epilogue form, taps {3, 7, 11}, plane sets in) as ONE grid -- the same-level
// convs of a generator stage's three ResBlocks; 0 = launched (check_only: would be), -1 = not such a triple (launch them one by one: the results are the same bits)
int launch_conv_gemm_group3(const ConvGemmParams* ps, hipStream_t s, bool check_only = false);
// which kernel launch_conv_gemm runs a DT_MX call on: 0 = the split-precision fallback (three fp16 MFMAs per product), 1 = conv_gemm_mx_kernel,
// 2 = conv_c64_mx_kernel (profiling records name the launch by this, not by what the caller hoped for)
int mx_launch_kind(const ConvGemmParams& p);
// per-device setup of the kernels in ev_gemm.hip (large-LDS opt-in, CU count of the persistent kernels); 0 = OK
int init_device_kernels(int device);

// Fused HiFi-GAN ResBlock1 pair for C = 32 (reference models/hifigan/models.py:50-57, one iteration of the zip loop):
//     xt = leaky_relu(c1(leaky_relu(x, .1)) + b1, .1);   out = epilogue(c2(xt) + b2 + x)
// c1 = Conv1d(C, C, k, dilation d), c2 = Conv1d(C, C, k, dilation 1).  Persistent blocks, both weight sets stationary in LDS,
// the intermediate xt never leaves LDS.  `epi` carries the c2 epilogue in ConvGemmParams form (bias = b2, res = x,
// out_scale / acc32 / post_lrelu / out16 / out32 / row_valid as for launch_conv_gemm); its GEMM fields are ignored.
struct ResPairParams {
    const void* x; int ldx;             // [rows][32] fp16 residual stream; rows [-64, M+64) readable
    const void* w1; const float* b1;    // [32][k][32] fp16
    const void* w2;
    int M, k, dil;
    int gmin, gmax;                     // rows g of x (relative to the x pointer) that exist in the tensor: gmin <= g < gmax.  {0, M} for a
                                        // whole tensor; a launch on a row sub-range (chunked execution) passes the tensor's real bounds so
                                        // that only true sequence edges are zero-padded.  gmax == 0 means {0, M}.
    const void* w1_mx; const void* w2_mx;   // launch_resblock_pair_c32_mx only: the convs' fp4 planes (mxfp4.py: pack_pair_weight_planes)
    ConvGemmParams epi;
};
void launch_resblock_pair_c32(const ResPairParams& p, hipStream_t s);
// the same pair in the MX arithmetic (ev_pair_mx.h): x / epi.res fp32 [rows][32], w1 / w2 the fp16 hi parts, w1_mx / w2_mx the fp4 planes;
// epilogue: bias, fp32 residual (= x), out_scale, optional acc32, row mask, out32 only.  Returns 0, or -1 for an unsupported call.
int launch_resblock_pair_c32_mx(const ResPairParams& p, hipStream_t s);
// the pair at C = 64, k = 3 in the MX arithmetic, plane sets in / out (ev_pair64_mx.h): x = the fp16 hi plane of the input plane set (ldx == 64), epi.mx_x4 /
// mx_xs its code / scale planes, w1 / w2 fp16 hi parts, w1_mx / w2_mx = mxfp4.pack_c64_weight_planes; the residual is rebuilt from the input plane set
// (epi.res_inv_slope); outputs epi.out32 and / or the plane set epi.mxo_*.  Returns 0, or -1 for an unsupported call.
int launch_resblock_pair_c64_mx(const ResPairParams& p, hipStream_t s);
// same pair at C = 64 (stage 2), k = 3 only: both weight sets (48 KB) stationary in LDS, x as two 64-byte K-chunk planes
void launch_resblock_pair_c64(const ResPairParams& p, hipStream_t s);

// LayerNorm over the channel dim (eps 1e-12, reference modules/encoder.py:112-127), fp32 in.
// out16/out32 optional; if dot_w != null additionally dot_out[r] = <LN(x[r]), dot_w> + dot_b.
struct LayerNormParams {
    const float* x; int ldx; int rows; int C;
    const float* gamma; const float* beta; float eps;
    const uint8_t* row_valid;
    void* out16; float* out32; int ldo;
    const float* dot_w; float dot_b; float* dot_out;
    // optional (C <= 512): the MX plane set of the output as a DT_MX consumer reads it (ConvGemmParams::mx_x4 ...): fp16 hi plane [rows][C], fp4 code
    // planes of the hi / lo parts [rows][C / 2], their E8M0 scale planes [C / 128][mxo_qs_stride / 4 rows][4]; mxo_h == null: none
    void* mxo_h; void* mxo_q4[2]; void* mxo_qs[2]; unsigned mxo_qs_stride;
};
// where a DT_MX call with an fp32 [M][K] input lays that input's plane set inside its mx_scratch (what launch_conv_gemm's own mx_planes_kernel pass fills):
// a producer that writes these planes itself passes them as ConvGemmParams::A (= h, lda = K) / mx_x4 / mx_xs / mx_xs_stride instead of the fp32 tensor
struct MxScratchPlanes { void* h; void* q4[2]; void* qs[2]; unsigned qs_stride; };
MxScratchPlanes mx_scratch_planes(void* scratch, int M, int K);
void launch_layernorm(const LayerNormParams& p, hipStream_t s);

// token embedding gather + alpha * PE[pos] (reference model_open_source.py:107, encoder.py:257-261)
void launch_embed_pe(const int64_t* ling_packed, const int32_t* cu_seqlens_dev, const int32_t* row_seq /* -1 = gap */,
                     const int32_t* row_pos, const float* emb, int n_vocab, const float* pe, float alpha, float* out, float* tap_out,
                     int rows, int C, hipStream_t s);

// Self-attention, one (utterance, head, 64-query tile) per wave; fp32 math, online softmax.
// qkv: [rows][3*C] (fp16 or fp32) with q | k | v column blocks; out: [rows][C].
struct AttnParams {
    const void* qkv; int dtype; int ld; int C; int heads;
    const int32_t* seq_off; const int32_t* seq_len; int B; int max_len;
    void* out; int ldo;     // same dtype as qkv
};
void launch_attention(const AttnParams& p, hipStream_t s);

// SimBERT (BERT-base) embeddings: out[row] = word[ids] + type[type_ids] + pos[position] (transformers BertEmbeddings; the LayerNorm that
// follows is launch_layernorm); pooler: out[b] = tanh(W x[first row of text b] + bias) (BertPooler).  reference
// models/prompt_tts_modified/simbert.py:37,49-55.
void launch_bert_embed(const int64_t* ids, const int64_t* type_ids /* or null = 0 */, const int32_t* cu_seqlens_dev, const int32_t* row_seq,
                       const int32_t* row_pos, const float* word, const float* pos_emb, const float* type_emb, int vocab, int max_pos,
                       int n_types, float* out, int rows, int C, hipStream_t s);
void launch_bert_pooler(const float* x, int ldx, const int32_t* seq_off, const float* W, const float* bias, float* out, int B, int C,
                        hipStream_t s);

// u[b, :] = bias + Wspk . spk_emb[speaker[b]] + Wsty . style[b] + Wcon . content[b]
// (columns 384..2303 of embed_projection1, reference model_open_source.py:110-111)
void launch_cond_vector(const int64_t* speaker, const float* style, const float* content, const float* spk_emb, int n_speaker,
                        const float* Wcond /* [C][C + 2*bert] */, const float* bias, float* u, int B, int C, int bert,
                        hipStream_t s);

// x_var = x_proj + pitch_embed(pitch) + energy_embed(energy)   (Conv1d 1->C, k taps, zero pad; :131-134)
void launch_var_embed_add(const float* x, const float* pitch, const float* energy, const float* wp, const float* bp,
                          const float* we, const float* be, const uint8_t* row_valid, float* out, int rows, int C, int k,
                          hipStream_t s);

// durations: d = max(rint(exp(log_d) - 1), 0) (variance.py:47-51); all-zero guard, cumsum, centres
// (alignment.py:183-202).  One block per utterance, wavefront prefix sum.
void launch_durations(const float* log_d /* token rows */, const int32_t* tok_off, const int32_t* tok_len, int B,
                      float alpha, const int64_t* forced /* packed or null */, const int32_t* cu_seqlens_dev,
                      int64_t* dur_packed, float* logd_packed, float* centre_rows, int32_t* mel_len, hipStream_t s);

// Gaussian upsampling (alignment.py:204-210) + decoder positional encoding (encoder.py:257-261).
void launch_gauss_upsample(const float* xvar, const float* centre_rows, const int32_t* tok_off, const int32_t* tok_len,
                           const int32_t* frm_row_seq, const int32_t* frm_row_pos, const float* pe, float pe_alpha,
                           float delta, float* out, float* tap_out, int rows, int C, hipStream_t s);

// mel (B x (n_mels, T_b), fp32/fp16) -> channels-last fp16 [rows][ldo] with zero gaps and zero pad channels
void launch_mel_to_rows(const void* mel, int is_f16, const int64_t* mel_elem_off, const int32_t* frm_row_seq,
                        const int32_t* frm_row_pos, const int32_t* mel_len, void* out, int out_f32, int rows, int n_mels, int ldo,
                        hipStream_t s);

// conv_post: input [rows][C] -> Conv1d(C->1, k) -> tanh -> wav fp32.  fp16 input: already leaky-relu'd by its producer;
// fp32 input (split-precision mode): raw MRF mean, leaky_relu(pre_slope) applied while reading (pre_slope in [0, 1])
void launch_conv_post(const void* x, int is_f32, int ldx, const float* w /* [k][C] */, float bias, int k, float pre_slope,
                      const uint8_t* row_valid, int valid_shift, float* wav_rows, int rows, int C, hipStream_t s);

// gather valid rows of a [rows][ld] buffer (fp16 or fp32) into a packed fp32 [n_valid][C] host-visible buffer
void launch_pack_rows(const void* src, int dtype, int ld, int C, const int64_t* seq_row_off /* per utterance first row */,
                      const int64_t* seq_out_off /* per utterance packed offset (rows) */, const int32_t* seq_rows, int B,
                      int64_t max_rows, float* dst, hipStream_t s);

// rows [row0, row1) of the sinusoid table: pe[t][2i] = sin(t * div[i]), pe[t][2i+1] = cos(t * div[i]) (encoder.py:216-237,
// which auto-extends its table the same way for inputs longer than max_len)
void launch_pe_extend(float* pe, const float* div, int row0, int row1, int C, hipStream_t s);
void launch_wav_to_i16(const float* wav, int16_t* out, int64_t n, hipStream_t s);
// gap-layout row maps on the device: seq[r] = utterance of row r (-1 in gaps), pos[r] = position inside it, valid[r]
void launch_row_maps(const int32_t* off, const int32_t* len, int B, int32_t* seq, int32_t* pos, uint8_t* valid, int rows, hipStream_t s);
void launch_fill_zero(void* p, size_t bytes, hipStream_t s);

}  // namespace ev
