/*
 * evhip.h -- C ABI of libevhip.so: the MI355X-native (gfx950) EmotiVoice inference hot path.
 *
 * The reference has no FFI/plugin interface for this path; its boundary is the Python object
 * protocol of JETSGenerator (reference models/prompt_tts_modified/jets.py:26-71).  Each entry
 * point below replaces one piece of that protocol:
 *
 *   ev_create            <- JETSGenerator.__init__(config) + .to(device)      (jets.py:27-47,
 *                           inference_am_vocoder_joint.py:70)
 *   ev_load_weights[_device] <- .load_state_dict(ckpt['generator'])           (inference_am_vocoder_joint.py:72-73)
 *   ev_synthesize        <- JETSGenerator.forward, inference branch           (jets.py:50-71 ->
 *                           model_open_source.py:102-163 -> models/hifigan/models.py:115-131)
 *   ev_vocoder           <- HiFiGANGenerator.forward on pre-computed mels     (models/hifigan/models.py:115-131)
 *   ev_get_stage         <- register_forward_hook taps used by the parity tests
 *   ev_last_error        <- Python exceptions (no exceptions cross the ABI)
 *
 * Conventions: 0 = OK, negative = error (message via ev_last_error).  A handle owns one device,
 * one HIP stream and one workspace arena; it is NOT thread-safe (the reference is single-threaded
 * per process as well).  Inputs are borrowed for the duration of a call.  Outputs are owned by
 * the handle and stay valid until the next ev_synthesize / ev_vocoder / ev_destroy on it.
 * No torch types appear anywhere in this interface: plain pointers and sizes only.
 */
#ifndef EVHIP_H_
#define EVHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EV_ABI_VERSION 7      /* 2: ev_config engine switches (mx_residual, decoder_attention, fused_pairs), ev_abi_info;
                                 3: ev_config.mx_mrf / decoder_ln_planes, partial plane sets in ev_conv_gemm_desc (acc_h ..., mxo_partial);
                                 4: ev_config.token_splitk, ev_conv_gemm_desc.ksplit (same struct sizes);
                                 5: ev_default_config() sets decoder_precision = vocoder_precision = EV_PREC_MX, the mode that meets the 1e-3 contract
                                    (same struct sizes; until 4 the default was EV_PREC_F16 = 2.4e-3 on zero-mean audio);
                                 6: ev_config.mx_act_format (was reserved[0], same struct size): the fused C = 32 pairs of the MX generator default to
                                    E5M2 activation operands in the cross terms -- the results of the default mode change in their last bits;
                                 7: ev_config.mx_group appended (sizeof(ev_config) + 4): grouped launches of a stage's same-level ResBlock convs -- same bits */

typedef struct ev_handle ev_handle;

/* Mirrors reference config/joint/config.yaml:36-94 (+ n_vocab/n_speaker patched in by the
 * callers, inference_am_vocoder_joint.py:57-58).  Use ev_default_config() and override. */
typedef struct ev_config {
    int32_t abi_version;        /* EV_ABI_VERSION */
    int32_t n_vocab;            /* 502   */
    int32_t n_speaker;          /* 2014  */
    int32_t n_mels;             /* 80    */
    int32_t hidden;             /* 384 (encoder/decoder/variance hidden) */
    int32_t heads;              /* 8     */
    int32_t enc_layers;         /* 4     */
    int32_t dec_layers;         /* 4     */
    int32_t ffn_kernel;         /* 3     */
    int32_t bert_dim;           /* 768   */
    int32_t dur_layers;         /* 2     */
    int32_t pitch_layers;       /* 3     */
    int32_t energy_layers;      /* 2     */
    int32_t var_kernel;         /* 3     */
    int32_t var_embed_kernel;   /* 9     */
    int32_t n_up;               /* 4     */
    int32_t up_rates[8];        /* 8,8,2,2 */
    int32_t up_kernels[8];      /* 16,16,4,4 */
    int32_t up_init_ch;         /* 512   */
    int32_t n_rb;               /* 3     */
    int32_t rb_kernels[8];      /* 3,7,11 */
    int32_t rb_dils[8][4];      /* {1,3,5} x3 */
    int32_t n_rb_dils;          /* 3     */
    int32_t sample_rate;        /* 16000 */
    /* engine options (not in the reference) */
    int32_t decoder_precision;  /* EV_PREC_MX (default since ABI 5: split precision with the conv-FFN / projections in the MX arithmetic, see
                                   vocoder_precision), EV_PREC_X3 (split precision), EV_PREC_F32 (exact fp32 MFMA) or EV_PREC_F16 (opt-in, out of
                                   the 1e-3 contract on zero-mean audio) */
    int32_t keep_stages;        /* !=0: keep every Appendix-C stage tap retrievable by ev_get_stage */
    int32_t token_rate_split;   /* 1 (default): fp32 token-rate GEMMs as 3 fp16 MFMAs on hi/lo splits (fp32-level accuracy,
                                   ~4x faster); 0: exact fp32 MFMA (v_mfma_f32_16x16x4_f32) */
    int32_t vocoder_chunk_mb;   /* EV_PREC_F16 generator only (accepted and without effect in the other precisions); > 0: the ResBlocks of a generator stage run on row chunks of about this many MB per fp16 tensor so
                                   that a chunk's intermediates stay in the 256 MB Infinity Cache (bit-identical results for any
                                   value; measured slower than whole tensors in the full forward, hence off); 0 (default): whole
                                   tensors */
    int32_t vocoder_streams;    /* 0 (default): the three ResBlocks of a generator stage run concurrently (two internal streams beside
                                   the handle's); 1: everything on the handle's stream.  The pitch / energy predictors use the same two streams beside the
                                   duration predictor. */
    int32_t vocoder_precision;  /* EV_PREC_MX (default since ABI 5): the X3 data flow, but a product is ONE fp16 MFMA (hi x hi) + two block-scaled
                                   fp4 MFMAs for the cross terms (v_mfma_scale_f32_16x16x128_f8f6f4) on operand planes written by the producing
                                   layer: waveform within ~5e-4 of the reference on every fixture, zero-mean audio included (contract: 1e-3);
                                   EV_PREC_X3: fp32 activations, every product as three fp16 MFMAs on hi/lo splits (fp32-class accuracy);
                                   EV_PREC_F16 (opt-in): fp16 operands / fp16 activations in HBM, fp32 accumulate -- 1.6x faster, but 2.4e-3
                                   on zero-mean audio, i.e. OUTSIDE the 1e-3 contract */
    /* engine switches that used to be environment variables (read per layer per forward); all default to 0 */
    int32_t mx_residual;        /* EV_PREC_MX generator: 0 (default) = the residual stream of a ResBlock travels ONLY as the plane set its
                                   conv1 reads (fp16 hi plane + fp4 remainder codes; conv2's epilogue rebuilds x from it: 8.7 instead of
                                   14.1 bytes per element and conv2 launch, worst fixture 5.3e-4 instead of 4.4e-4); 1 = a separate fp32
                                   residual tensor beside the planes (round 3's flow) */
    int32_t decoder_attention;  /* decoder self-attention in the X3 / MX modes: 0 (default) = split precision (three fp16 MFMAs per
                                   product, K / V tiles staged through LDS); 1 = exact fp32 MFMA kernel (the token-rate encoder's) */
    int32_t fused_pairs;        /* 0 (default) = fused ResBlock-pair kernels where they exist (C = 32 every k, C = 64 / k = 3 in fp16); 1 = every
                                   conv as its own launch (A/B switch, bit-identical in the fp16 mode) */
    int32_t mx_mrf;             /* EV_PREC_MX generator, stages with >= 128 channels: 0 (default) = the running MRF sum of a stage's three ResBlocks travels
                                   as a partial plane set (fp16 hi plane + fp4 remainder codes + block scales: 2.53 instead of 4 bytes per element and
                                   transfer, re-quantised once per ResBlock; emulated cost 5e-6 of waveform error); 1 = an fp32 running sum */
    int32_t decoder_ln_planes;  /* EV_PREC_MX decoder: 0 (default) = the LayerNorms in front of the QKV projection and the conv-FFN write the plane sets those
                                   layers read (no fp32 copy, no separate quantisation pass; the same bits); 1 = fp32 output + a planes pass */
    int32_t token_splitk;       /* token-rate stack with split hi/lo GEMMs: 0 (default) = the phoneme encoder's second conv-FFN conv (N = hidden, K x taps = 4608:
                                   144 sequential (K-chunk, tap) steps per tile) runs split-K -- 4 ranges, partial sums reduced in range order by a second
                                   kernel -- because that chain is what a single utterance waits for (B = 1, 64 phonemes: 4.25 -> 3.95 ms; +0.08 ms per
                                   32 x 256-token batch).  Chosen by layer shape only: an utterance alone and in a batch gets the same bits.
                                   1 = every GEMM in one pass (the summation order of rounds 1-3) */
    int32_t mx_act_format;      /* EV_PREC_MX generator, format of the ACTIVATION operand in the two cross-term MFMAs where a kernel offers the choice (the fused
                                   ResBlock pairs at 32 channels): 0 (default) = OCP E5M2 without block maxima -- Q(xh) = the top byte of the fp16 hi part,
                                   Q(xl) = E5M2 of the remainder at the constant block scale 2^-11; per-element exponents, ~3x fewer quantiser instructions,
                                   emulated waveform error 2 % LOWER than fp4's; 1 = block-scaled fp4 (e2m1) as in ABI <= 5.  Weights are fp4 planes either way. */
    int32_t mx_group;           /* EV_PREC_MX generator, stages with >= 128 channels, large batches (ABI 7): 0 (default) = the same-level convs of a stage's three
                                   ResBlocks (k = 3 / 7 / 11: independent until the MRF sum) are issued as ONE grouped launch per level -- a launch of its own
                                   costs each conv 30-50 us of ramp and tail at B = 32 x 1024 frames -- with one set of intermediates per ResBlock
                                   (+ ~5 GB of workspace at that size); 1 = one launch per conv, ResBlock after ResBlock.  The same bits either way. */
} ev_config;

/* Precision of the frame-rate path (ev_default_config: MX for both components).  F16: fp16 MFMA operands (what BASELINE.json's bf16 / fp16 configs name).
 * F32: exact fp32 MFMA (decoder only; v_mfma_f32_16x16x4_f32, 1/16 of the fp16 rate).
 * X3:  fp32 activations in HBM, weights and activations split into fp16 hi + lo parts, x*w = hi*hi + hi*lo + lo*hi as three
 *      fp16 MFMAs with fp32 accumulation (2^-22 relative truncation: the fp32 rounding class at 1/3 of the fp16 rate).
 *      With decoder_precision = vocoder_precision = EV_PREC_X3 ("strict") the waveform matches the fp32 reference to ~1e-5
 *      relative L2 also on DC-free audio, where fp16 operands measure ~2e-3. */
enum { EV_PREC_F16 = 0, EV_PREC_F32 = 1, EV_PREC_X3 = 2, EV_PREC_MX = 3 };

/* flags for ev_synthesize / ev_vocoder */
enum {
    EV_FLAG_DEVICE_INPUTS = 1,  /* all input pointers are device pointers (zero-copy from a torch-ROCm tensor) */
    EV_FLAG_NO_VOCODER = 2,     /* acoustic model only (mel out) */
    EV_FLAG_WANT_INT16 = 4,     /* also produce the caller epilogue wav*32768 -> int16 (inference_am_vocoder_joint.py:130-131) */
    EV_FLAG_FORCED_DURATIONS = 8 /* teacher-forced durations (test mode): use result-independent durations passed via ev_set_forced_durations */
};

/* Result of one call.  All pointers are DEVICE pointers owned by the handle.
 * Packed layouts: utterance b occupies [mel_offsets[b], mel_offsets[b]+mel_lens[b]) rows of mel and
 * 256x that range of wav; tokens are packed exactly like the `ling` input (cu_seqlens). */
typedef struct ev_result {
    int32_t batch;
    int32_t total_tokens;
    int64_t total_frames;           /* sum of mel_lens */
    int64_t total_samples;          /* total_frames * prod(up_rates) */
    const float*   wav;             /* (total_samples,) fp32 in [-1,1]        = wav_predictions   */
    const int16_t* wav_i16;         /* (total_samples,) or NULL                                   */
    const float*   mel;             /* (total_frames, n_mels) fp32 row-major  = dec_outputs       */
    const int64_t* durations;       /* (total_tokens,)  int64                 = log_duration_predictions (inference) */
    const float*   log_durations;   /* (total_tokens,)  fp32, pre-round (test tap)                */
    const float*   pitch;           /* (total_tokens,)  fp32                  = pitch_predictions  */
    const float*   energy;          /* (total_tokens,)  fp32                  = energy_predictions */
    const int32_t* mel_lens;        /* (batch,) HOST pointer                                      */
    const int64_t* mel_offsets;     /* (batch+1,) HOST pointer: exclusive prefix sum of mel_lens  */
} ev_result;

void ev_default_config(ev_config* cfg);

/* What the loaded library was built as: EV_ABI_VERSION and the sizes of the structs a binding mirrors (ev_config, ev_result, and the two
 * descriptors of include/evhip_ops.h), so that a stale binding or a stale libevhip.so is an error at load time instead of a mis-parsed struct.
 * sizes: 4 entries {sizeof(ev_config), sizeof(ev_result), sizeof(ev_conv_gemm_desc), sizeof(ev_res_pair_desc)}; returns EV_ABI_VERSION. */
int ev_abi_info(size_t sizes[4]);

int ev_create(int device_id, const ev_config* cfg, ev_handle** out);
void ev_destroy(ev_handle* h);
const char* ev_last_error(ev_handle* h);   /* h may be NULL: returns the last creation error */

/* Use an externally owned hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) instead of the
 * handle's own stream.  Pass NULL to go back to the internal stream. */
int ev_set_stream(ev_handle* h, void* hip_stream);

/* Packed weight blob produced by emotivoice_amd/packer.py (self-describing; manifest_json is an
 * optional human-readable copy of the table and may be NULL).  Host pointer: copied to the device.
 * Device pointer variant (after an RCCL broadcast): borrowed, must outlive the handle. */
int ev_load_weights(ev_handle* h, const void* blob, size_t nbytes, const char* manifest_json);
int ev_load_weights_device(ev_handle* h, const void* dptr, size_t nbytes, const char* manifest_json);

/* JETSGenerator.forward (inference): B utterances, tokens packed back to back.
 *   ling      (cu_seqlens[B],) int64 phoneme ids            = inputs_ling (unpadded)
 *   cu_seqlens(B+1,) int32 HOST pointer, cu_seqlens[0] = 0   = input_lengths as prefix sums
 *   speaker   (B,) int64                                     = inputs_speaker
 *   style     (B, bert_dim) fp32                             = inputs_style_embedding
 *   content   (B, bert_dim) fp32                             = inputs_content_embedding
 *   alpha     duration scale as GaussianUpsampling.forward applies it (alignment.py:183).  NB: the reference's inference branch
 *             never forwards JETSGenerator.forward's alpha to the length regulator (model_open_source.py:142), so the drop-in
 *             Python mirror always passes 1.0; values != 1 are an extension (speed control)
 * Every utterance is evaluated with the reference's B = 1 semantics (zero halo at sequence edges,
 * attention restricted to its own tokens / frames). */
int ev_synthesize(ev_handle* h, int B, const int64_t* ling, const int32_t* cu_seqlens,
                  const int64_t* speaker, const float* style, const float* content,
                  float alpha, uint32_t flags, ev_result* out);

/* Durations for EV_FLAG_FORCED_DURATIONS: (total_tokens,) int64 HOST pointer, copied. */
int ev_set_forced_durations(ev_handle* h, const int64_t* durations, int64_t n);

/* HiFi-GAN generator only.  mel: B tensors packed back to back, each (n_mels, mel_lens[b]) row-major
 * (the reference's (B,80,T) layout per utterance), fp32 or fp16 (mel_is_f16). mel_lens: HOST pointer. */
int ev_vocoder(ev_handle* h, int B, const void* mel, int mel_is_f16, const int32_t* mel_lens,
               uint32_t flags, ev_result* out);

/* SimBERT prompt / content encoder on the device (reference models/prompt_tts_modified/simbert.py:48-72, called twice per utterance
 * at inference_am_vocoder_joint.py:25-38,106-107 and predict.py:142-158 -- on the CPU there): BERT-base forward, the result is
 * outputs["pooled_output"] = tanh(W_pool h[CLS] + b_pool), which the callers pass as inputs_style_embedding /
 * inputs_content_embedding.  The weights are a second packed blob (emotivoice_amd/packer.py: pack_bert_state_dict, from the
 * StyleEncoder state dict); fp32-class arithmetic (split-precision GEMMs, exact-fp32 MFMA attention). */
typedef struct ev_bert_config {
    int32_t vocab_size;        /* 13685 (WangZeJun/simbert-base-chinese) */
    int32_t hidden;            /* 768  */
    int32_t layers;            /* 12   */
    int32_t heads;             /* 12 (64-wide heads) */
    int32_t intermediate;      /* 3072 */
    int32_t max_position;      /* 512  */
    int32_t type_vocab;        /* 2    */
    float   ln_eps;            /* 1e-12 */
    int32_t reserved[8];
} ev_bert_config;
void ev_default_bert_config(ev_bert_config* cfg);
int ev_style_load_weights(ev_handle* h, const ev_bert_config* cfg, const void* blob, size_t nbytes);   /* host pointer, copied */
/* B texts, token ids packed back to back (what the tokenizer returns per text: [CLS] ... [SEP]).
 *   input_ids      (cu_seqlens[B],) int64          = tokenizer(...)["input_ids"]
 *   token_type_ids (cu_seqlens[B],) int64 or NULL  = tokenizer(...)["token_type_ids"] (NULL: all 0)
 *   cu_seqlens     (B+1,) int32 HOST pointer       (attention_mask is all ones per text: each text attends to its own tokens)
 *   out            (B, hidden) fp32: host pointer, or a device pointer with EV_FLAG_DEVICE_INPUTS (then ids are device pointers too) */
int ev_style_embed(ev_handle* h, int B, const int64_t* input_ids, const int64_t* token_type_ids, const int32_t* cu_seqlens,
                   uint32_t flags, float* out);

/* Copy a named stage tap (SURVEY.md Appendix C names) of the LAST call to host memory as fp32
 * (integer taps as int64), in the packed utterance-major layout (rows x channels, valid rows only).
 * Returns the number of bytes written, or a negative error (e.g. cap too small, unknown name,
 * keep_stages disabled).  With host_dst == NULL returns the required size. */
int64_t ev_get_stage(ev_handle* h, const char* name, void* host_dst, size_t cap);

/* Convenience for callers without a HIP runtime binding (numpy/ctypes): synchronous device -> host copy
 * of one of the result pointers. */
int ev_memcpy_d2h(ev_handle* h, void* host_dst, const void* dev_src, size_t nbytes);

/* Timing of the last call, measured with hipEvents on the handle's stream (ms).  Names: "total",
 * "am", "encoder", "variance", "decoder", "vocoder".  Enabled by ev_set_profiling(h, 1). */
int ev_set_profiling(ev_handle* h, int enable);
int ev_get_timing(ev_handle* h, const char* name, float* ms);

/* Per-kernel-family accounting of the last call (profiling enabled): number of launches, summed
 * hipEvent duration and algorithmic FLOPs / bytes.  idx in [0, ev_kernel_stat_count). */
typedef struct ev_kernel_stat {
    char name[48];
    int32_t launches;
    float ms;
    double flops;
    double bytes;
} ev_kernel_stat;
int ev_kernel_stat_count(ev_handle* h);
int ev_get_kernel_stat(ev_handle* h, int idx, ev_kernel_stat* out);

/* Per-LAUNCH records of the last profiled call, in launch order: kernel family, the GEMM / conv shape (M rows, N output channels,
 * K input channels, taps, dilation; 0 for non-GEMM kernels), hipEvent duration and algorithmic FLOPs / bytes.  Lets a reader
 * recompute TF/s and TB/s per layer (profiles/ r2_*_launches.json). */
typedef struct ev_launch_record {
    char name[48];
    int32_t M, N, K, taps, dil;
    float ms;
    double flops;
    double bytes;
} ev_launch_record;
int ev_launch_record_count(ev_handle* h);
int ev_get_launch_record(ev_handle* h, int idx, ev_launch_record* out);

#ifdef __cplusplus
}
#endif
#endif /* EVHIP_H_ */
