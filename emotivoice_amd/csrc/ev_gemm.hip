// Implicit-GEMM convolution / linear kernel for gfx950 (CDNA4), channels-last activations.
//
//   D[co][t] = sum_{tap} sum_{k} W[co][tap][k] * X[t + (tap-center)*dil][k]
//
// The MFMA "A" operand is the weight tile (rows = output channels) and the "B" operand is the
// activation tile (columns = time), so that each lane ends up holding 4 consecutive output
// channels of one time step -> 8-byte (fp16) / 16-byte (fp32) channels-last stores.
// The activation slab [BM + (taps-1)*dil rows][64 bytes of K] is staged ONCE per K-chunk in LDS
// and re-used by every tap (this is what makes a k=11 dilated conv 11x more arithmetic-intense
// than a GEMM on an im2col matrix); the weight tile of each (K-chunk, tap) step is double-buffered.
// Staged rows are 64 B with an XOR swizzle of the 16-byte parts (see "main kernel" below) so that
// the 16 rows of a ds_read_b128 fragment read hit distinct LDS slots at any tap offset.
//
// File map: generic LDS-transposed epilogue (every operand a run-time flag; fp32 / split kernels' fallback) ->
// specialised straight-line epilogue (template flags, the frame-rate path) -> main conv-GEMM kernel -> phased 8-wave kernel
// (LDS-DMA staging, alternating matrix / load phases: the large fp16 launches) + launch heuristics -> split-precision kernels (fp16 hi/lo, 3 MFMAs per product: first generation 128 x 64 / 4 waves for few-tile
// token-rate GEMMs, second generation 256 x 128 / 8 waves for the EV_PREC_X3 frame-rate path) -> fused ResBlock-pair kernels
// for C = 32 / C = 64 -> per-device setup.
//
// DT_F16: v_mfma_f32_16x16x32_f16, fp32 accumulate.  DT_F32: v_mfma_f32_16x16x4_f32 (bit-exact
// fp32 FMA chain) for the duration-critical token-rate path.  Both share the byte geometry:
// one 16-byte fragment read feeds 1 f16 MFMA (K=32: lane group q holds k = 8q..8q+7) or 4 f32
// MFMAs (K=16 per read: lane group q holds k = 4q..4q+3, MFMA e consumes element e of both operands).
//
// Replaces (reference): nn.Linear / Conv1d / ConvTranspose1d calls of
// models/prompt_tts_modified/modules/encoder.py:50-52,72-109, modules/variance.py:41-46,
// model_open_source.py:111,147 and models/hifigan/models.py:50-57,116-128.
#include <hip/hip_fp16.h>
#include <stdlib.h>

#include "ev_kernels.h"
#include "ev_mxq.h"

// No implicit FMA contraction in this file.  The engine promises results that do not depend on how an utterance is batched
// (tests: test_batch_invariance_bit_exact), but which kernel / tile / epilogue variant a GEMM takes DOES depend on the batch's
// row count.  All variants accumulate the MFMA products in the same order; what differed (found by the split-precision mode's
// bit-exact test) was hipcc contracting "(v + res) * scale + acc" into an fma in the straight-line epilogue and not in the
// branchy one.  With contraction off every epilogue rounds each operation separately, whatever the code shape; explicit fmaf()
// calls (gelu_fast) are unaffected.
#pragma clang fp contract(off)

namespace ev {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

static constexpr int MAX_SPAN = 64;

// v_max without the canonicalising "v_max x, x" hipcc puts in front of every IEEE-mode fmax whose operand is not provably
// quiet (one extra VALU per element in VALU-bound epilogues); the operands here are finite activations.
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// max(|a|, |b|, m) in one VALU (source modifiers), m >= 0: the block maxima of mx_quant8 took an and + a max per element before
__device__ __forceinline__ float max3_abs_raw(float a, float b, float m) {
    float r;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}
// max over lanes l, l ^ 16, l ^ 32, l ^ 48 without the LDS pipe (__shfl_xor is ds_bpermute_b32): v_permlane32_swap / v_permlane16_swap exchange
// the upper half (odd 16-lane rows) of one register with the lower half (even rows) of another; applied to two copies of v they leave
// {lo, lo} and {hi, hi}, whose maximum is the xor-32 (xor-16) reduction on every lane.  (inline asm: hipcc's builtin folded the second result away)
__device__ __forceinline__ float max_xor32_raw(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return max_raw(a, b);
}
__device__ __forceinline__ float max_xor16_raw(float v) {
    float a = v, b = v;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return max_raw(a, b);
}
__device__ __forceinline__ unsigned pk_max_f16_raw(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint4 lrelu_h8(uint4 v, float slope) {
    half8 h = *reinterpret_cast<half8*>(&v);
    const _Float16 s = (_Float16)slope;
    h = h * s;                                  // slope in (0,1): max(x, slope*x) == leaky_relu(x)
    const uint4 t = *reinterpret_cast<uint4*>(&h);
    return make_uint4(pk_max_f16_raw(v.x, t.x), pk_max_f16_raw(v.y, t.y), pk_max_f16_raw(v.z, t.z), pk_max_f16_raw(v.w, t.w));
}
__device__ __forceinline__ uint4 lrelu_f4(uint4 v, float slope) {
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = fmaxf(f[i], f[i] * slope);
    return v;
}

// leaky-relu with slope in [0, 1] on a packed pair: max(v, slope * v) = v_pk_mul_f32 + 2 v_max_f32
__device__ __forceinline__ f32x2 lrelu2(f32x2 v, f32x2 slope2) {
    const f32x2 t = v * slope2;
    return f32x2{max_raw(v[0], t[0]), max_raw(v[1], t[1])};
}

// GELU for the fp16-output epilogue: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, two orders below the fp16 rounding
// of the stored result), branch-free: ~14 VALU + v_rcp + v_exp instead of libm erff's divergent range split (which made the
// decoder's conv1 GEMM 2.6x slower than its equally large conv2).  The fp32 kernels keep the exact erff (apply_act).
__device__ __forceinline__ float gelu_fast(float v) {
    const float a = fabsf(v);
    const float t = __builtin_amdgcn_rcpf(fmaf(a, 0.3275911f * 0.70710678118654752440f, 1.0f));
    float pl = fmaf(t, 1.061405429f, -1.453152027f);
    pl = fmaf(pl, t, 1.421413741f);
    pl = fmaf(pl, t, -0.284496736f);
    pl = fmaf(pl, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(v * v * (-0.5f * 1.44269504088896340736f));
    const float q = 0.5f * v * (pl * t * e);            // = 0.5 v erfc(|v| / sqrt 2)
    return v > 0.f ? v - q : q;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        case ACT_LRELU: return v > 0.f ? v : v * slope;
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// ---- coalesced epilogue: every wave transposes its accumulator tile through LDS (32 rows at a time) so that each lane
// ends up with 8 CONSECUTIVE output channels of one time step and a wave-instruction stores whole 64..128-byte row segments
// of consecutive rows (the MFMA layout gives 4 channels x 16 strided rows per lane: 32-byte segments, which measured at
// 20 % (C=128) to 60 % (C=32) of the kernel time).  Bias / activation / residual / MRF accumulate are applied on the
// coalesced side with 16/32-byte loads.
template <int TC>
__device__ __forceinline__ constexpr int epi_pitch() { return TC * 4 + 16; }
template <int TC>
__device__ __forceinline__ constexpr int epi_wave_bytes() { return 32 * epi_pitch<TC>(); }

template <int MT, int NT>
__device__ __forceinline__ void gemm_epilogue_lds(const ConvGemmParams& p, f32x4 (&acc)[NT][MT], char* wave_lds, int t0, int co0,
                                                  int t_end = 0x7fffffff) {
    constexpr int TC = NT * 16;
    constexpr int PITCH = epi_pitch<TC>();
    constexpr int LPR = TC / 8;          // lanes per row (8 channels each)
    constexpr int RPI = 64 / LPR;        // rows per wave-instruction
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane / LPR, g = lane % LPR;
    const int co = co0 + g * 8;
    float bias8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bias8[e] = 0.f;
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
        bias8[0] = b0.x; bias8[1] = b0.y; bias8[2] = b0.z; bias8[3] = b0.w; bias8[4] = b1.x; bias8[5] = b1.y; bias8[6] = b1.z; bias8[7] = b1.w;
    }
#pragma unroll
    for (int pass = 0; pass < MT / 2; ++pass) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                *reinterpret_cast<f32x4*>(wave_lds + (bb * 16 + fr) * PITCH + (a * 16 + 4 * fq) * 4) = acc[a][pass * 2 + bb];
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int lr = it * RPI + rr;
            const int t = t0 + pass * 32 + lr;
            if (t >= t_end) continue;            // rows owned by the next tile (fused pair kernel) or beyond the tensor
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(wave_lds + lr * PITCH + g * 32);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(wave_lds + lr * PITCH + g * 32 + 16);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            const bool valid = p.row_valid ? (p.row_valid[t >> p.valid_shift] != 0) : true;
            if (valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += bias8[e];
                if (p.act != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act, p.act_slope);
                }
                if (p.seq_bias) {
                    const float* sb = p.seq_bias + (long)p.row_seq[t] * p.ld_seq_bias + co;
                    const float4 s0 = *reinterpret_cast<const float4*>(sb), s1 = *reinterpret_cast<const float4*>(sb + 4);
                    v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
                }
                if (p.res) {
                    if (p.res_dtype == DT_F16) {
                        const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(p.res) + (long)t * p.ldres + co);
                        const __half2* h = reinterpret_cast<const __half2*>(&r);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); v[2 * e] += f.x; v[2 * e + 1] += f.y; }
                    } else {
                        const float* rp = reinterpret_cast<const float*>(p.res) + (long)t * p.ldres + co;
                        const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                        v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                    }
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
                if (p.acc32) {
                    const float* rp = p.acc32 + (long)t * p.ldacc + co;
                    const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                    v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
                }
                if (p.add16_a) {
#pragma unroll
                    for (int ab = 0; ab < 2; ++ab) {
                        const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(ab ? p.add16_b : p.add16_a) + (long)t * p.ldadd + co);
                        const __half2* h = reinterpret_cast<const __half2*>(&r);
#pragma unroll
                        for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); v[2 * e] += f.x; v[2 * e + 1] += f.y; }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
            }
            if (p.out32 && p.out32_before_post) {
                float* op = p.out32 + (long)t * p.ldo + co;
                *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (p.post_lrelu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.post_slope;
            }
            if (p.out32 && !p.out32_before_post) {
                float* op = p.out32 + (long)t * p.ldo + co;
                *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
            if (p.out16) {
                uint4 o;
                __half2* h = reinterpret_cast<__half2*>(&o);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = __floats2half2_rn(v[2 * e], v[2 * e + 1]);
                *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(p.out16) + (long)t * p.ldo + co) = o;
            }
        }
    }
}

// Tuning instrumentation (-DEV_TRACE, tools/trace_gemm.py; never part of the product build): every wave stamps s_memtime
// around each barrier into LDS, sampled blocks dump their stamps + HW_ID at the end.
#ifdef EV_TRACE
static constexpr int TRACE_N = 192;
static unsigned* g_trace_ptr = nullptr;
static int g_trace_flags = 0;                     // bit 0: drop the output stores (compute + loads only ceiling)
extern "C" void ev_trace_set(void* p) { g_trace_ptr = reinterpret_cast<unsigned*>(p); }
extern "C" void ev_trace_flags(int f) { g_trace_flags = f; }
#define EV_TRACE_ARG , unsigned* trace_out, int trace_flags
#define EV_TRACE_EPI_PARAMS , unsigned* tr_lds, int& tr_i, int wave
#define EV_TRACE_EPI_ARGS , tr_lds, tr_i, wave
#define EV_TRACE_EPI_DUMMY unsigned* tr_lds = nullptr; int tr_i = TRACE_N;      /* kernels without a timeline: stamps disabled */
#define EV_STAMP()                                                                      \
    {                                                                                   \
        const unsigned t_ = (unsigned)__builtin_readcyclecounter();                     \
        if (lane == 0 && tr_i < TRACE_N) tr_lds[wave * TRACE_N + tr_i] = t_;            \
        ++tr_i;                                                                         \
    }
#else
#define EV_TRACE_ARG
#define EV_TRACE_EPI_PARAMS
#define EV_TRACE_EPI_ARGS
#define EV_TRACE_EPI_DUMMY
#define EV_STAMP()
#endif

// ---- fast epilogue of the fp16 kernel.  Same transposed, coalesced stores as gemm_epilogue_lds, but the memory instruction
// stream is straight-line: per-wave timelines (tools/trace_gemm.py) showed 30-50 % of a tile's lifetime in the generic
// epilogue, because every iteration loaded its row_valid byte (and residual) inside runtime-flag branches and hipcc had to
// place s_waitcnt vmcnt(0) in every iteration -- on gfx9 stores share that counter, so each 1-KB store waited for the previous
// one's round trip.  Here (a) the optional operands are template flags, (b) all row_valid bytes of the wave's tile are
// requested before the first store (a null row_valid reads a constant 1 through a zero shift), (c) the residual / MRF rows of
// pass p+1 are requested before the stores of pass p, (d) row masking is a select, not a branch.
// RARE_ACT: relu / gelu / tanh (runtime switch).  O16 / O32: exactly one fp16 / one fp32 (after post) output, so that the number
// of stores per iteration is a compile-time constant and the counted vmcnt waits of the prefetches never drain them.
// LEAN: the operands of a pass are requested at the top of THAT pass instead of one pass ahead (one register set instead of two:
// the MRF variant of the 128-register phased kernel, whose epilogue is covered by the CU's other block anyway).
// MXP: additionally (or, without O32 / O16, only) the MX plane set of lrelu(result, mxo_slope) for a DT_MX consumer (mx_emit_planes).
// RESPL: the residual comes from a plane set (ConvGemmParams::res_x4 ...): fp16 hi plane + fp4 codes of the remainder + their block scales, 2.5 bytes
// per element instead of 4, and no fp32 copy of the residual stream has to be written by its producer.
// ACCPL: the accumulate-in operand is a PARTIAL plane set (ConvGemmParams::acc_h ...: fp16 hi plane + fp4 codes of the remainder + their block scales, no
// activation) instead of an fp32 tensor; PART (with MXP): the output plane set is such a partial one -- no hi-code plane, no hi scales.  Together they carry the
// MRF sum of a stage's three ResBlocks at 2.53 instead of 4 bytes per element and transfer (the running sum is re-quantised once per ResBlock).
enum { EPI_RES16 = 1, EPI_RES32 = 2, EPI_ACC32 = 4, EPI_GENERIC = 8, EPI_RARE_ACT = 16, EPI_O16 = 32, EPI_O32 = 64, EPI_ADD16 = 128, EPI_LEAN = 256,
       EPI_MXP = 512, EPI_RESPL = 1024, EPI_ACCPL = 2048, EPI_PART = 4096 };

// ---- MX plane set of an activation (format: ev_gemm_mx.h / emotivoice_amd/mxfp4.py; mx_scale_byte / mx_fp4_code: ev_mxq.h)
// value of lane ^ 1 / lane ^ 2 (v_mov_b32_dpp quad_perm:[1,0,3,2] / [2,3,0,1])
__device__ __forceinline__ float quad_xor1(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)); }
__device__ __forceinline__ float quad_xor2(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)); }
// 8 consecutive channels a[0..3] (packed pairs) of one lane -> their share of the plane set: ho = the fp16 hi parts, ch / cl = the eight fp4
// codes of the hi parts / of the fp32 remainders, bh / bl = the E8M0 scale bytes of the 32-channel block the lane belongs to.  The four
// lanes of a block are consecutive (lane & 3 = position in the block) and agree on the block maxima by two quad shuffles.
__device__ __forceinline__ void mx_quant8(const f32x2 (&a)[4], uint4& ho, unsigned& ch, unsigned& cl, unsigned& bh, unsigned& bl) {
    half2v hh[4];
    f32x2 hf[4], lf[4];
    float mh = 0.f, ml = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hh[j] = __builtin_convertvector(a[j], half2v);
        hf[j] = __builtin_convertvector(hh[j], f32x2);
        lf[j] = a[j] - hf[j];
        mh = max3_abs_raw(hf[j][0], hf[j][1], mh);
        ml = max3_abs_raw(lf[j][0], lf[j][1], ml);
    }
    // (DPP quad permutes: __shfl_xor compiles to ds_bpermute_b32 -- an LDS-pipe round trip with its own address arithmetic, four of them in
    // a dependent chain per call, which measured a third of conv_c64_mx_kernel's k = 3 launches)
    mh = max_raw(mh, quad_xor1(mh)); mh = max_raw(mh, quad_xor2(mh));
    ml = max_raw(ml, quad_xor1(ml)); ml = max_raw(ml, quad_xor2(ml));
    bh = mx_scale_byte(mh); bl = mx_scale_byte(ml);
    const float sh = __uint_as_float(bh << 23), sl = __uint_as_float(bl << 23);       // the block scales as fp32 (2^(b - 127))
    ch = 0; cl = 0;                                                                    // v_cvt_scalef32_pk_fp4_f32: fp4(x / scale), RNE, saturating
    ch = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ch, hf[0][0], hf[0][1], sh, 0); cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[0][0], lf[0][1], sl, 0);
    ch = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ch, hf[1][0], hf[1][1], sh, 1); cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[1][0], lf[1][1], sl, 1);
    ch = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ch, hf[2][0], hf[2][1], sh, 2); cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[2][0], lf[2][1], sl, 2);
    ch = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ch, hf[3][0], hf[3][1], sh, 3); cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[3][0], lf[3][1], sl, 3);
    ho.x = *reinterpret_cast<unsigned*>(&hh[0]); ho.y = *reinterpret_cast<unsigned*>(&hh[1]);
    ho.z = *reinterpret_cast<unsigned*>(&hh[2]); ho.w = *reinterpret_cast<unsigned*>(&hh[3]);
}
// One lane = 8 consecutive channels at linear element offset `lin` of the [rows][C] tensor: writes 16 B of the fp16 hi plane, 4 B of each
// code plane and (first lane of the block) one byte of each scale plane.
// (row, co) = the element's position in the launch's own [M][N] output; mxo_logC == 0: that IS the plane set's geometry (C = N, any multiple
// of 128), else the planes are [M * N / C][C] with C = 2^mxo_logC (a transposed conv's polyphase output).
// byte offset of the scale of element (row, co) of the launch's [M][N] output inside a scale plane
__device__ __forceinline__ long mx_scale_offset(const ConvGemmParams& p, long lin, long row_, int co) {
    const long row = p.mxo_logC ? (lin >> p.mxo_logC) : row_;
    const unsigned c = p.mxo_logC ? ((unsigned)lin & ((1u << p.mxo_logC) - 1u)) : (unsigned)co;
    return (long)(c >> 7) * p.mxo_qs_stride + row * 4 + ((c >> 5) & 3);
}
// A wave-uniform pointer pinned to an SGPR pair, opaque to the optimiser (an empty asm with an "s" constraint; readfirstlane folds away when the value
// already is scalar, and LLVM then re-associates `uniform + zext(lane offset)` into one 64-bit VALU add per access instead of the `saddr + voffset`
// addressing form).  The value must be wave-uniform.
template <typename T>
__device__ __forceinline__ T* uptr(T* q) {
    unsigned long long b = reinterpret_cast<unsigned long long>(q);
    asm("" : "+s"(b));
    typedef __attribute__((address_space(1))) T* gptr;        // (an integer -> pointer cast alone gives a generic pointer: flat_load / flat_store)
    return (T*)(gptr)b;
}
// ... and the lane part re-issued where it is used: hoisted out of the unrolled row loop, its zero-extension lives in another basic block, instruction
// selection no longer sees `sgpr + zext(vgpr32)` and falls back to a 64-bit VALU add per access (no instruction is emitted for this)
__device__ __forceinline__ unsigned vlane(unsigned x) { asm volatile("" : "+v"(x)); return x; }
// (so = mx_scale_offset of the element: the caller advances it by 4 bytes per plane-set row instead of recomputing it -- ~12 VALU per call.
//  Addresses are split into a wave-uniform 64-bit part (ulin / uso: SALU arithmetic, an SGPR base) and a 32-bit lane part (llin / lso, fixed per
//  lane for the whole epilogue): the stores take the `saddr + voffset` form and cost no 64-bit VALU adds per row)
template <bool PART = false>
__device__ __forceinline__ void mx_emit_planes(const ConvGemmParams& p, const f32x2 (&a)[4], long ulin, unsigned llin_, long uso, unsigned lso_, int lane) {
    uint4 ho;
    unsigned ch, cl, bh, bl;
    mx_quant8(a, ho, ch, cl, bh, bl);
    const unsigned lh = vlane(llin_ * 2u), lq = vlane(llin_ >> 1), lso = vlane(lso_);      // (the products are loop-invariant: hoisted, re-issued as moves)
    char* const ph = uptr(reinterpret_cast<char*>(p.mxo_h) + ulin * 2);
    char* const pq0 = uptr(reinterpret_cast<char*>(p.mxo_q4[0]) + (ulin >> 1));
    char* const pq1 = uptr(reinterpret_cast<char*>(p.mxo_q4[1]) + (ulin >> 1));
#ifdef EV_MX_ABL            // tuning build (tools/bench_mxgemm.py): reserved0 bit 6 = no plane stores at all, bit 7 = fp16 hi plane only, bit 8 = no scale bytes
    const int abl = p.reserved0 >> 4;
    if (abl & 4) { if (p.M < 0) *reinterpret_cast<uint4*>(ph + lh) = uint4{ho.x ^ ch, ho.y ^ cl, ho.z ^ bh, ho.w ^ bl}; return; }
    if (abl & 8) { if (p.M < 0) ho.x ^= ch ^ cl ^ bh ^ bl; *reinterpret_cast<uint4*>(ph + lh) = ho; return; }
    if (abl & 16) {
        if (p.M < 0) ho.x ^= bh ^ bl;
        *reinterpret_cast<uint4*>(ph + lh) = ho;
        *reinterpret_cast<unsigned*>(pq0 + lq) = ch;
        *reinterpret_cast<unsigned*>(pq1 + lq) = cl;
        return;
    }
#endif
    *reinterpret_cast<uint4*>(ph + lh) = ho;
    if constexpr (!PART) *reinterpret_cast<unsigned*>(pq0 + lq) = ch;
    *reinterpret_cast<unsigned*>(pq1 + lq) = cl;
    if ((lane & 3) == 0) {
        if constexpr (!PART) uptr(reinterpret_cast<uint8_t*>(p.mxo_qs[0]) + uso)[lso] = (uint8_t)bh;
        uptr(reinterpret_cast<uint8_t*>(p.mxo_qs[1]) + uso)[lso] = (uint8_t)bl;
    }
}
// wave-uniform copy of a 64-bit value (lane 0's)
__device__ __forceinline__ long uniform64(long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long)v >> 32));
    return (long)(((unsigned long)hi << 32) | lo);
}
__device__ uint8_t g_row_always_valid[4] = {1, 1, 1, 1};   // not const: a constant-address-space object would turn the select below into FLAT loads

// RP = rows per transposing pass: 32 (scratch of 32 padded rows = 8.5 KB per wave, aliasing the staging buffers) or 16 (the persistent MX kernel: 4 KB per
// wave, 256-byte rows with an XOR swizzle of the 16-byte column -- conflict-free for the ds_write_b128 / ds_read_b128 lane groups of MI355X_MICROARCH.md --
// which leaves one slab buffer and three ring slots free for the next tile's first requests while this one drains).  Same values, same order per element.
template <int MT, int NT, int EPI, int RP = 32>
__device__ __forceinline__ void gemm_epilogue_fast(const ConvGemmParams& p, f32x4 (&acc)[NT][MT], char* wave_lds, int t0, int co0 EV_TRACE_EPI_PARAMS) {
    constexpr int TC = NT * 16;
    constexpr bool SWZ = RP == 16;
    constexpr int PITCH = SWZ ? TC * 4 : epi_pitch<TC>();
    // EV_MXT (tuning build, tools/bench_mxgemm.py --timeline): s_memtime at the entry, in front of the pass loop, inside every pass (scratch written + read back
    // issued, i.e. behind the pass's operand requests) and behind every pass; lane 0 of every wave dumps them behind the blocks' records of conv_gemm_mx_kernel
#ifdef EV_MXT
    unsigned long long mxt_e[12];
    int mxt_i = 0;
#define EV_MXT_STAMP() { __builtin_amdgcn_sched_barrier(0); mxt_e[mxt_i++] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
#else
#define EV_MXT_STAMP()
#endif
    EV_MXT_STAMP()
    constexpr int LPR = TC / 8, RPI = 64 / LPR, IT = RP / RPI, NP = MT * 16 / RP, NG = MT / 2;
    static_assert((RP == 32 || (RP == 16 && TC == 64)) && IT >= 1, "rows per pass");
    int lane_ = threadIdx.x & 63;
    if constexpr (SWZ) asm volatile("" : "+v"(lane_));          // (called once per tile of a persistent block: nothing lane-derived may become an invariant of the tile
                                                                //  loop, or it is live through the K loop -- spilled registers, reloaded behind an s_waitcnt vmcnt(0))
    const int lane = lane_;
    const int fr = lane & 15, fq = lane >> 4;
    const int rr = lane / LPR, g = lane % LPR;
    const int co = co0 + g * 8;
    const int trow = t0 + rr;                                  // row of (pass 0, it 0); + pass * 32 + it * RPI
    // Every tensor address below = wave-uniform 64-bit part (rows t0u + pass * 32 + it * RPI, channel co0u: SALU, an SGPR base) + a 32-bit lane
    // part (row rr, channel g * 8) that is fixed for the whole epilogue: global_load / global_store `saddr + voffset`, no 64-bit VALU adds per row.
    const int t0u = __builtin_amdgcn_readfirstlane(t0), co0u = __builtin_amdgcn_readfirstlane(co0);
    const unsigned l_o0 = (unsigned)(rr * p.ldo + g * 8), l_res0 = (unsigned)(rr * p.ldres + g * 8), l_acc0 = (unsigned)(rr * p.ldacc + g * 8),
                   l_add0 = (unsigned)(rr * p.ldadd + g * 8);                      // in elements; the byte offsets below are what the accesses use
    const unsigned b_o32 = l_o0 * 4u, b_o16 = l_o0 * 2u, b_res16 = l_res0 * 2u, b_res32 = l_res0 * 4u, b_resq = l_res0 >> 1, b_acc = l_acc0 * 4u, b_add = l_add0 * 2u;
    const unsigned b_acc16 = l_acc0 * 2u, b_accq = l_acc0 >> 1;

    // one byte load per 32-row pass (lane l <-> row l & 31), turned into a wave-uniform bit mask by a ballot: no VGPRs held
    const uint8_t* vptr = p.row_valid ? p.row_valid : g_row_always_valid;
    const int vshift = p.row_valid ? p.valid_shift : 31;
    uint8_t vld[NG];
#pragma unroll
    for (int pass = 0; pass < NG; ++pass) vld[pass] = vptr[(t0 + pass * 32 + (lane & 31)) >> vshift];
    uint4 r16[2][IT], a16[2][IT][2];
    float4 r32[2][IT][2], a32[2][IT][2];
    uint4 rph[2][IT];                       // EPI_RESPL: 8 fp16 hi parts, 8 fp4 remainder codes, the block's scale byte
    unsigned rpc[2][IT], rps[2][IT];
    long rp_su = 0;                         // scale byte of (row, co): [co >> 7][row][(co >> 5) & 3]; lane 0 has the lowest address of the wave
    unsigned rp_sl0 = 0;
    if constexpr (EPI & EPI_RESPL) {
        const long so = (long)(co >> 7) * p.res_xs_stride + ((co >> 5) & 3) + (long)rr * 4;
        rp_su = uniform64(so) + (long)t0u * 4;
        rp_sl0 = (unsigned)(so - uniform64(so));
    }
    uint4 aph[2][IT];                       // EPI_ACCPL: the partial plane set of the accumulate-in operand, as rph / rpc / rps
    unsigned apc[2][IT], aps[2][IT];
    long ap_su = 0;
    unsigned ap_sl0 = 0;
    if constexpr (EPI & EPI_ACCPL) {
        const long so = (long)(co >> 7) * p.acc_xs_stride + ((co >> 5) & 3) + (long)rr * 4;
        ap_su = uniform64(so) + (long)t0u * 4;
        ap_sl0 = (unsigned)(so - uniform64(so));
    }
    constexpr bool LEAN = (EPI & EPI_LEAN) != 0;
    // byte offset of 16-byte half H of the lane's 8 channels in transposed row IT_ * RPI + rr of the scratch
#define EV_EPI_RD(IT_, H) (SWZ ? ((IT_) * RPI + rr) * PITCH + (((2 * g + (H)) ^ (((IT_) * RPI + rr) & 15)) << 4) : ((IT_) * RPI + rr) * PITCH + g * 32 + (H) * 16)
#define EV_EPI_SET(PASS) (LEAN ? 0 : ((PASS) & 1))
#define EV_EPI_PREFETCH(PASS)                                                                                              \
    _Pragma("unroll") for (int it = 0; it < IT; ++it) {                                                                    \
        const long tu_ = t0u + (PASS) * RP + it * RPI;                                                                     \
        if constexpr (EPI & EPI_RES16)                                                                                     \
            r16[EV_EPI_SET(PASS)][it] = *reinterpret_cast<const uint4*>(uptr(reinterpret_cast<const char*>(p.res) + (tu_ * p.ldres + co0u) * 2) + vlane(b_res16)); \
        if constexpr (EPI & EPI_RES32) {                                                                                   \
            const char* rp_ = uptr(reinterpret_cast<const char*>(p.res) + (tu_ * p.ldres + co0u) * 4) + vlane(b_res32);    \
            r32[EV_EPI_SET(PASS)][it][0] = *reinterpret_cast<const float4*>(rp_);                                          \
            r32[EV_EPI_SET(PASS)][it][1] = *reinterpret_cast<const float4*>(rp_ + 16);                                     \
        }                                                                                                                  \
        if constexpr (EPI & EPI_RESPL) {                                                                                   \
            const long ulin_ = tu_ * p.ldres + co0u;                                                                       \
            rph[EV_EPI_SET(PASS)][it] = *reinterpret_cast<const uint4*>(uptr(reinterpret_cast<const char*>(p.res) + ulin_ * 2) + vlane(b_res16));  \
            rpc[EV_EPI_SET(PASS)][it] = *reinterpret_cast<const unsigned*>(uptr(reinterpret_cast<const char*>(p.res_x4) + (ulin_ >> 1)) + vlane(b_resq)); \
            rps[EV_EPI_SET(PASS)][it] = uptr(reinterpret_cast<const uint8_t*>(p.res_xs) + (rp_su + ((PASS) * RP + it * RPI) * 4))[vlane(rp_sl0)]; \
        }                                                                                                                  \
        if constexpr (EPI & EPI_ACCPL) {                                                                                   \
            const long ulin_ = tu_ * p.ldacc + co0u;                                                                       \
            aph[EV_EPI_SET(PASS)][it] = *reinterpret_cast<const uint4*>(uptr(reinterpret_cast<const char*>(p.acc_h) + ulin_ * 2) + vlane(b_acc16));  \
            apc[EV_EPI_SET(PASS)][it] = *reinterpret_cast<const unsigned*>(uptr(reinterpret_cast<const char*>(p.acc_x4) + (ulin_ >> 1)) + vlane(b_accq)); \
            aps[EV_EPI_SET(PASS)][it] = uptr(reinterpret_cast<const uint8_t*>(p.acc_xs) + (ap_su + ((PASS) * RP + it * RPI) * 4))[vlane(ap_sl0)]; \
        }                                                                                                                  \
        if constexpr (EPI & EPI_ACC32) {                                                                                   \
            const char* ap_ = uptr(reinterpret_cast<const char*>(p.acc32) + (tu_ * p.ldacc + co0u) * 4) + vlane(b_acc);    \
            a32[EV_EPI_SET(PASS)][it][0] = *reinterpret_cast<const float4*>(ap_);                                          \
            a32[EV_EPI_SET(PASS)][it][1] = *reinterpret_cast<const float4*>(ap_ + 16);                                     \
        }                                                                                                                  \
        if constexpr (EPI & EPI_ADD16) {                                                                                   \
            a16[EV_EPI_SET(PASS)][it][0] = *reinterpret_cast<const uint4*>(uptr(reinterpret_cast<const char*>(p.add16_a) + (tu_ * p.ldadd + co0u) * 2) + vlane(b_add)); \
            a16[EV_EPI_SET(PASS)][it][1] = *reinterpret_cast<const uint4*>(uptr(reinterpret_cast<const char*>(p.add16_b) + (tu_ * p.ldadd + co0u) * 2) + vlane(b_add)); \
        }                                                                                                                  \
    }
    if constexpr (!LEAN) { EV_EPI_PREFETCH(0) }
    unsigned vmask[NG];
#pragma unroll
    for (int pass = 0; pass < NG; ++pass) vmask[pass] = (unsigned)__builtin_amdgcn_ballot_w64(vld[pass] != 0);

    // The per-element VALU work bounds this epilogue (timeline: ~100 VALU per 8 outputs took ~800 cycles per iteration with or
    // without the stores), so everything is kept in packed-fp32 pairs (v_pk_add / v_pk_mul), leaky-relu is max(v, slope * v)
    // (slope in [0, 1], checked by the launcher), row masking is skipped for fully valid 32-row groups and works on the packed
    // fp16 words otherwise.
    f32x2 bias2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias2[j] = f32x2{0.f, 0.f};
    if (p.bias) {
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
        bias2[0] = f32x2{b0.x, b0.y}; bias2[1] = f32x2{b0.z, b0.w}; bias2[2] = f32x2{b1.x, b1.y}; bias2[3] = f32x2{b1.z, b1.w};
    }
    const int act = p.act;
    const f32x2 act_slope2 = f32x2{p.act_slope, p.act_slope};
    const f32x2 out_scale2 = f32x2{p.out_scale, p.out_scale};
    const bool has_post = p.post_lrelu != 0;
    const f32x2 post_slope2 = f32x2{p.post_slope, p.post_slope};
    const f32x2 mxo_slope2 = f32x2{p.mxo_slope, p.mxo_slope};
    const f32x2 res_inv2 = f32x2{p.res_inv_slope, p.res_inv_slope};
    // (x * 1 is exact: skipping it changes no bit.  Compared as bit patterns: a float compare of two kernel arguments is a VALU compare whose lane mask the
    // unrolled loop below keeps re-deriving through a VGPR; the integer compare is one s_cmp)
    const bool mxo_act = __float_as_uint(p.mxo_slope) != 0x3f800000u, scaled = __float_as_uint(p.out_scale) != 0x3f800000u;
    __half* const o16 = reinterpret_cast<__half*>(p.out16);
    float* const o32a = p.out32_before_post ? p.out32 : nullptr;
    float* const o32b = p.out32_before_post ? nullptr : p.out32;
    const long rowoff_u = (long)t0u * p.ldo + co0u;             // uniform part of the output offsets; the lane part is l_o
    unsigned lrbit[32 / RPI];
#pragma unroll
    for (int it = 0; it < 32 / RPI; ++it) lrbit[it] = 1u << (it * RPI + rr);
    // plane-set output: the lane's scale byte moves by a fixed distance per output row (ldo == N is a multiple of the plane set's C, so a
    // step of one [M][N] row is N / C whole plane-set rows and the channel block stays the lane's own)
    long mx_su = 0;
    unsigned mx_sl = 0;
    int mx_sstep = 0;
    if constexpr (EPI & EPI_MXP) {
        // (the lanes of a wave cover 64 channels from a multiple of 64: one 128-channel group of one plane-set row per output row, so lane 0's scale
        //  byte has the lowest address of the wave whatever the plane geometry)
        static_assert(TC == 64, "plane-set epilogue: 64-channel wave tiles");
        const long so = mx_scale_offset(p, rowoff_u + l_o0, (long)trow, co);
        mx_su = uniform64(so);
        mx_sl = (unsigned)(so - mx_su);
        mx_sstep = 4 * (p.mxo_logC ? (p.ldo >> p.mxo_logC) : 1);
    }
    EV_STAMP()
    EV_MXT_STAMP()

#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
#ifdef EV_MXT
        // A/B (tools/bench_mxgemm.py --dbg 4): the two epilogue waves of a SIMD (w, w + 4) end 3-5 us apart because the arbiter prefers the older one; swap their
        // priorities half-way so that both end together and the block's slot is free earlier
        if constexpr (SWZ) {
            if (p.reserved0 & 4) {
                const bool late = (threadIdx.x >> 8) != 0;          // waves 4-7
                if (pass == 0) { if (late) __builtin_amdgcn_s_setprio(1); }
                if (pass == NP / 2) { if (late) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1); }
            }
        }
#endif
        __builtin_amdgcn_wave_barrier();
        if constexpr (LEAN) { EV_EPI_PREFETCH(pass) }
        if constexpr (SWZ) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
                *reinterpret_cast<f32x4*>(wave_lds + fr * PITCH + (((a * 4 + fq) ^ fr) << 4)) = acc[a][pass];
        } else {
#pragma unroll
            for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                for (int a = 0; a < NT; ++a)
                    *reinterpret_cast<f32x4*>(wave_lds + (bb * 16 + fr) * PITCH + (a * 16 + 4 * fq) * 4) = acc[a][pass * 2 + bb];
        }
        __builtin_amdgcn_wave_barrier();
        if constexpr (!LEAN) { if (pass + 1 < NP) { EV_EPI_PREFETCH(pass + 1) } }
        EV_STAMP()
        f32x4 lv[IT][2];          // all of the pass's transposed rows first: one exposed LDS latency per pass, not per iteration
        if constexpr (!LEAN) {
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                lv[it][0] = *reinterpret_cast<const f32x4*>(wave_lds + EV_EPI_RD(it, 0));
                lv[it][1] = *reinterpret_cast<const f32x4*>(wave_lds + EV_EPI_RD(it, 1));
            }
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            if constexpr (LEAN) {         // (row by row: 8 registers instead of 8 * IT)
                lv[it][0] = *reinterpret_cast<const f32x4*>(wave_lds + EV_EPI_RD(it, 0));
                lv[it][1] = *reinterpret_cast<const f32x4*>(wave_lds + EV_EPI_RD(it, 1));
            }
            const long off_u = rowoff_u + (long)(pass * RP + it * RPI) * p.ldo;
            const f32x4 v0 = lv[it][0], v1 = lv[it][1];
            f32x2 v[4] = {f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}};
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bias2[j];
            if constexpr (EPI & EPI_RARE_ACT) {
                if (act == ACT_GELU && (EPI & (EPI_O16 | EPI_MXP)) && !(EPI & EPI_O32)) {          // the A&S erf (|error| <= 1.5e-7) only where the result is rounded
                                                                                                    // to fp16 or to an MX plane set (fp16 hi + fp4 remainder: ~2^-14)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j][0] = gelu_fast(v[j][0]); v[j][1] = gelu_fast(v[j][1]); }
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j][0] = apply_act(v[j][0], act, 0.f); v[j][1] = apply_act(v[j][1], act, 0.f); }
                }
            } else if (act == ACT_LRELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = lrelu2(v[j], act_slope2);
            }
            if constexpr (EPI & EPI_RES16) {
                const half2v* h = reinterpret_cast<const half2v*>(&r16[EV_EPI_SET(pass)][it]);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += __builtin_convertvector(h[j], f32x2);
            }
            if constexpr (EPI & EPI_RES32) {
                const float4 r0 = r32[EV_EPI_SET(pass)][it][0], r1 = r32[EV_EPI_SET(pass)][it][1];
                v[0] += f32x2{r0.x, r0.y}; v[1] += f32x2{r0.z, r0.w}; v[2] += f32x2{r1.x, r1.y}; v[3] += f32x2{r1.z, r1.w};
            }
            if constexpr (EPI & EPI_RESPL) {
                // a' = hi + code * 2^(scale - 127) is the plane set's value of lrelu(x, 1 / inv): x = min(a', a' * inv) undoes it (inv >= 1)
                const half2v* h = reinterpret_cast<const half2v*>(&rph[EV_EPI_SET(pass)][it]);
                const unsigned cw = rpc[EV_EPI_SET(pass)][it];
                const float sc = __uint_as_float(rps[EV_EPI_SET(pass)][it] << 23);
                f32x2 a[4];
                a[0] = __builtin_convertvector(h[0], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, sc, 0);
                a[1] = __builtin_convertvector(h[1], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, sc, 1);
                a[2] = __builtin_convertvector(h[2], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, sc, 2);
                a[3] = __builtin_convertvector(h[3], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, sc, 3);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 t = a[j] * res_inv2;
                    v[j] += f32x2{min_raw(a[j][0], t[0]), min_raw(a[j][1], t[1])};
                }
            }
            if (scaled) {                 // (a real branch: as a select it is 12 VALU per 8 outputs in every launch that does not scale, and the two
                                          // value sets it creates cost another 12 moves at the row-mask join below)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] *= out_scale2;
            }
            if constexpr (EPI & EPI_ACC32) {
                const float4 r0 = a32[EV_EPI_SET(pass)][it][0], r1 = a32[EV_EPI_SET(pass)][it][1];
                v[0] += f32x2{r0.x, r0.y}; v[1] += f32x2{r0.z, r0.w}; v[2] += f32x2{r1.x, r1.y}; v[3] += f32x2{r1.z, r1.w};
            }
            if constexpr (EPI & EPI_ACCPL) {             // the partial's value is hi + code * 2^(scale - 127), no activation
                const half2v* h = reinterpret_cast<const half2v*>(&aph[EV_EPI_SET(pass)][it]);
                const unsigned cw = apc[EV_EPI_SET(pass)][it];
                const float sc = __uint_as_float(aps[EV_EPI_SET(pass)][it] << 23);
                v[0] += __builtin_convertvector(h[0], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, sc, 0);
                v[1] += __builtin_convertvector(h[1], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, sc, 1);
                v[2] += __builtin_convertvector(h[2], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, sc, 2);
                v[3] += __builtin_convertvector(h[3], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(cw, sc, 3);
            }
            if constexpr (EPI & EPI_ADD16) {
                const half2v* ha = reinterpret_cast<const half2v*>(&a16[EV_EPI_SET(pass)][it][0]);
                const half2v* hb = reinterpret_cast<const half2v*>(&a16[EV_EPI_SET(pass)][it][1]);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += __builtin_convertvector(ha[j], f32x2) + __builtin_convertvector(hb[j], f32x2);
            }
            // rows outside the utterances give zeros in every output (leaky-relu and the conversions keep +0): masked once, and only in a
            // 32-row group that has such rows at all (wave-uniform; VALU only, so no store count depends on it)
            if (vmask[pass * RP / 32] != 0xffffffffu) {
                const bool valid = (vmask[pass * RP / 32] & lrbit[(pass * RP % 32) / RPI + it]) != 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j][0] = valid ? v[j][0] : 0.f; v[j][1] = valid ? v[j][1] : 0.f; }
            }
            // (an EPI_MXP launch has no run-time outputs either: mx_epi_variant admits plane sets only beside / instead of ONE fp32 output)
            constexpr bool STATIC_OUT = (EPI & (EPI_O16 | EPI_O32 | EPI_MXP)) != 0;
            if (!STATIC_OUT && o32a) {
                char* op = uptr(reinterpret_cast<char*>(o32a + off_u)) + vlane(b_o32);
                *reinterpret_cast<float4*>(op) = make_float4(v[0][0], v[0][1], v[1][0], v[1][1]);
                *reinterpret_cast<float4*>(op + 16) = make_float4(v[2][0], v[2][1], v[3][0], v[3][1]);
            }
            if (!(EPI & EPI_MXP) && has_post) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = lrelu2(v[j], post_slope2);
            }
            if ((EPI & EPI_O32) || (!STATIC_OUT && o32b)) {
                char* op = uptr(reinterpret_cast<char*>(o32b + off_u)) + vlane(b_o32);
                *reinterpret_cast<float4*>(op) = make_float4(v[0][0], v[0][1], v[1][0], v[1][1]);
                *reinterpret_cast<float4*>(op + 16) = make_float4(v[2][0], v[2][1], v[3][0], v[3][1]);
            }
            if ((EPI & EPI_O16) || (!STATIC_OUT && o16)) {
                uint4 o;
                half2v* h = reinterpret_cast<half2v*>(&o);
#pragma unroll
                for (int j = 0; j < 4; ++j) h[j] = __builtin_convertvector(v[j], half2v);
                *reinterpret_cast<uint4*>(uptr(reinterpret_cast<char*>(o16 + off_u)) + vlane(b_o16)) = o;
            }
            if constexpr (EPI & EPI_MXP) {            // the consumer's leaky-relu, then its operand planes (ldo == N: `off` is the linear offset)
                f32x2 am[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) am[j] = v[j];
                if (mxo_act) {            // (slope 1 = none: conv1 of a ResBlock pair, whose own leaky-relu is `act`)
#pragma unroll
                    for (int j = 0; j < 4; ++j) am[j] = lrelu2(v[j], mxo_slope2);
                }
                mx_emit_planes<(EPI & EPI_PART) != 0>(p, am, off_u, l_o0, mx_su + (long)((pass * RP + it * RPI) * mx_sstep), mx_sl, lane);
            }
            EV_STAMP()
        }
        if constexpr (SWZ) { EV_MXT_STAMP() }
    }
#ifdef EV_MXT
    if constexpr (SWZ) {
        if (p.row_seq && lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<int32_t*>(p.row_seq)) + (size_t)gridDim.x * 4 + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 8;
            for (int i = 0; i < 2 + NP && i < 8; ++i) o[i] = mxt_e[i];
        }
    }
#endif
#undef EV_MXT_STAMP
#undef EV_EPI_PREFETCH
#undef EV_EPI_SET
#undef EV_EPI_RD
}

// ---- main kernel.  LDS: 64-byte pitch with an XOR swizzle (16-B part ^= (row >> 1) & 3): conflict-free ds_read_b128 for
// the 16-row fragment reads at ANY tap offset and conflict-free ds_write_b128 staging (bank model of MI355X_MICROARCH.md
// section LDS; the first version's 80-B padded pitch measured SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.50).
// Pipeline: the weight tile of step s+2 is loaded into registers while step s computes and is written to LDS one step
// later; the activation slab of K-chunk kc+1 is loaded at the first tap of chunk kc and written at its last tap.
// The steady-state loop body is branch-free (clamped addresses instead of predicates, exact chunk counts) so that hipcc
// keeps counted vmcnt waits; register sets rotate by moves (a runtime A/B parity or lambdas taking array references made
// hipcc spill the prefetch registers to scratch inside the loop).
__device__ __forceinline__ int swz(int row, int part) { return row * 64 + ((part ^ ((row >> 1) & 3)) << 4); }

// 2 waves / SIMD (<= 256 VGPRs) for the 256 x 128 tile, 3 for the smaller ones -- including their MRF-epilogue instantiations, which
// round 1 had capped at 2: at 3 the RES16 | ADD16 | O16 epilogue spills 6 registers, outside the main loop, and the last conv of
// stage 2 still gains 7 % (755 -> 705 us, profiles/r2_h_gemm_launch_bounds.txt).
#define EV_GEMM_MIN_WAVES(BM, BN, EPI) (((BM) * (BN) > 128 * 128) ? 2 : 3)
template <typename TIn, int BM, int BN, int WT, int WC, int EPI>
__global__ __launch_bounds__(256, EV_GEMM_MIN_WAVES(BM, BN, EPI)) void conv_gemm_kernel(const ConvGemmParams p EV_TRACE_ARG) {
    constexpr int ES = sizeof(TIn);
    constexpr int TT = BM / WT, TC = BN / WC, MT = TT / 16, NT = TC / 16;
    constexpr int SLAB = BM + MAX_SPAN;              // rows staged per K-chunk (>= BM + (taps-1)*dil, multiple of 64)
    constexpr int XCH = SLAB * 4 / 256;              // 16-B chunks per thread
    constexpr int WCH = (BN * 4 + 255) / 256;
    static_assert(WCH <= 2, "weight tile staging uses at most two 16-B chunks per thread");
    constexpr int XBUF = SLAB * 64, WBUF = BN * 64;
    static_assert(WT * WC == 4 && SLAB % 64 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wt = wave / WC, wc = wave % WC;
    char* Xs = smem;
    char* Ws = smem + 2 * XBUF;
#ifdef EV_TRACE
    constexpr int EPI_ = 4 * 32 * ((BN / WC) * 4 + 16), PIPE_ = 2 * XBUF + 2 * WBUF;
    unsigned* tr_lds = reinterpret_cast<unsigned*>(smem + (EPI_ > PIPE_ ? EPI_ : PIPE_));
    int tr_i = 0;
    EV_STAMP()
#endif

    // XCD-aware block remap (bijective): consecutive logical tiles -> same XCD (shared L2 for the A rows)
    const int nN = p.N / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int mb = bid / nN, nb = bid % nN;
    const int m0 = mb * BM, n0 = nb * BN;

    const int taps = p.taps;
    const int nkc = (p.K * ES) >> 6;
    const long a_pitch = (long)p.lda * ES;
    const long w_tap_pitch = (long)p.K * ES;
    const long w_row_pitch = w_tap_pitch * taps;

    // per-thread staging sources / swizzled LDS destinations (no predicates: SLAB*4 and the W chunk count are multiples
    // of the block size, or wrap around onto duplicate chunks for BN = 32)
    const char* xsrc[XCH]; int xdst[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + i * 256, r = c >> 2, part = c & 3;
        // the slab buffer always has BM + 64 rows (exact chunk counts, no predicates), but rows beyond the conv's real span
        // re-read the last needed row (a cache hit) instead of pulling 62 unneeded rows per tile from HBM for k = 3
        const int rs = min(r, BM + (p.taps - 1) * p.dil - 1);
        xsrc[i] = reinterpret_cast<const char*>(p.A) + ((long)m0 - (long)p.center * p.dil + rs) * a_pitch + part * 16;
        xdst[i] = swz(r, part);
    }
    const char* wsrc[WCH]; int wdst[WCH];
#pragma unroll
    for (int i = 0; i < WCH; ++i) {
        const int c = (tid + i * 256) % (BN * 4), r = c >> 2, part = c & 3;
        wsrc[i] = reinterpret_cast<const char*>(p.W) + (long)(n0 + r) * w_row_pitch + part * 16;
        wdst[i] = swz(r, part);
    }

    uint4 xr[XCH];
    uint4 wA0, wA1, wB0, wB1;   // scalars, not arrays: hipcc kept uint4 wA[2]/wB[2] in scratch
    wA1 = wB1 = make_uint4(0, 0, 0, 0);
#define EV_GLOAD_X(KC) \
    _Pragma("unroll") for (int i = 0; i < XCH; ++i) xr[i] = *reinterpret_cast<const uint4*>(xsrc[i] + (long)(KC) * 64);
#define EV_SSTORE_X(BUF)                                                                         \
    _Pragma("unroll") for (int i = 0; i < XCH; ++i) {                                            \
        uint4 v_ = xr[i];                                                                        \
        if (p.pro_lrelu) v_ = (ES == 2) ? lrelu_h8(v_, p.pro_slope) : lrelu_f4(v_, p.pro_slope); \
        *reinterpret_cast<uint4*>(Xs + (BUF) * XBUF + xdst[i]) = v_;                             \
    }
#define EV_GLOAD_W(DST, KC, TAP)                                                                                     \
    {                                                                                                                \
        DST##0 = *reinterpret_cast<const uint4*>(wsrc[0] + (long)(TAP) * w_tap_pitch + (long)(KC) * 64);             \
        if constexpr (WCH > 1) DST##1 = *reinterpret_cast<const uint4*>(wsrc[WCH - 1] + (long)(TAP) * w_tap_pitch + (long)(KC) * 64); \
    }
#define EV_SSTORE_W(SRC, BUF)                                                                  \
    {                                                                                          \
        *reinterpret_cast<uint4*>(Ws + (BUF) * WBUF + wdst[0]) = SRC##0;                       \
        if constexpr (WCH > 1) *reinterpret_cast<uint4*>(Ws + (BUF) * WBUF + wdst[WCH - 1]) = SRC##1; \
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    int woff[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) woff[a] = swz(wc * TC + a * 16 + fr, fq);
    const bool prio_ = (p.reserved0 & 1) != 0;

#define EV_COMPUTE(WBUFSEL, KC, TAP)                                                                           \
    {                                                                                                          \
        const int row0_ = wt * TT + fr + (TAP) * p.dil;                                                        \
        const char* Xb_ = Xs + ((KC) & 1) * XBUF + row0_ * 64 + ((fq ^ ((row0_ >> 1) & 3)) << 4);               \
        const char* Wb_ = Ws + (WBUFSEL) * WBUF;                                                               \
        uint4 xf[MT], wf[NT];                                                                                  \
        _Pragma("unroll") for (int b = 0; b < MT; ++b) xf[b] = *reinterpret_cast<const uint4*>(Xb_ + b * 16 * 64); \
        _Pragma("unroll") for (int a = 0; a < NT; ++a) wf[a] = *reinterpret_cast<const uint4*>(Wb_ + woff[a]);  \
        if (prio_) __builtin_amdgcn_s_setprio(1);      /* tuning switch (ConvGemmParams::reserved0 bit 0, tools/bench_gemm.py) */ \
        _Pragma("unroll") for (int a = 0; a < NT; ++a) {                                                       \
            _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                   \
                if constexpr (ES == 2) {                                                                       \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wf[a]),      \
                                                                       *reinterpret_cast<half8*>(&xf[b]), acc[a][b], 0, 0, 0); \
                } else {                                                                                       \
                    const float* wa_ = reinterpret_cast<const float*>(&wf[a]);                                 \
                    const float* xb_ = reinterpret_cast<const float*>(&xf[b]);                                 \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                              \
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa_[e], xb_[e], acc[a][b], 0, 0, 0);  \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
        if (prio_) __builtin_amdgcn_s_setprio(0);                                                              \
    }

    // prologue: X(0), W(step 0) -> LDS; W(step 1) -> registers
    EV_GLOAD_X(0)
    EV_GLOAD_W(wA, 0, 0)
    EV_SSTORE_X(0)
    EV_SSTORE_W(wA, 0)
    {
        const int k1 = (taps > 1) ? 0 : (nkc > 1 ? 1 : 0), t1 = (taps > 1) ? 1 : 0;
        EV_GLOAD_W(wB, k1, t1)
    }
    EV_STAMP()
    __syncthreads();
    EV_STAMP()

    int wsel = 0;                        // LDS weight buffer holding the current step's tile
    int kc = 0, tap = 0;
    const int steps = nkc * taps;
    // One pipeline step.  LD receives W(s+2) (clamped to a valid tile at the very end: loaded, never used); ST holds W(s+1),
    // loaded one step ago, and is written to the idle LDS buffer.  Two textual copies with the register sets swapped give a
    // static A/B parity (no register moves: a move is a "use" and drags the vmcnt wait to the top of the next step).
#define EV_STEP(LD, ST)                                                                    \
    {                                                                                      \
        const bool more_ = kc + 1 < nkc;                                                   \
        int t2_ = tap + 2, k2_ = kc;                                                       \
        if (t2_ >= taps) { t2_ -= taps; k2_ = kc + 1; }                                    \
        if (t2_ >= taps) { t2_ -= taps; k2_ += 1; }                                        \
        if (k2_ >= nkc) { k2_ = kc; t2_ = tap; }                                           \
        EV_GLOAD_W(LD, k2_, t2_)                                                           \
        if (tap == 0 && more_) { EV_GLOAD_X(kc + 1) }                                      \
        EV_COMPUTE(wsel, kc, tap)                                                          \
        EV_SSTORE_W(ST, wsel ^ 1)                                                          \
        if (tap == taps - 1) {                                                             \
            if (more_) { EV_SSTORE_X((kc + 1) & 1) }                                       \
            tap = 0; ++kc;                                                                 \
        } else {                                                                           \
            ++tap;                                                                         \
        }                                                                                  \
        EV_STAMP()                                                                         \
        __syncthreads();                                                                   \
        EV_STAMP()                                                                         \
        wsel ^= 1;                                                                         \
    }
    int s = 0;
    for (; s + 1 < steps; s += 2) {
        EV_STEP(wA, wB)
        EV_STEP(wB, wA)
    }
    if (s < steps) EV_STEP(wA, wB)
#undef EV_STEP
#undef EV_COMPUTE
#undef EV_GLOAD_X
#undef EV_SSTORE_X
#undef EV_GLOAD_W
#undef EV_SSTORE_W

    // all waves passed the barrier that ended the last step: the staging buffers are free for the transpose
#ifdef EV_TRACE
    ConvGemmParams pe = p;
    if (trace_flags & 1) { pe.out16 = nullptr; pe.out32 = nullptr; }
#else
    const ConvGemmParams& pe = p;
#endif
    if constexpr (EPI == EPI_GENERIC) gemm_epilogue_lds<MT, NT>(pe, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC);
    else gemm_epilogue_fast<MT, NT, EPI>(pe, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC EV_TRACE_EPI_ARGS);
#ifdef EV_TRACE
    EV_STAMP()
    if (trace_out && bid % 61 == 0 && bid / 61 < 128) {
        unsigned* o = trace_out + (size_t)((bid / 61) * 4 + wave) * (TRACE_N + 8);
        for (int i = lane; i < TRACE_N; i += 64) o[8 + i] = tr_lds[wave * TRACE_N + i];
        if (lane == 0) {
            o[0] = 0xE7ACE000u + wave; o[1] = bid; o[2] = tr_i;
            o[3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
            o[5] = blockIdx.x;
        }
    }
#endif
}

template <typename TIn, int BM, int BN, int WT, int WC, int EPI>
static void launch_epi(const ConvGemmParams& p, hipStream_t s) {
    const int grid = (p.M / BM) * (p.N / BN);
    size_t lds = 2 * (size_t)(BM + MAX_SPAN) * 64 + 2 * (size_t)BN * 64;
    const size_t epi = 4 * (size_t)(32 * ((BN / WC) * 4 + 16));
    if (epi > lds) lds = epi;
#ifdef EV_TRACE
    lds += 4 * TRACE_N * sizeof(unsigned);
    hipLaunchKernelGGL((conv_gemm_kernel<TIn, BM, BN, WT, WC, EPI>), dim3(grid), dim3(256), lds, s, p, g_trace_ptr, g_trace_flags);
#else
    hipLaunchKernelGGL((conv_gemm_kernel<TIn, BM, BN, WT, WC, EPI>), dim3(grid), dim3(256), lds, s, p);
#endif
}

// which straight-line epilogue instantiation a fp16 launch asks for (EPI_GENERIC: none)
static int fp16_epi_variant(const ConvGemmParams& p) {
    static const bool force_generic = tuning_env("EV_EPI_GENERIC") != nullptr;     // A/B switch for tools/bench_gemm.py
    const bool odd_slope = (p.act == ACT_LRELU && !(p.act_slope >= 0.f && p.act_slope <= 1.f)) ||
                           (p.post_lrelu && !(p.post_slope >= 0.f && p.post_slope <= 1.f));     // max(v, s v) form needs s in [0, 1]
    const bool rare_act = p.act != ACT_NONE && p.act != ACT_LRELU;
    const int omode = (p.out16 && !p.out32) ? EPI_O16 : ((p.out32 && !p.out16 && !p.out32_before_post) ? EPI_O32 : 0);
    const int base = (p.res ? (p.res_dtype == DT_F16 ? EPI_RES16 : EPI_RES32) : 0) | (p.acc32 ? EPI_ACC32 : 0) | (rare_act ? EPI_RARE_ACT : 0) |
                     (p.add16_a ? EPI_ADD16 : 0);
    return p.seq_bias || force_generic || odd_slope ? EPI_GENERIC : (base | omode);
}

// epilogue variant: the fp16 kernel specialises the combinations the frame-rate path uses (plain / fp16 residual / fp16
// residual + MRF accumulate / fp32 residual); anything else (per-utterance bias, the fp32 kernel) takes the generic one
template <typename TIn, int BM, int BN, int WT, int WC>
static void launch_cfg(const ConvGemmParams& p, hipStream_t s) {
    if constexpr (sizeof(TIn) == 2) {
        int e = fp16_epi_variant(p);
#define EV_EPI_CASE(E) case (E): return launch_epi<TIn, BM, BN, WT, WC, (E)>(p, s);
        for (int attempt = 0; attempt < 2; ++attempt) {
            switch (e) {
                EV_EPI_CASE(0) EV_EPI_CASE(EPI_O16) EV_EPI_CASE(EPI_O32)
                EV_EPI_CASE(EPI_RARE_ACT) EV_EPI_CASE(EPI_RARE_ACT | EPI_O16)
                EV_EPI_CASE(EPI_RES16) EV_EPI_CASE(EPI_RES16 | EPI_O16) EV_EPI_CASE(EPI_RES16 | EPI_O32)
                EV_EPI_CASE(EPI_RES16 | EPI_ADD16) EV_EPI_CASE(EPI_RES16 | EPI_ADD16 | EPI_O16)     /* (acc32: generic epilogue) */
                EV_EPI_CASE(EPI_RES32) EV_EPI_CASE(EPI_RES32 | EPI_O32)
                default: break;
            }
            if (e == EPI_GENERIC) break;
            e = (e & (EPI_O16 | EPI_O32)) ? (e & ~(EPI_O16 | EPI_O32)) : EPI_GENERIC;    // drop the static-output flag, then give up
        }
#undef EV_EPI_CASE
    }
    launch_epi<TIn, BM, BN, WT, WC, EPI_GENERIC>(p, s);
}

// =====================================================================================================================
// Phased conv-GEMM (fp16) for the MFMA-bound shapes (N % 128 == 0, 3 / 7 / 11 taps, enough tiles to fill the chip).
// Per-wave timelines of conv_gemm_kernel (tools/trace_gemm.py, round 2) show 1 480 cycles of work + 220 of barrier wait per
// (K-chunk, tap) step against 512 cycles of MFMA: the two 4-wave blocks of a CU are independent, so whether one wave's
// fragment-read / staging / barrier time is covered by its SIMD partner's MFMA burst is left to chance, and every wave needs
// the matrix pipe for a third of its time only.  Here the alternation is built in:
//   * 8 waves per block, two blocks per CU (4 waves per SIMD, <= 128 VGPRs): tile 256 x 128, each wave owns 64 x 64 outputs
//     (16 accumulator tiles, 16 MFMAs against 8 fragment reads per step);
//   * waves 0-3 (one per SIMD) and waves 4-7 (their SIMD partners) run the same code one barrier apart: while one group issues
//     the 16 MFMAs of step s (s_setprio 1) -- and, in their shadow, its share of the staging requests --, the other one reads the
//     fragments of its next step; a barrier ends every phase.  On every SIMD exactly one wave of each resident block wants the
//     matrix pipe at any time;
//   * staging is LDS-DMA only (global_load_lds_dwordx4, issued through inline asm so that hipcc neither drains vmcnt in front
//     of the fragment reads nor sees staging registers): the weight tile of step s + 3 (8 KB, one 1-KB piece per wave) into a
//     ring of four buffers, the activation slab of K-chunk kc + 1 (384 rows x 64 B, three pieces per wave) at the first tap of
//     chunk kc.  The 64-byte-pitch XOR swizzle of conv_gemm_kernel is applied on the SOURCE side (lane l of a 16-row piece fetches
//     part (l & 3) ^ ((l >> 3) & 3) of row l >> 2; the DMA writes lane-linear), the fragment reads are unchanged;
//   * waits are counted, never 0 inside the loop: VMEM returns in order, so "the pieces this wave requested two matrix phases ago
//     have landed" is s_waitcnt vmcnt(#pieces requested in the latest matrix phase) in the load phase, a compile-time number because
//     the tap loop is unrolled (TAPS is a template parameter); the piece is visible to the other waves after the phase's barrier
//     and is first read two barriers later;
//   * leaky-relu of the consumer side (pro_lrelu) cannot be applied in flight any more: every wave fixes up its own three slab
//     pieces in place (ds_read / v_pk_mul + v_pk_max / ds_write) right after the wait that retires them, once per K-chunk;
//   * a buffer is re-targeted by a DMA only after a barrier that follows the retirement of its last reader's fragment reads (they
//     retire in front of that reader's first MFMA, one phase before the earliest request for the buffer is issued).
// EV_PH_SCHED selects the earlier schedules for A/B builds (requests in the load phase, reads retired before the barrier, one barrier
// per step): all bit-identical and within 1 % of each other.
// Same accumulation order per output element as conv_gemm_kernel (K-chunks outer, taps inner, one 32-deep MFMA per step) and the
// same epilogue functions: results are bit-identical to the 4-wave kernel (tests/test_gpu_ops.py), so the batch-invariance
// guarantee is untouched by which kernel a launch takes.
static constexpr int PH_BM = 256, PH_BN = 128, PH_SLABR = 384, PH_NW = 4;
static constexpr int PH_XBUF = PH_SLABR * 64, PH_WBUF = PH_BN * 64;
static constexpr size_t PH_LDS = 2 * (size_t)PH_XBUF + PH_NW * (size_t)PH_WBUF;          // 80 KB: two blocks per CU

// 64-bit products are vector instructions on gfx9, so a wave-uniform pointer that involved one lives in VGPRs (and so does everything
// derived from it); readfirstlane pins it to an SGPR pair (and folds away when the value already is scalar)
__device__ __forceinline__ const char* uniform_ptr(const char* q) {
    const unsigned long long b = reinterpret_cast<unsigned long long>(q);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}
// one LDS-DMA wave-instruction: 64 lanes x 16 B from sbase + voff[lane] to LDS [lds_off + 16 * lane] (M0 carries the LDS address);
// sbase and lds_off are wave-uniform
__device__ __forceinline__ void glds16(const char* sbase, unsigned voff, unsigned lds_off) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(uniform_ptr(sbase)), "s"(__builtin_amdgcn_readfirstlane(lds_off)) : "memory");
}

// accumulate in place.  With the builtin hipcc rotates the 16 accumulator tiles of the unrolled tap loop through fresh registers
// (dst != srcC in the first steps of every K-chunk): 30 registers over the 128 of four waves per SIMD, and every spill reload inside
// the loop costs an s_waitcnt vmcnt(0), i.e. the DMA pipeline.  Consecutive uses of one accumulator are 16 MFMAs apart and the
// fragments were waited for (lgkmcnt(0)) before the phase barrier, so no wait states are owed inside the sequence.
__device__ __forceinline__ void mfma_inplace(f32x4& c, const half8& a, const half8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// Software wait states between the LAST inline-asm MFMA of a sequence and the first VALU / LDS / VMEM instruction that reads (or overwrites) its
// accumulators.  gfx9 has no hardware interlock for "XDL write VGPR -> non-MFMA read": hipcc's hazard recognizer pads MFMAs it emits itself, but cannot see
// into an asm statement.  Found in round 4 by the bit-identity test of the two-group pair kernel: resblock_pair_c32_mx_kernel<3, 1> (round 3's product) read
// conv1's accumulators for the xt quantiser two MFMA issue slots after the last block-scaled MFMA and, depending on how the SIMD's other wave was scheduled,
// sometimes got the value from BEFORE it -- one fp4 cross term (2^-11 of the result, 3e-4 absolute on unit-variance data) missing in a few rows, run to run.
// 20 wait states cover the 16-pass worst case of the gfx940 / gfx950 tables (XDL write -> VALU read: passes + 3).
template <int NA, int NB>
__device__ __forceinline__ void mfma_asm_fence(f32x4 (&acc)[NA][NB]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) asm volatile("" : "+v"(acc[a][b]));          // (every later use of an accumulator is ordered behind the nops)
}

// Tuning builds only (build.py --variant <tag> EV_PH_ABLATE=<bits>): drop one ingredient of the main loop to see what the phases wait
// for -- 1: no MFMAs, 2: no fragment reads, 4: no DMA requests, 8: no phase barriers, 16: no epilogue.  Results are garbage by design.
#ifndef EV_PH_ABLATE
#define EV_PH_ABLATE 0
#endif
// EV_PH_TIMING (tuning builds): every wave accumulates, in SGPRs, the cycles it spends in each part of a step (s_memtime deltas) and lane 0
// writes the seven sums to ((unsigned*)p.row_seq)[(block * 8 + wave) * 8 + k] at the end: k = 0 prologue, 1 vmcnt wait, 2 rest of the load
// phase, 3 barrier after the load phase, 4 matrix phase, 5 barrier after the matrix phase, 6 epilogue (row_seq is unused by these launches).
#ifdef EV_PH_TIMING
#define EV_PH_T0() unsigned long long t_prev_ = __builtin_readcyclecounter(); unsigned t_acc_[7] = {0, 0, 0, 0, 0, 0, 0};
#define EV_PH_TICK(K) { const unsigned long long t_now_ = __builtin_readcyclecounter(); t_acc_[K] += (unsigned)(t_now_ - t_prev_); t_prev_ = t_now_; }
#else
#define EV_PH_T0()
#define EV_PH_TICK(K)
#endif
// schedule switches (A/B builds): bit 0: staging requests issued inside the matrix phase, bit 1: fragment reads retired after the barrier,
// bit 3: one barrier per step (group 0: between load and matrix phase, group 1: after the matrix phase)
#ifndef EV_PH_SCHED
#define EV_PH_SCHED 3
#endif

template <int TAPS, int BN, int EPI>
__device__ __forceinline__ void conv_gemm_phased_body(const ConvGemmParams& p, int bid, const int nblk) {
    // BN = 128: one tap per step, each wave 64 x 64 outputs.  BN = 64 (the C = 64 stage): each wave 64 x 32 outputs and a step covers TWO
    // taps, so that a matrix phase is 16 MFMAs in both cases and the weight tile of a step is 8 KB = one DMA piece per wave in both
    // (BN = 64: [2 taps][64 rows][64 B]; the second tap of the last, odd step is a harmless re-request of the last tap).
    constexpr int BM = PH_BM, XBUF = PH_XBUF, WBUF = PH_WBUF;
    constexpr int TPS = (BN == 64) ? 2 : 1;              // taps per step
    constexpr int U = (TAPS + TPS - 1) / TPS;            // steps per K-chunk
    constexpr int TC = BN / 2, MT = 4, NT = TC / 16;
    constexpr bool DMA_IN_MMA = (EV_PH_SCHED & 1) != 0, LGKM_AFTER = (EV_PH_SCHED & 2) != 0, ONE_BAR = (EV_PH_SCHED & 8) != 0;
    static_assert(!ONE_BAR || DMA_IN_MMA, "one barrier per step needs the staging requests inside the matrix phase");
    static_assert(BN == 128 || BN == 64, "tile width");
    static_assert(!LGKM_AFTER || DMA_IN_MMA, "a buffer may be re-targeted one phase after its last read only if that read was retired before the barrier");
    static_assert(U >= 3 && PH_SLABR >= BM + MAX_SPAN && 8 * 32 * (TC * 4 + 16) <= (int)PH_LDS, "pipeline depth / slab / epilogue scratch");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Xs = smem;                 // [2][SLABR][64]
    char* const Ws = smem + 2 * XBUF;      // [NW][8 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave & 3, wc = wave >> 2;          // wc is also the phase group: waves w and w + 4 share a SIMD
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    EV_PH_T0()
#ifdef EV_PH_TIMING
    const int bid_raw_ = bid;
#endif

    const int nN = p.N / BN;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (bid / nN) * BM, n0 = (bid % nN) * BN;
    const int nkc = p.K >> 5;
    const unsigned a_pitch = (unsigned)p.lda * 2u, w_tap_pitch = (unsigned)p.K * 2u, w_row_pitch = w_tap_pitch * TAPS;

    // DMA sources: uniform 64-bit bases (SGPRs) + one 32-bit lane offset per piece.  Weight piece of wave w: rows 16 w .. 16 w + 15 of the
    // tile (BN = 128) / rows 16 (w & 3) .. of tap 2 u + (w >> 2) (BN = 64)
    const int prow = lane >> 2, ppart = (lane & 3) ^ ((lane >> 3) & 3);
    const int wprow = (BN == 128 ? wave : (wave & 3)) * 16 + prow, wtapj = (BN == 128) ? 0 : (wave >> 2);
    const char* const wbase = uniform_ptr(reinterpret_cast<const char*>(p.W) + (long)n0 * w_row_pitch);
    const unsigned wvoff = __umul24((unsigned)wprow, w_row_pitch) + ppart * 16;
    const char* const xbase = uniform_ptr(reinterpret_cast<const char*>(p.A) + ((long)m0 - (long)p.center * p.dil) * a_pitch);
    unsigned xvoff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        // rows beyond the conv's real span re-read the last needed row (cache hits) instead of pulling unneeded rows from HBM
        const int r = (wave + 8 * i) * 16 + prow, rs = min(r, BM + (TAPS - 1) * p.dil - 1);
        xvoff[i] = __umul24((unsigned)rs, a_pitch) + ppart * 16;         // (24-bit factors: a 32-bit v_mad, not a 64-bit pair)
    }
    const unsigned xdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024), wdst = xdst + 2 * XBUF;
    const bool pro = p.pro_lrelu != 0;
    const float pro_slope = p.pro_slope;

    // weight tile of step US of K-chunk KC -> ring slot SLOT
#define EV_PH_ISSUE_W(KC, US, SLOT)                                                                                \
    if (!(EV_PH_ABLATE & 4)) {                                                                                     \
        const int tp_ = min((US) * TPS + wtapj, TAPS - 1);                                                         \
        glds16(wbase + (unsigned)tp_ * w_tap_pitch + (unsigned)(KC) * 64u, wvoff, wdst + (unsigned)(SLOT) * WBUF);  \
    }
#define EV_PH_ISSUE_X1(KC, BUF, I) if (!(EV_PH_ABLATE & 4)) glds16(xbase + (unsigned)(KC) * 64u, xvoff[I], xdst + (unsigned)(BUF) * XBUF + (I) * 8192);
#define EV_PH_FIXUP(BUF)                                                                                           \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                                \
        uint4* q_ = reinterpret_cast<uint4*>(Xs + (BUF) * XBUF + (wave + 8 * i) * 1024 + lane * 16);              \
        *q_ = lrelu_h8(*q_, pro_slope);                                                                            \
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int woff = swz(wc * TC + fr, fq);

    // prologue: slab 0 and the weight tiles of steps 0..2, drained; fix-up; everybody meets, then group 1 falls one phase behind
    EV_PH_ISSUE_X1(0, 0, 0)
    EV_PH_ISSUE_X1(0, 0, 1)
    EV_PH_ISSUE_X1(0, 0, 2)
    EV_PH_ISSUE_W(0, 0, 0)
    EV_PH_ISSUE_W(0, 1, 1)
    EV_PH_ISSUE_W(0, 2, 2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (pro) { EV_PH_FIXUP(0) }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (!ONE_BAR && wc == 1) __builtin_amdgcn_s_barrier();
    EV_PH_TICK(0)

    for (int kc = 0; kc < nkc; ++kc) {
        const bool more = kc + 1 < nkc;
        const char* const Xb = Xs + (kc & 1) * XBUF;
        // (opaque copy per K-chunk: otherwise hipcc hoists the fragment addresses of all TAPS taps out of this loop -- 2 VGPRs per
        // tap, spilled, and a scratch reload inside the loop drains vmcnt, i.e. the whole DMA pipeline)
        int dil_ = p.dil;
        asm volatile("" : "+s"(dil_));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int s = kc * U + u;
            const int ntap = (u * TPS + TPS <= TAPS) ? TPS : 1;       // taps of this step (static after unrolling)
            // ---------------- load phase: this step's fragments; the staging requests of two steps ago are retired
            int k3 = (u + 3 < U) ? kc : kc + 1, u3 = (u + 3 < U) ? u + 3 : u + 3 - U;
            if (k3 >= nkc) { k3 = nkc - 1; u3 = U - 1; }               // past the end: re-request the last tile (keeps the counts static)
            if constexpr (!DMA_IN_MMA) {
                EV_PH_ISSUE_W(k3, u3, (s + 3) & 3)
                if (u == 0 && more) { EV_PH_ISSUE_X1(kc + 1, (kc + 1) & 1, 0) EV_PH_ISSUE_X1(kc + 1, (kc + 1) & 1, 1) EV_PH_ISSUE_X1(kc + 1, (kc + 1) & 1, 2) }
            }
            // VMEM returns in order: "the pieces this wave requested for step s + 1 (and the slab, at step 2) have landed" = at most the
            // requests issued after them are outstanding
#define EV_PH_WAIT_VM                                                                                                   \
            if constexpr (DMA_IN_MMA) { if (u == 1 && more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); } \
            else { if (u <= 1 && more) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
            if (u == 2) {                         // (fix-up before the fragment reads: its 24 transient registers and theirs never coexist)
                EV_PH_TICK(2)
                EV_PH_WAIT_VM
                EV_PH_TICK(1)
                if (more && pro) {
                    EV_PH_FIXUP((kc + 1) & 1)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            uint4 xf[TPS][MT], wf[TPS][NT];
#pragma unroll
            for (int j = 0; j < TPS; ++j) {
                if (j >= ntap) continue;
                const int row0 = wt * 64 + fr + (u * TPS + j) * dil_;
                const char* xp = Xb + row0 * 64 + ((fq ^ ((row0 >> 1) & 3)) << 4);
                const char* wp = Ws + (s & 3) * WBUF + j * 4096 + woff;
                if constexpr ((EV_PH_ABLATE & 2) == 0) {
#pragma unroll
                    for (int a = 0; a < NT; ++a) wf[j][a] = *reinterpret_cast<const uint4*>(wp + a * 1024);
#pragma unroll
                    for (int b = 0; b < MT; ++b) xf[j][b] = *reinterpret_cast<const uint4*>(xp + b * 1024);
                } else {
#pragma unroll
                    for (int a = 0; a < NT; ++a) { wf[j][a] = make_uint4(s, lane, a, u); asm volatile("" : "+v"(wf[j][a].x), "+v"(wf[j][a].y), "+v"(wf[j][a].z), "+v"(wf[j][a].w)); }
#pragma unroll
                    for (int b = 0; b < MT; ++b) { xf[j][b] = make_uint4(s, lane, b, u); asm volatile("" : "+v"(xf[j][b].x), "+v"(xf[j][b].y), "+v"(xf[j][b].z), "+v"(xf[j][b].w)); }
                    asm volatile("" :: "v"(xp), "v"(wp));
                }
            }
            if (u != 2) { EV_PH_TICK(2) EV_PH_WAIT_VM EV_PH_TICK(1) }
#undef EV_PH_WAIT_VM
            if constexpr (!LGKM_AFTER) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            EV_PH_TICK(2)
            if constexpr (ONE_BAR) {
                // ONE barrier per step: group 0 between its load and matrix phase, group 1 after its matrix phase.  Between two barriers
                // group 0 runs [matrix(s-1), load(s)] and group 1 [load(s), matrix(s)]: the alternation without the barrier after the
                // matrix phase.  Group 0 retires its fragment reads before the barrier (its buffers are re-targeted by group 1's
                // requests inside the interval that follows).
                if (wc == 0) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (!(EV_PH_ABLATE & 8)) __builtin_amdgcn_s_barrier();
                }
            } else {
                if (!(EV_PH_ABLATE & 8)) __builtin_amdgcn_s_barrier();
            }
            EV_PH_TICK(3)
            // (LGKM_AFTER: no explicit wait here -- hipcc's own counted lgkmcnt in front of each MFMA's first use retires the reads)
            __builtin_amdgcn_sched_barrier(0);
            // ---------------- matrix phase (+ the staging requests, in the shadow of the MFMAs)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < TPS; ++j) {
                if (j >= ntap) continue;
#pragma unroll
                for (int a = 0; a < NT; ++a) {
#pragma unroll
                    for (int b = 0; b < MT; ++b) {
                        if constexpr ((EV_PH_ABLATE & 1) == 0) mfma_inplace(acc[a][b], *reinterpret_cast<half8*>(&wf[j][a]), *reinterpret_cast<half8*>(&xf[j][b]));
                        else asm volatile("" : "+v"(acc[a][b]) : "v"(*reinterpret_cast<half8*>(&wf[j][a])), "v"(*reinterpret_cast<half8*>(&xf[j][b])));
                        if constexpr (DMA_IN_MMA) {
                            const int idx = (j * NT + a) * MT + b;            // position in the phase's MFMA sequence (8 or 16 long)
                            if (idx == 2) { EV_PH_ISSUE_W(k3, u3, (s + 3) & 3) }
                            if (u == 0 && more) {
                                if (idx == 4) { EV_PH_ISSUE_X1(kc + 1, (kc + 1) & 1, 0) }
                                if (idx == 5) { EV_PH_ISSUE_X1(kc + 1, (kc + 1) & 1, 1) }
                                if (idx == 6) { EV_PH_ISSUE_X1(kc + 1, (kc + 1) & 1, 2) }
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            EV_PH_TICK(4)
            if constexpr (ONE_BAR) { if (wc == 1 && !(EV_PH_ABLATE & 8)) __builtin_amdgcn_s_barrier(); }
            else { if (!(EV_PH_ABLATE & 8)) __builtin_amdgcn_s_barrier(); }
            EV_PH_TICK(5)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef EV_PH_ISSUE_W
#undef EV_PH_ISSUE_X1
#undef EV_PH_FIXUP
    // drain the re-requested tail tiles, let group 1 catch up, then everybody may overwrite the staging buffers
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!ONE_BAR && wc == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    mfma_asm_fence(acc);
    EV_PH_TICK(5)
    if constexpr ((EV_PH_ABLATE & 16) == 0) {
        EV_TRACE_EPI_DUMMY
        gemm_epilogue_fast<MT, NT, EPI>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * 64, n0 + wc * TC EV_TRACE_EPI_ARGS);
    } else {
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) asm volatile("" :: "v"(acc[a][b]));
    }
#ifdef EV_PH_TIMING
    EV_PH_TICK(6)
    if (p.row_seq && !p.seq_bias && lane == 0) {
        unsigned* o_ = reinterpret_cast<unsigned*>(const_cast<int32_t*>(p.row_seq)) + ((size_t)bid_raw_ * 8 + wave) * 8;
#pragma unroll
        for (int k = 0; k < 7; ++k) o_[k] = t_acc_[k];
        o_[7] = 0xC0FFEE;
    }
#endif
}

template <int TAPS, int BN, int EPI>
__global__ __launch_bounds__(512, 4) void conv_gemm_phased_kernel(const ConvGemmParams p) {
    conv_gemm_phased_body<TAPS, BN, EPI>(p, blockIdx.x, gridDim.x);
}

template <int TAPS, int BN, int EPI>
static void launch_phased_epi(const ConvGemmParams& p, hipStream_t s) {
    const int grid = (p.M / PH_BM) * (p.N / BN);
    hipLaunchKernelGGL((conv_gemm_phased_kernel<TAPS, BN, EPI>), dim3(grid), dim3(512), PH_LDS, s, p);
}
template <int TAPS, int BN>
static bool launch_phased_taps(const ConvGemmParams& p, int e, hipStream_t s) {
    switch (e) {
        case EPI_O16: launch_phased_epi<TAPS, BN, EPI_O16>(p, s); return true;
        case EPI_RES16 | EPI_O16: launch_phased_epi<TAPS, BN, EPI_RES16 | EPI_O16>(p, s); return true;
        case EPI_RES16 | EPI_ADD16 | EPI_O16: launch_phased_epi<TAPS, BN, EPI_RES16 | EPI_ADD16 | EPI_O16 | EPI_LEAN>(p, s); return true;
        case EPI_RARE_ACT | EPI_O16:
            if constexpr (BN == 128) { launch_phased_epi<TAPS, BN, EPI_RARE_ACT | EPI_O16>(p, s); return true; }
            return false;
        case EPI_RES32 | EPI_O32:               // fp32 residual stream of the mel decoder (conv-FFN's second conv)
            if constexpr (BN == 128 && TAPS == 3) { launch_phased_epi<TAPS, BN, EPI_RES32 | EPI_O32 | EPI_LEAN>(p, s); return true; }
            return false;
        default: return false;
    }
}
// true if the launch was taken: fp16; N % 128 == 0 with 3 / 7 / 11 taps, or N % 64 == 0 with 11 taps; one of the five epilogue
// variants of the frame-rate path
static bool launch_phased(const ConvGemmParams& p, int e, hipStream_t s) {
    if (p.M % PH_BM != 0 || p.K % 32 != 0 || (p.taps - 1) * p.dil > MAX_SPAN) return false;
    if (p.N % 128 == 0) {
        switch (p.taps) {
            case 3: return launch_phased_taps<3, 128>(p, e, s);
            case 7: return launch_phased_taps<7, 128>(p, e, s);
            case 11: return launch_phased_taps<11, 128>(p, e, s);
            default: return false;
        }
    }
    if (p.N % 64 == 0) {
        // measured (tools/bench_gemm.py --dbg 4,0): k = 11 +0...10 %, k = 7 -8 % (4.4 TB/s: HBM-bound, the 4-wave kernel's 3 blocks per CU hide
        // the latency better) -- only the 11-tap layers take it
        if (p.taps == 11) return launch_phased_taps<11, 64>(p, e, s);
        return false;
    }
    return false;
}
template <int TAPS, int BN>
static hipError_t phased_attr_taps() {
    hipError_t e = hipSuccess, r;
    r = hipFuncSetAttribute((const void*)conv_gemm_phased_kernel<TAPS, BN, EPI_O16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_gemm_phased_kernel<TAPS, BN, EPI_RES16 | EPI_O16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_gemm_phased_kernel<TAPS, BN, EPI_RES16 | EPI_ADD16 | EPI_O16 | EPI_LEAN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    if constexpr (BN == 128) {
        r = hipFuncSetAttribute((const void*)conv_gemm_phased_kernel<TAPS, BN, EPI_RARE_ACT | EPI_O16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    }
    if constexpr (BN == 128 && TAPS == 3) {
        r = hipFuncSetAttribute((const void*)conv_gemm_phased_kernel<TAPS, BN, EPI_RES32 | EPI_O32 | EPI_LEAN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    }
    return e;
}
static hipError_t phased_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = phased_attr_taps<3, 128>(); if (r != hipSuccess) e = r;
    r = phased_attr_taps<7, 128>(); if (r != hipSuccess) e = r;
    r = phased_attr_taps<11, 128>(); if (r != hipSuccess) e = r;
    r = phased_attr_taps<11, 64>(); if (r != hipSuccess) e = r;
    return e;
}

template <typename TIn>
static void launch_dt(const ConvGemmParams& p_in, hipStream_t s) {
    // s_setprio(1) around a step's MFMA cluster: the two blocks of a CU are not in lockstep, and raising the priority of the wave
    // that is in its matrix phase over its SIMD partner's staging / LDS phase measured +4 % on the C = 256 layers (s0 k = 11:
    // 998 -> 1039 TF/s), +1-4 % on the up-convs and decoder GEMMs, nothing at C = 128 / k >= 7 and -1...-4 % on the HBM-bound
    // C = 128 / k = 3 layers (tools/bench_gemm.py --dbg 0,1,0,1, profiles/r2_e_gemm_setprio.txt): on for N >= 256 only.
    ConvGemmParams p = p_in;
    static const char* prio_env = tuning_env("EV_GEMM_PRIO");            // "0" / "1": A/B override
    if (prio_env ? prio_env[0] == '1' : p.N >= 256) p.reserved0 |= 1;
    if (p.N % 128 == 0) {
        // 256-row tiles halve the weight-tile traffic per FLOP and the barrier count per MFMA; since the epilogue stopped scaling
        // with vmcnt round trips they win on every shape measured (tools/bench_gemm.py, EV_GEMM_TILE=128/256), so 128-row
        // tiles are only used to fill the 256 CUs when there are few tiles.  The fp32 kernel is MFMA-rate bound (1/16 of fp16),
        // weight traffic is irrelevant there, so it only takes the big tile when there are plenty of them.
        static const char* force = tuning_env("EV_GEMM_TILE");            // "128" / "256": A/B switch for tools/bench_gemm.py
        static const char* smallm = tuning_env("EV_GEMM_SMALLM");         // "0" / "1": A/B switch for the latency configuration below
        const long tiles256 = (long)(p.M / 256) * (p.N / 128);
        // Latency configuration (single utterances: B = 1 is the reference's own call pattern): with a handful of 128-wide tiles most
        // CUs idle while each tile walks its whole (K-chunk, tap) sequence -- 88 steps = ~70 us for one C = 256 / k = 11 conv of a
        // 256-phoneme utterance, 169 such dependent launches per forward.  256 x 32 tiles give 4x the blocks with a quarter of the
        // MFMAs per step.  Same accumulation order per output element, same epilogue: bit-identical results (batch-invariance tests).
        const bool latency_cfg = sizeof(TIn) == 2 && (smallm ? smallm[0] == '1' : true) && tiles256 < 64;
        if (latency_cfg && !force) return launch_cfg<TIn, 256, 32, 4, 1>(p, s);
        if constexpr (sizeof(TIn) == 2) {
            // phased 8-wave kernel for the shapes that fill the chip with 256 x 128 tiles (bit-identical results, see its header)
            static const char* ph_env = tuning_env("EV_GEMM_PHASED");       // "0" / "1": A/B switch
            const bool ph_on = (ph_env ? ph_env[0] == '1' : true) && !(p.reserved0 & 4);      // reserved0 bit 2: in-process A/B (tools/bench_gemm.py --dbg)
            if (ph_on && !force && tiles256 >= 256 && launch_phased(p, fp16_epi_variant(p), s)) return;
        }
        bool big = tiles256 >= (sizeof(TIn) == 4 ? 2048 : 256);
        if (force) big = force[0] == '2';
        if (big) launch_cfg<TIn, 256, 128, 2, 2>(p, s);
        else launch_cfg<TIn, 128, 128, 2, 2>(p, s);
    } else if (p.N % 64 == 0) {
        if constexpr (sizeof(TIn) == 2) {          // C = 64 stage, 7 / 11 taps: phased kernel with two taps per step
            static const char* ph_env = tuning_env("EV_GEMM_PHASED");
            const bool ph_on = (ph_env ? ph_env[0] == '1' : true) && !(p.reserved0 & 4);
            if (ph_on && (long)(p.M / 256) * (p.N / 64) >= 512 && launch_phased(p, fp16_epi_variant(p), s)) return;
        }
        launch_cfg<TIn, 256, 64, 4, 1>(p, s);
    } else launch_cfg<TIn, 256, 32, 4, 1>(p, s);
}

// =====================================================================================================================
// Split-precision GEMM for the fp32 token-rate path (text encoder, embed_projection1, variance / duration predictors).
// The fp32-input MFMA runs at 1/16 of the fp16 rate; instead every fp32 operand is written as
//        x = x_hi + 2^-11 * x_lo,   x_hi = fp16(x),  x_lo = fp16((x - x_hi) * 2^11)        (weights: split once by the packer)
// and  x*w = x_hi*w_hi + 2^-11 (x_hi*w_lo + x_lo*w_hi) + O(2^-22 |x w|):  three fp16 MFMAs (fp16 products are exact in fp32,
// accumulation is fp32) into two accumulators.  The dropped term and the representation error are 2^-22 relative, the same
// order as fp32 rounding itself, so the duration path keeps its fp32-level agreement with the reference (tests hold the
// token-rate taps to 1e-4 and the durations bit-exact) at ~4x the speed.  The 2^11 scaling keeps the low parts out of fp16's
// subnormal range.  Activations are split while staging (fp32 global -> two fp16 LDS slabs).
// Tile 128 x 64, 4 waves (64 x 32 per wave), one K step = 32 elements, register prefetch of the next step's tiles.
template <int BM, int BN, int EPI>
__global__ __launch_bounds__(256, 2) void conv_gemm_split_kernel(const ConvGemmParams p) {
    constexpr int TT = 64, TC = BN / 2, MT = 4, NT = TC / 16;
    constexpr int SLAB = BM + MAX_SPAN, XCH = SLAB * 4 / 256;
    constexpr int XBUF = SLAB * 64, WBUF = BN * 64;
    static_assert(BM == 128 && (BN == 64 || BN == 32) && XCH == 3, "tile shape");      // BN = 32: the C = 32 layers of the generator's last stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xh = smem;                    // [2][SLAB][64]  hi parts
    char* Xl = Xh + 2 * XBUF;           // [2][SLAB][64]  lo parts
    char* Wh = Xl + 2 * XBUF;           // [2][BN][64]
    char* Wl = Wh + 2 * WBUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave >> 1, wc = wave & 1;
    const int nN = p.N / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    // split-K (ConvGemmParams::ksplit, launch_split): the grid is tiles x ksplit, the ranges of a tile are neighbours in the XCD-contiguous order (one L2
    // holds the slab they share), block (tile, ks) walks the K-chunks [kc0, kc1) and writes its partial sums to the ks-th [M][N] slice behind out32
    const int nks = p.ksplit > 1 ? p.ksplit : 1;
    const int ks = bid % nks;
    bid /= nks;
    const int m0 = (bid / nN) * BM, n0 = (bid % nN) * BN;
    const int taps = p.taps, nkc_all = p.K >> 5, kc0 = ks * (nkc_all / nks), nkc = nkc_all / nks, steps = nkc * taps;
    const long a_pitch = (long)p.lda * 4;
    const long w_tap_pitch = (long)p.K * 2, w_row_pitch = w_tap_pitch * taps;

    // staging: chunk c -> slab row c >> 2, 16-byte fp16 part c & 3  <-  32 bytes (8 floats) of the fp32 row
    const char* xsrc[XCH]; int xdst[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + i * 256, r = c >> 2, part = c & 3;
        const int rs = min(r, BM + (p.taps - 1) * p.dil - 1);
        xsrc[i] = reinterpret_cast<const char*>(p.A) + ((long)m0 - (long)p.center * p.dil + rs) * a_pitch + part * 32;
        xdst[i] = swz(r, part);
    }
    const int wr_ = (tid >> 2) % BN, wp_ = tid & 3;      // BN = 32: threads 128-255 duplicate the chunks of threads 0-127 (no predicate)
    const long wsrc_off = (long)(n0 + wr_) * w_row_pitch + wp_ * 16;
    const int wdst = swz(wr_, wp_);
    const char* Whg = reinterpret_cast<const char*>(p.W);
    const char* Wlg = reinterpret_cast<const char*>(p.W_lo);

    float4 xa0, xb0, xa1, xb1, xa2, xb2;      // 3 chunks x 8 floats
    uint4 whr, wlr;
#define EV_S_GLOAD_X(KC)                                                                         \
    {                                                                                            \
        xa0 = *reinterpret_cast<const float4*>(xsrc[0] + (long)(KC) * 128);                      \
        xb0 = *reinterpret_cast<const float4*>(xsrc[0] + (long)(KC) * 128 + 16);                 \
        xa1 = *reinterpret_cast<const float4*>(xsrc[1] + (long)(KC) * 128);                      \
        xb1 = *reinterpret_cast<const float4*>(xsrc[1] + (long)(KC) * 128 + 16);                 \
        xa2 = *reinterpret_cast<const float4*>(xsrc[2] + (long)(KC) * 128);                      \
        xb2 = *reinterpret_cast<const float4*>(xsrc[2] + (long)(KC) * 128 + 16);                 \
    }
#define EV_S_SPLIT_STORE(XA, XB, DST, BUF)                                                       \
    {                                                                                            \
        float f_[8] = {XA.x, XA.y, XA.z, XA.w, XB.x, XB.y, XB.z, XB.w};                          \
        if (p.pro_lrelu) {          /* leaky-relu on the fp32 value, BEFORE the split (slope in [0, 1]: checked by the launcher) */ \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) f_[e] = fmaxf(f_[e], f_[e] * p.pro_slope); \
        }                                                                                        \
        half8 hi_, lo_;                                                                          \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                          \
            const _Float16 h_ = (_Float16)f_[e];                                                 \
            hi_[e] = h_;                                                                         \
            lo_[e] = (_Float16)((f_[e] - (float)h_) * 2048.0f);                                  \
        }                                                                                        \
        *reinterpret_cast<half8*>(Xh + (BUF) * XBUF + (DST)) = hi_;                              \
        *reinterpret_cast<half8*>(Xl + (BUF) * XBUF + (DST)) = lo_;                              \
    }
#define EV_S_SSTORE_X(BUF)                                                                       \
    {                                                                                            \
        EV_S_SPLIT_STORE(xa0, xb0, xdst[0], BUF)                                                 \
        EV_S_SPLIT_STORE(xa1, xb1, xdst[1], BUF)                                                 \
        EV_S_SPLIT_STORE(xa2, xb2, xdst[2], BUF)                                                 \
    }
#define EV_S_GLOAD_W(KC, TAP)                                                                    \
    {                                                                                            \
        const long o_ = wsrc_off + (long)(TAP) * w_tap_pitch + (long)(KC) * 64;                  \
        whr = *reinterpret_cast<const uint4*>(Whg + o_);                                         \
        wlr = *reinterpret_cast<const uint4*>(Wlg + o_);                                         \
    }
#define EV_S_SSTORE_W(BUF)                                                                       \
    {                                                                                            \
        *reinterpret_cast<uint4*>(Wh + (BUF) * WBUF + wdst) = whr;                               \
        *reinterpret_cast<uint4*>(Wl + (BUF) * WBUF + wdst) = wlr;                               \
    }

    f32x4 acc[NT][MT], accl[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) { acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int fr = lane & 15, fq = lane >> 4;
    int woff[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) woff[a] = swz(wc * TC + a * 16 + fr, fq);

    // (round 6) Prefetch distance.  With the next step's tiles requested at the top of a step and written to LDS at its bottom, a step of the token-rate GEMMs (24 MFMAs per
    // wave, two 4-wave blocks per CU) waited for an L2 round trip every time: 1.36 us per step against 0.2 us of matrix work, MFMA-busy 0.23.  Now the WEIGHT tile of step s + 2 is
    // requested while step s computes (a second register pair; the one written at the bottom of step s was requested during step s - 1), and the activation slab of the next
    // K-chunk is requested at the chunk's FIRST tap instead of its last (taps >= 2; it is still written behind the last tap).  Same products in the same order: bit-identical.
    uint4 whr2 = uint4{0u, 0u, 0u, 0u}, wlr2 = uint4{0u, 0u, 0u, 0u};
    EV_S_GLOAD_X(kc0)
    EV_S_GLOAD_W(kc0, 0)
    EV_S_SSTORE_X(0)
    EV_S_SSTORE_W(0)
    __syncthreads();
    if (steps > 1) {                                   // W of step 1 (tap 1 of chunk 0, or tap 0 of chunk 1 for a one-tap GEMM)
        if (taps > 1) { EV_S_GLOAD_W(kc0, 1) } else { EV_S_GLOAD_W(kc0 + 1, 0) }
    }
    int kc = 0, tap = 0, wsel = 0;          // kc counts from the range's first chunk (it also picks the slab buffer); kc0 + kc is the chunk
    // one step; (SH, SL): the register pair written to LDS at the bottom (W of step s + 1), (LH, LL): the pair requested at the top (W of step s + 2).
    // STATIC = the three-tap schedule below: every request is unconditional (clamped to the range's last tile / chunk; what is written to LDS behind the last step is never
    // read), so the body is straight-line code and hipcc keeps COUNTED vmcnt waits -- with a run-time condition around a request it falls back to vmcnt(0) at every merge.
#define EV_S_STEP(SH, SL, LH, LL, STATIC, LOADX, STOREX)                                                                         \
    {                                                                                                                            \
        int t1 = tap + 1, k1 = kc;                                                                                               \
        if (t1 == taps) { t1 = 0; k1 = kc + 1; }                                                                                 \
        int t2 = t1 + 1, k2 = k1;                                                                                                \
        if (t2 == taps) { t2 = 0; k2 = k1 + 1; }                                                                                 \
        const bool has_next = s + 1 < steps;                                                                                     \
        const bool store_x = (STATIC) ? (STOREX) : (has_next && t1 == 0);                                                        \
        const bool load_x = (STATIC) ? (LOADX) : (taps > 1 ? (tap == 0 && kc + 1 < nkc) : has_next);                             \
        if ((STATIC) || s + 2 < steps) {                                                                                         \
            const bool in_ = s + 2 < steps;                                                                                      \
            const long o_ = wsrc_off + (long)(in_ ? t2 : taps - 1) * w_tap_pitch + (long)(kc0 + (in_ ? k2 : nkc - 1)) * 64;      \
            LH = *reinterpret_cast<const uint4*>(Whg + o_);                                                                      \
            LL = *reinterpret_cast<const uint4*>(Wlg + o_);                                                                      \
        }                                                                                                                        \
        if (load_x) EV_S_GLOAD_X(kc0 + min(kc + 1, nkc - 1))                                                                     \
        {                                                                                                                        \
            const int row0 = wt * TT + fr + tap * p.dil;                                                                         \
            const int xo = (kc & 1) * XBUF + row0 * 64 + ((fq ^ ((row0 >> 1) & 3)) << 4);                                        \
            uint4 xh[MT], xl[MT], wh[NT], wl[NT];                                                                                \
            _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                                     \
                xh[b] = *reinterpret_cast<const uint4*>(Xh + xo + b * 16 * 64);                                                  \
                xl[b] = *reinterpret_cast<const uint4*>(Xl + xo + b * 16 * 64);                                                  \
            }                                                                                                                    \
            _Pragma("unroll") for (int a = 0; a < NT; ++a) {                                                                     \
                wh[a] = *reinterpret_cast<const uint4*>(Wh + wsel * WBUF + woff[a]);                                             \
                wl[a] = *reinterpret_cast<const uint4*>(Wl + wsel * WBUF + woff[a]);                                             \
            }                                                                                                                    \
            _Pragma("unroll") for (int a = 0; a < NT; ++a)                                                                       \
                _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                                 \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wh[a]), *reinterpret_cast<half8*>(&xh[b]), acc[a][b], 0, 0, 0); \
                    accl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wh[a]), *reinterpret_cast<half8*>(&xl[b]), accl[a][b], 0, 0, 0); \
                    accl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wl[a]), *reinterpret_cast<half8*>(&xh[b]), accl[a][b], 0, 0, 0); \
                }                                                                                                                \
        }                                                                                                                        \
        if ((STATIC) || has_next) {                                                                                              \
            *reinterpret_cast<uint4*>(Wh + (wsel ^ 1) * WBUF + wdst) = SH;                                                       \
            *reinterpret_cast<uint4*>(Wl + (wsel ^ 1) * WBUF + wdst) = SL;                                                       \
        }                                                                                                                        \
        if (store_x) EV_S_SSTORE_X(k1 & 1)                                                                                       \
        __syncthreads();                                                                                                         \
        wsel ^= 1;                                                                                                               \
        tap = t1;                                                                                                                \
        kc = k1;                                                                                                                 \
    }
    if (taps == 3 && (nkc & 1) == 0) {
        // three taps, an even number of K-chunks (every token-rate conv: K = 384 or a 384-wide split-K range): two chunks = six steps per iteration, roles static
        for (int s = 0; s < steps; s += 6) {
            EV_S_STEP(whr, wlr, whr2, wlr2, true, true, false)   ++s;
            EV_S_STEP(whr2, wlr2, whr, wlr, true, false, false)  ++s;
            EV_S_STEP(whr, wlr, whr2, wlr2, true, false, true)   ++s;
            EV_S_STEP(whr2, wlr2, whr, wlr, true, true, false)   ++s;
            EV_S_STEP(whr, wlr, whr2, wlr2, true, false, false)  ++s;
            EV_S_STEP(whr2, wlr2, whr, wlr, true, false, true)   s -= 5;
        }
    } else {
        for (int s = 0; s < steps; ++s) {
            EV_S_STEP(whr, wlr, whr2, wlr2, false, false, false)
            if (++s >= steps) break;
            EV_S_STEP(whr2, wlr2, whr, wlr, false, false, false)
        }
    }
#undef EV_S_STEP
#undef EV_S_GLOAD_X
#undef EV_S_SPLIT_STORE
#undef EV_S_SSTORE_X
#undef EV_S_GLOAD_W
#undef EV_S_SSTORE_W
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] += accl[a][b] * (1.0f / 2048.0f);
#ifdef EV_TRACE
    gemm_epilogue_lds<MT, NT>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC);
#else
    if constexpr (EPI == EPI_GENERIC) gemm_epilogue_lds<MT, NT>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC);
    else if constexpr (EPI == EPI_O32) {
        if (nks > 1) {          // (launch_split hands a split-K launch to this instantiation with a bare epilogue: out32 = the partial-sum slices, ldo = N)
            ConvGemmParams pe = p;
            pe.out32 = p.out32 + (size_t)ks * (size_t)p.M * (size_t)p.N;
            EV_TRACE_EPI_DUMMY
            gemm_epilogue_fast<MT, NT, EPI>(pe, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC EV_TRACE_EPI_ARGS);
        } else {
            EV_TRACE_EPI_DUMMY
            gemm_epilogue_fast<MT, NT, EPI>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC EV_TRACE_EPI_ARGS);
        }
    } else {
        EV_TRACE_EPI_DUMMY
        gemm_epilogue_fast<MT, NT, EPI>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC EV_TRACE_EPI_ARGS);
    }
#endif
}

// second half of a split-K launch: out = epilogue( sum over the ksplit partial-sum slices, in range order ), the formula and the order of operations of
// gemm_epilogue_lds.  One thread = 8 consecutive channels of a row.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvGemmParams p, const float* __restrict__ ws) {
    const int lpr = p.N >> 3;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long t = gid / lpr;
    if (t >= p.M) return;
    const int co = (int)(gid - t * lpr) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool valid = p.row_valid ? (p.row_valid[t >> p.valid_shift] != 0) : true;
    if (valid) {
        const size_t slice = (size_t)p.M * (size_t)p.N;
        const float* wp = ws + (size_t)t * p.N + co;
        for (int s = 0; s < p.ksplit; ++s) {
            const float4 a0 = *reinterpret_cast<const float4*>(wp + s * slice), a1 = *reinterpret_cast<const float4*>(wp + s * slice + 4);
            v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
        }
        if (p.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(p.bias + co), b1 = *reinterpret_cast<const float4*>(p.bias + co + 4);
            v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
        }
        if (p.act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = apply_act(v[e], p.act, p.act_slope);
        }
        if (p.seq_bias) {
            const float* sb = p.seq_bias + (long)p.row_seq[t] * p.ld_seq_bias + co;
            const float4 s0 = *reinterpret_cast<const float4*>(sb), s1 = *reinterpret_cast<const float4*>(sb + 4);
            v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
        }
        if (p.res) {
            if (p.res_dtype == DT_F16) {
                const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(p.res) + t * p.ldres + co);
                const __half2* h = reinterpret_cast<const __half2*>(&r);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float2 f = __half22float2(h[e]); v[2 * e] += f.x; v[2 * e + 1] += f.y; }
            } else {
                const float* rp = reinterpret_cast<const float*>(p.res) + t * p.ldres + co;
                const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= p.out_scale;
        if (p.acc32) {
            const float* rp = p.acc32 + t * p.ldacc + co;
            const float4 r0 = *reinterpret_cast<const float4*>(rp), r1 = *reinterpret_cast<const float4*>(rp + 4);
            v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
        }
    }
    if (p.out32 && p.out32_before_post) {
        float* op = p.out32 + t * p.ldo + co;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (p.post_lrelu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.post_slope;
    }
    if (p.out32 && !p.out32_before_post) {
        float* op = p.out32 + t * p.ldo + co;
        *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(op + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (p.out16) {
        uint4 o;
        __half2* h = reinterpret_cast<__half2*>(&o);
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = __floats2half2_rn(v[2 * e], v[2 * e + 1]);
        *reinterpret_cast<uint4*>(reinterpret_cast<__half*>(p.out16) + t * p.ldo + co) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-precision conv-GEMM, second generation: the kernel of the EV_PREC_X3 frame-rate path (mel decoder GEMMs and every
// HiFi-GAN conv in "strict" mode) and of the token-rate path.  Same arithmetic as conv_gemm_split_kernel above (which stays as
// the A/B reference, EV_X3_OLD=1), restructured for the MFMA-bound regime that three MFMAs per product create:
//   * 512 threads = 8 waves on one CU (2 per SIMD), tile 256 rows x BN, wave tile (256 / WM) x (BN / WN): with BN = 128 a wave
//     owns 64 x 64 outputs = 16 accumulator tiles x 2 (hi*hi and the 2^11-scaled cross terms) = 128 accumulator registers,
//     and a (K-chunk, tap) step is 48 MFMAs per wave against 16 fragment reads (3 MFMAs per ds_read_b128; the fp16 kernel's
//     256 x 128 tile has 2.7);
//   * the pipeline of the fp16 kernel: the fp32 activation slab of K-chunk kc+1 is requested at the first tap of chunk kc and
//     split into its hi / lo fp16 planes when it is written to LDS at the last tap; the weight tiles (hi and lo planes) of
//     step s+2 are requested while step s computes and written one step later; static A/B register parity, branch-free
//     steady state, one barrier per step;
//   * leaky-relu of the consumer side (models/hifigan/models.py:51,118) is applied to the fp32 value while staging, before
//     the split; epilogues are the straight-line LDS-transposed ones of the fp16 kernel.
// LDS: 2 buffers x 2 planes x (384 x 64 B) for the slab (the last 64 rows only pad the chunk count to 3 per thread) +
// 2 x 2 x (BN x 64 B) for the weights = 128 KB at BN = 128: one block per CU.
template <int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(512, 1) void conv_gemm_x3_kernel(const ConvGemmParams p) {
    constexpr int BM = 256;
    constexpr int TT = BM / WM, TC = BN / WN, MT = TT / 16, NT = TC / 16;
    constexpr int SLABR = 384, XCH = 3;                 // staged rows (>= BM + MAX_SPAN), 32-byte source chunks per thread
    constexpr int XBUF = SLABR * 64, WBUF = BN * 64;
    static_assert(WM * WN == 8 && SLABR * 4 == XCH * 512 && SLABR >= BM + MAX_SPAN, "tile shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // a single K-chunk (K = 32: the C = 32 layers of the generator's last stage) needs one slab buffer, not two: 57 KB instead of
    // 106 KB at BN = 32, i.e. two blocks per CU, so that one block's slab load (the whole of its HBM latency: there is no next
    // chunk to prefetch) overlaps the other's taps.  The launcher sizes the dynamic LDS the same way (x3_lds_bytes).
    const int nxb = (p.K > 32) ? 2 : 1;
    char* Xh = smem;                    // [nxb][SLABR][64]  hi parts
    char* Xl = Xh + nxb * XBUF;         // [nxb][SLABR][64]  lo parts (scaled by 2^11)
    char* Wh = Xl + nxb * XBUF;         // [2][BN][64]
    char* Wl = Wh + 2 * WBUF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave / WN, wc = wave % WN;
    const int nN = p.N / BN;
    const int nblk = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (bid / nN) * BM, n0 = (bid % nN) * BN;
    const int taps = p.taps, nkc = p.K >> 5, steps = nkc * taps;
    const long a_pitch = (long)p.lda * 4;
    const long w_tap_pitch = (long)p.K * 2, w_row_pitch = w_tap_pitch * taps;

    // slab staging: chunk c -> slab row c >> 2, 16-byte fp16 part c & 3  <-  32 bytes (8 floats) of the fp32 row.  Rows beyond
    // the conv's real span re-read the last needed row (cache hits, exact chunk counts, no predicates)
    const char* xsrc[XCH]; int xdst[XCH];
#pragma unroll
    for (int i = 0; i < XCH; ++i) {
        const int c = tid + i * 512, r = c >> 2, part = c & 3;
        const int rs = min(r, BM + (p.taps - 1) * p.dil - 1);
        xsrc[i] = reinterpret_cast<const char*>(p.A) + ((long)m0 - (long)p.center * p.dil + rs) * a_pitch + part * 32;
        xdst[i] = swz(r, part);
    }
    const int wr_ = (tid >> 2) % BN, wp_ = tid & 3;          // BN < 128: the upper threads duplicate chunks of the lower ones
    const char* const whg = reinterpret_cast<const char*>(p.W) + (long)(n0 + wr_) * w_row_pitch + wp_ * 16;
    const char* const wlg = reinterpret_cast<const char*>(p.W_lo) + (long)(n0 + wr_) * w_row_pitch + wp_ * 16;
    const int wdst = swz(wr_, wp_);
    const bool pro = p.pro_lrelu != 0;
    const float pro_slope = p.pro_slope;

    float4 xa0, xb0, xa1, xb1, xa2, xb2;                     // 3 chunks x 8 floats (scalars: arrays behind macros end up in scratch)
    uint4 whA, wlA, whB, wlB;
#define EV_X_GLOAD_X(KC)                                                                         \
    {                                                                                            \
        xa0 = *reinterpret_cast<const float4*>(xsrc[0] + (long)(KC) * 128);                      \
        xb0 = *reinterpret_cast<const float4*>(xsrc[0] + (long)(KC) * 128 + 16);                 \
        xa1 = *reinterpret_cast<const float4*>(xsrc[1] + (long)(KC) * 128);                      \
        xb1 = *reinterpret_cast<const float4*>(xsrc[1] + (long)(KC) * 128 + 16);                 \
        xa2 = *reinterpret_cast<const float4*>(xsrc[2] + (long)(KC) * 128);                      \
        xb2 = *reinterpret_cast<const float4*>(xsrc[2] + (long)(KC) * 128 + 16);                 \
    }
#define EV_X_SPLIT_STORE(XA, XB, DST, BUF)                                                       \
    {                                                                                            \
        float f_[8] = {XA.x, XA.y, XA.z, XA.w, XB.x, XB.y, XB.z, XB.w};                          \
        if (pro) {                                                                               \
            _Pragma("unroll") for (int e = 0; e < 8; ++e) f_[e] = fmaxf(f_[e], f_[e] * pro_slope); \
        }                                                                                        \
        half8 hi_, lo_;                                                                          \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                          \
            const _Float16 h_ = (_Float16)f_[e];                                                 \
            hi_[e] = h_;                                                                         \
            lo_[e] = (_Float16)((f_[e] - (float)h_) * 2048.0f);                                  \
        }                                                                                        \
        *reinterpret_cast<half8*>(Xh + (BUF) * XBUF + (DST)) = hi_;                              \
        *reinterpret_cast<half8*>(Xl + (BUF) * XBUF + (DST)) = lo_;                              \
    }
#define EV_X_SSTORE_X(BUF)                                                                       \
    {                                                                                            \
        EV_X_SPLIT_STORE(xa0, xb0, xdst[0], BUF)                                                 \
        EV_X_SPLIT_STORE(xa1, xb1, xdst[1], BUF)                                                 \
        EV_X_SPLIT_STORE(xa2, xb2, xdst[2], BUF)                                                 \
    }
#define EV_X_GLOAD_W(DST, KC, TAP)                                                               \
    {                                                                                            \
        const long o_ = (long)(TAP) * w_tap_pitch + (long)(KC) * 64;                             \
        wh##DST = *reinterpret_cast<const uint4*>(whg + o_);                                     \
        wl##DST = *reinterpret_cast<const uint4*>(wlg + o_);                                     \
    }
#define EV_X_SSTORE_W(SRC, BUF)                                                                  \
    {                                                                                            \
        *reinterpret_cast<uint4*>(Wh + (BUF) * WBUF + wdst) = wh##SRC;                           \
        *reinterpret_cast<uint4*>(Wl + (BUF) * WBUF + wdst) = wl##SRC;                           \
    }

    f32x4 acc[NT][MT], accl[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) { acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[a][b] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int fr = lane & 15, fq = lane >> 4;
    int woff[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) woff[a] = swz(wc * TC + a * 16 + fr, fq);

#define EV_X_COMPUTE(WSEL, KC, TAP)                                                                            \
    {                                                                                                          \
        const int row0_ = wt * TT + fr + (TAP) * p.dil;                                                        \
        const int xo_ = ((KC) & 1) * XBUF + row0_ * 64 + ((fq ^ ((row0_ >> 1) & 3)) << 4);                      \
        uint4 wh_[NT], wl_[NT];                                                                                \
        _Pragma("unroll") for (int a = 0; a < NT; ++a) {                                                       \
            wh_[a] = *reinterpret_cast<const uint4*>(Wh + (WSEL) * WBUF + woff[a]);                            \
            wl_[a] = *reinterpret_cast<const uint4*>(Wl + (WSEL) * WBUF + woff[a]);                            \
        }                                                                                                      \
        _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                       \
            const uint4 xh_ = *reinterpret_cast<const uint4*>(Xh + xo_ + b * 16 * 64);                         \
            const uint4 xl_ = *reinterpret_cast<const uint4*>(Xl + xo_ + b * 16 * 64);                         \
            _Pragma("unroll") for (int a = 0; a < NT; ++a) {                                                   \
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&wh_[a]), *reinterpret_cast<const half8*>(&xh_), acc[a][b], 0, 0, 0);   \
                accl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&wh_[a]), *reinterpret_cast<const half8*>(&xl_), accl[a][b], 0, 0, 0); \
                accl[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&wl_[a]), *reinterpret_cast<const half8*>(&xh_), accl[a][b], 0, 0, 0); \
            }                                                                                                  \
        }                                                                                                      \
    }

    // prologue: X(0), W(step 0) -> LDS; W(step 1) -> registers
    EV_X_GLOAD_X(0)
    EV_X_GLOAD_W(A, 0, 0)
    EV_X_SSTORE_X(0)
    EV_X_SSTORE_W(A, 0)
    {
        const int k1 = (taps > 1) ? 0 : (nkc > 1 ? 1 : 0), t1 = (taps > 1) ? 1 : 0;
        EV_X_GLOAD_W(B, k1, t1)
    }
    __syncthreads();

    int wsel = 0, kc = 0, tap = 0;
    // One pipeline step (see conv_gemm_kernel): LD receives W(s+2), ST holds W(s+1) and is written to the idle LDS buffer.
#define EV_X_STEP(LD, ST)                                                                  \
    {                                                                                      \
        const bool more_ = kc + 1 < nkc;                                                   \
        int t2_ = tap + 2, k2_ = kc;                                                       \
        if (t2_ >= taps) { t2_ -= taps; k2_ = kc + 1; }                                    \
        if (t2_ >= taps) { t2_ -= taps; k2_ += 1; }                                        \
        if (k2_ >= nkc) { k2_ = kc; t2_ = tap; }                                           \
        EV_X_GLOAD_W(LD, k2_, t2_)                                                         \
        if (tap == 0 && more_) { EV_X_GLOAD_X(kc + 1) }                                    \
        EV_X_COMPUTE(wsel, kc, tap)                                                        \
        EV_X_SSTORE_W(ST, wsel ^ 1)                                                        \
        if (tap == taps - 1) {                                                             \
            if (more_) { EV_X_SSTORE_X((kc + 1) & 1) }                                     \
            tap = 0; ++kc;                                                                 \
        } else {                                                                           \
            ++tap;                                                                         \
        }                                                                                  \
        __syncthreads();                                                                   \
        wsel ^= 1;                                                                         \
    }
    int s = 0;
    for (; s + 1 < steps; s += 2) {
        EV_X_STEP(A, B)
        EV_X_STEP(B, A)
    }
    if (s < steps) EV_X_STEP(A, B)
#undef EV_X_STEP
#undef EV_X_COMPUTE
#undef EV_X_GLOAD_X
#undef EV_X_SPLIT_STORE
#undef EV_X_SSTORE_X
#undef EV_X_GLOAD_W
#undef EV_X_SSTORE_W
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] += accl[a][b] * (1.0f / 2048.0f);
    if constexpr (EPI == EPI_GENERIC) gemm_epilogue_lds<MT, NT>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC);
    else {
        EV_TRACE_EPI_DUMMY
        gemm_epilogue_fast<MT, NT, EPI>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * TT, n0 + wc * TC EV_TRACE_EPI_ARGS);
    }
}

template <int BN>
static constexpr size_t x3_lds_bytes(int nxb = 2) { return 2 * (size_t)nxb * 384 * 64 + 4 * (size_t)BN * 64; }

template <int BN, int WM, int WN, int EPI>
static void launch_x3_cfg(const ConvGemmParams& p, hipStream_t s) {
    static_assert(8 * 32 * ((BN / WN) * 4 + 16) <= x3_lds_bytes<BN>(2), "epilogue scratch aliases the staging buffers");
    const int grid = (p.M / 256) * (p.N / BN);
    size_t lds = x3_lds_bytes<BN>(p.K > 32 ? 2 : 1);
    const size_t epi = 8 * 32 * (size_t)((BN / WN) * 4 + 16);          // the transposed epilogue re-uses the staging buffers
    if (epi > lds) lds = epi;
    hipLaunchKernelGGL((conv_gemm_x3_kernel<BN, WM, WN, EPI>), dim3(grid), dim3(512), lds, s, p);
}
template <int EPI>
static void launch_x3_epi(const ConvGemmParams& p, hipStream_t s) {
    if (p.N % 128 == 0) launch_x3_cfg<128, 4, 2, EPI>(p, s);
    else if (p.N % 64 == 0) launch_x3_cfg<64, 8, 1, EPI>(p, s);
    else launch_x3_cfg<32, 8, 1, EPI>(p, s);
}
// large-LDS opt-in of every instantiation launch_x3 can reach (called per device from init_device_kernels)
template <int EPI>
static hipError_t x3_attr_epi() {
    hipError_t e = hipSuccess, r;
    r = hipFuncSetAttribute((const void*)conv_gemm_x3_kernel<128, 4, 2, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes<128>(2)); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_gemm_x3_kernel<64, 8, 1, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes<64>(2)); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_gemm_x3_kernel<32, 8, 1, EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x3_lds_bytes<32>(2)); if (r != hipSuccess) e = r;
    return e;
}
static hipError_t x3_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = x3_attr_epi<EPI_O32>(); if (r != hipSuccess) e = r;
    r = x3_attr_epi<EPI_RARE_ACT | EPI_O32>(); if (r != hipSuccess) e = r;
    r = x3_attr_epi<EPI_RES32 | EPI_O32>(); if (r != hipSuccess) e = r;
    r = x3_attr_epi<EPI_RES32 | EPI_ACC32 | EPI_O32>(); if (r != hipSuccess) e = r;
    r = x3_attr_epi<EPI_GENERIC>(); if (r != hipSuccess) e = r;
    return e;
}

template <int BN, int EPI>
static void launch_split_bn(const ConvGemmParams& p, hipStream_t s) {
    constexpr int BM = 128;
    size_t lds = 4 * (size_t)(BM + MAX_SPAN) * 64 + 4 * (size_t)BN * 64;
    const size_t epi = 4 * (size_t)(32 * ((BN / 2) * 4 + 16));
    if (epi > lds) lds = epi;
    static_assert(4 * (size_t)(BM + MAX_SPAN) * 64 + 4 * (size_t)BN * 64 <= 65536, "within the default dynamic-LDS limit: no per-device opt-in needed");
    const int grid = (p.M / BM) * (p.N / BN) * (p.ksplit > 1 ? p.ksplit : 1);
    hipLaunchKernelGGL((conv_gemm_split_kernel<BM, BN, EPI>), dim3(grid), dim3(256), lds, s, p);
}
template <int EPI>
static void launch_split_epi(const ConvGemmParams& p, hipStream_t s) {
    if (p.N % 64 == 0) launch_split_bn<64, EPI>(p, s);
    else launch_split_bn<32, EPI>(p, s);
}

static void launch_split(const ConvGemmParams& p, hipStream_t s) {
    // split-precision GEMMs write fp32 only: plain / leaky-relu / fp32 residual (+ fp32 accumulate-in) / relu-gelu (exact erff:
    // fp32 output) use the straight-line epilogue, the per-utterance bias of embed_projection1 (and anything else) the generic one
    static const bool old_kernel = tuning_env("EV_X3_OLD") != nullptr;       // A/B switch: first-generation 128 x 64 kernel
    static const bool force_generic = tuning_env("EV_EPI_GENERIC") != nullptr;
    const bool rare_act = p.act != ACT_NONE && p.act != ACT_LRELU;
    const bool odd_slope = p.act == ACT_LRELU && !(p.act_slope >= 0.f && p.act_slope <= 1.f);       // max(v, s v) form needs s in [0, 1]
    const bool o32 = p.out32 && !p.out16 && !p.out32_before_post && !p.post_lrelu;
    const bool plain = !p.seq_bias && !p.add16_a && o32 && !odd_slope && !force_generic;
    const bool res32 = p.res && p.res_dtype == DT_F32;
    // few tiles (the token-rate GEMMs: 33 x 3..12 tiles of 256 x 128 at 32 x 256 tokens) cannot fill 256 CUs with one 8-wave block
    // each: the first-generation kernel's 128 x 64 tiles at two 4-wave blocks per CU give 4x the blocks (measured: encoder GEMMs
    // 1.40 ms vs 1.81 ms, predictors 0.31 vs 0.54 ms)
    if (p.ksplit > 1) {
        // split-K (checked by splitk_check): every (tile, K range) is a block of the 128 x 64-tile kernel with a bare fp32 epilogue into the scratch slices,
        // then the reduction applies the call's own epilogue.  Whatever M is: the summation order must not depend on the batch.
        ConvGemmParams q = p;
        q.bias = nullptr; q.act = ACT_NONE; q.row_valid = nullptr; q.row_seq = nullptr; q.seq_bias = nullptr; q.res = nullptr; q.acc32 = nullptr;
        q.out_scale = 1.0f; q.post_lrelu = 0; q.out16 = nullptr; q.out32_before_post = 0; q.add16_a = nullptr; q.add16_b = nullptr;
        q.out32 = reinterpret_cast<float*>(p.mx_scratch); q.ldo = p.N;
        launch_split_bn<64, EPI_O32>(q, s);
        const long threads = (long)p.M * (p.N / 8);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, p, reinterpret_cast<const float*>(p.mx_scratch));
        return;
    }
    const long tiles_x3 = (long)(p.M / 256) * (p.N % 128 == 0 ? p.N / 128 : (p.N % 64 == 0 ? p.N / 64 : p.N / 32));
    if (old_kernel || (tiles_x3 < 512 && p.N % 64 == 0)) {      // (same epilogue variant as the second-generation kernel would take)
        if (plain && !p.acc32 && !p.res && !rare_act) return launch_split_epi<EPI_O32>(p, s);
        if (plain && !p.acc32 && !p.res && rare_act) return launch_split_epi<EPI_RARE_ACT | EPI_O32>(p, s);
        if (plain && !p.acc32 && res32 && !rare_act) return launch_split_epi<EPI_RES32 | EPI_O32>(p, s);
        if (plain && p.acc32 && res32 && !rare_act) return launch_split_epi<EPI_RES32 | EPI_ACC32 | EPI_O32>(p, s);
        return launch_split_epi<EPI_GENERIC>(p, s);
    }
    if (plain && !p.acc32 && !p.res && !rare_act) return launch_x3_epi<EPI_O32>(p, s);
    if (plain && !p.acc32 && !p.res && rare_act) return launch_x3_epi<EPI_RARE_ACT | EPI_O32>(p, s);
    if (plain && !p.acc32 && res32 && !rare_act) return launch_x3_epi<EPI_RES32 | EPI_O32>(p, s);
    if (plain && p.acc32 && res32 && !rare_act) return launch_x3_epi<EPI_RES32 | EPI_ACC32 | EPI_O32>(p, s);
    launch_x3_epi<EPI_GENERIC>(p, s);
}

// =====================================================================================================================
// Fused ResBlock pair, C = 32 (HiFi-GAN stage 3: 256 samples x 32 channels per mel frame, the HBM-bound end of the vocoder:
// layer-wise each conv moves 64 B in + 64 B out per sample for 2*32*32*k FLOP).  One persistent block per CU:
//   * W1 and W2 (k taps x 32 x 32 fp16 each, <= 22.5 KB) are loaded into LDS once per block and stay there; every wave
//     pulls the current conv's 2*k fragments into registers once per tile, so the tap loop reads only activation fragments;
//   * a tile = 256 rows of conv1 output = 256 - (k-1) rows of conv2 output; the raw input slab (256 + (k-1)*d rows) of the
//     NEXT tile is prefetched into registers during the current tile; leaky-relu is applied on conv1's fragments and the
//     residual add reads the same slab, so x crosses HBM exactly once per pair;
//   * conv1's bias + leaky-relu + sequence-edge masking happen in registers, xt goes to LDS as fp16, conv2 reads it with
//     dilation 1, and the residual / MRF epilogue is the same LDS-transposed coalesced epilogue as the GEMM kernel.
// HBM traffic per pair: read x once (+ the residual re-read, an L2 hit), write once -- vs 5 tensor passes layer-wise.
__device__ char g_store_trash[64 * 64];     // masked lanes of the fused kernel's stores land here (no exec-masked branch)

// Tuning builds only (build.py --variant <tag> EV_PAIR_ABLATE=<bits>): 1 no MFMAs, 2 no slab requests (the staged registers keep their
// first tile), 4 no output stores, 8 no residual / MRF operand loads, 16 no block barriers.  Results are garbage by design.
#ifndef EV_PAIR_ABLATE
#define EV_PAIR_ABLATE 0
#endif
#define EV_PAIR_SYNC() { if (!(EV_PAIR_ABLATE & 16)) __syncthreads(); }
#define EV_PAIR_MFMA(ACC, A, B) { if (!(EV_PAIR_ABLATE & 1)) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, ACC, 0, 0, 0); else asm volatile("" : "+v"(ACC) : "v"(A), "v"(B)); }

// ACCMODE: 0 = none, 1 = fp32 accumulate-in (epi.acc32), 2 = two fp16 addends (epi.add16_a / add16_b)
// TWOB (k = 3, 7): two blocks per CU.  Ablations (profiles/r2_m_pair_c32_ablation.txt) showed the kernel insensitive to every single
// ingredient -- no slab requests, no stores, no MFMAs, no barriers: 307 -> 267...299 us at k = 3 -- i.e. bound by the latency of its own
// dependent instruction chain at two waves per SIMD, not by HBM.  The footprint shrinks to <= 71 KB: 320-row slabs, the transposing
// scratch aliased onto the slab conv1 has finished with (16-row passes), the weights staged through the slab area before the first
// tile (they live in registers afterwards; conv2's set stays in LDS at k = 7 to fit 128 VGPRs).
// As built: k = 3 only; the weight fragments are read per tap (no resident sets: 128 VGPRs), the weights stay in LDS (69 KB per block).
template <int K, int ACCMODE, bool TWOB>
__global__ __launch_bounds__(512, TWOB ? 4 : 1) void resblock_pair_c32_kernel(const ResPairParams p) {
    // 8 waves x 32 rows, one persistent block per CU.  The memory instruction stream of the tile loop is straight-line: the
    // first version loaded row_valid bytes / MRF rows inside runtime-flag branches and an exec-masked third slab chunk, and
    // hipcc answered with s_waitcnt vmcnt(0) right after the next tile's slab prefetch (no prefetch at all) and between the
    // stores.  Now per tile, in issue order: [MRF rows of THIS tile] [slab + row-valid byte of the NEXT tile] conv1 -> xt ->
    // conv2 -> next slab to LDS -> stores; every wait is counted, the stores drain under the next tile's conv1.
    // VALU budget: timelines showed the block VALU-bound (leaky-relu re-applied to every tap's fragment, scalar fp32 epilogue
    // math, 64-bit clamped addressing ~ 300-450 VALU per wave and tile against 24-88 MFMAs), so leaky-relu(x) is applied ONCE
    // while staging the slab (the raw residual rows are re-read from L2 at the top of the tile), xt / output math runs on
    // packed fp32 pairs, masks act on packed fp16 words and only for 32-row groups that contain invalid rows.
    constexpr int C = 32, H2 = (K - 1) / 2, BMO = 256 - 2 * H2;
    constexpr int XROWS = TWOB ? 320 : 384, XTROWS = 272, XCH = 3;
    constexpr int WBYTES = K * C * 64, XBYTES = XROWS * 64, XTBYTES = XTROWS * 64;
    constexpr int EROWS = TWOB ? 16 : 32;                             // rows per transposing pass
    constexpr int EPITCH = C * 4 + 16, EBYTES = EROWS * EPITCH;       // per-wave transpose scratch: EROWS rows x 32 fp32
    // conv1's weight fragments (k * 2 x 16 B per lane) stay in registers for every tile of this persistent block, conv2's too
    // up to k = 7 (k = 11: 176 VGPRs for both sets; conv2's set is then re-read from LDS per tile); TWOB (128 VGPRs): up to k = 3
    constexpr bool W2_RESIDENT = TWOB ? false : (K <= 7);
    constexpr bool W1_RESIDENT = !TWOB;                               // TWOB: both sets are read per tap (the 128-VGPR budget)
    static_assert(!TWOB || (8 * EBYTES <= XBYTES && 256 + 2 * 5 * H2 + 2 * H2 <= XROWS), "TWOB aliasing");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [W1][W2][Xa 2 x slab][Xt][Es]; TWOB: no Es region -- the scratch is the slab of the tile being finished
    char* W1s = smem;
    char* W2s = smem + WBYTES;
    char* Xa = smem + 2 * WBYTES;                                     // [2][XROWS][64]
    char* Xt = Xa + 2 * XBYTES;                                       // [XTROWS][64]
    char* Es = Xt + XTBYTES;                                          // [8][EBYTES] (1 block / CU)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int h1 = H2 * p.dil;
    const int x_pitch = p.ldx * 2;           // bytes; (long)row * x_pitch is one v_mad_i64_i32
    const char* xg = reinterpret_cast<const char*>(p.x);
    const int ntiles = (p.M + BMO - 1) / BMO;
    const ConvGemmParams& e = p.epi;
    const int gmin = p.gmax ? p.gmin : 0, gmax = p.gmax ? p.gmax : p.M;      // rows of x that exist (sub-range launches)

    for (int c = tid; c < K * C * 4; c += 512) {
        const int row = c >> 2, part = c & 3, tap = row >> 5, co = row & 31;
        const long off = ((long)(co * K + tap) * C) * 2 + part * 16;
        *reinterpret_cast<uint4*>(W1s + swz(row, part)) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w1) + off);
        *reinterpret_cast<uint4*>(W2s + swz(row, part)) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w2) + off);
    }
    f32x2 b1v[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) b1v[a][q] = f32x2{p.b1[a * 16 + 4 * fq + 2 * q], p.b1[a * 16 + 4 * fq + 2 * q + 1]};
    // coalesced-side mapping of the epilogue: 4 lanes per row (8 channels each), 16 rows per wave-instruction
    const int er = lane >> 2, eg = lane & 3, eco = eg * 8;
    const unsigned frbit = 1u << fr, erbit = 1u << er;
    f32x2 b2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = e.bias ? f32x2{e.bias[eco + 2 * q], e.bias[eco + 2 * q + 1]} : f32x2{0.f, 0.f};
    const f32x2 out_scale2 = f32x2{e.out_scale, e.out_scale};
    const bool has_post = e.post_lrelu != 0;
    const f32x2 post_slope2 = f32x2{e.post_slope, e.post_slope};
    const f32x2 slope01 = f32x2{0.1f, 0.1f};
    __half* const o16 = reinterpret_cast<__half*>(e.out16);
    float* const o32a = e.out32_before_post ? e.out32 : nullptr;
    float* const o32b = e.out32_before_post ? nullptr : e.out32;
    char* const trash = g_store_trash + lane * 64;

    // row validity of a tile: lane l of wave w looks at global row (tile * BMO - H2 + 32 w + l); a ballot turns the 64 bytes
    // into a wave-uniform mask.  conv1's xt rows use bits [0, 32), the output rows bits [H2, H2 + 32).
    const uint8_t* vptr = e.row_valid ? e.row_valid : g_row_always_valid;
    const int vshift = e.row_valid ? e.valid_shift : 31;
#define EV_PAIR_VROW(TILE) ((TILE) * BMO - H2 + wave * 32 + lane)          /* 32-bit: M < 2^31 rows */
#define EV_PAIR_VLOAD(TILE, DST)                                                                           \
    {                                                                                                      \
        const int g_ = EV_PAIR_VROW(TILE);                                                                 \
        DST = vptr[min(max(g_, gmin), gmax - 1) >> vshift];                                                \
    }
#define EV_PAIR_VMASK(TILE, SRC) __builtin_amdgcn_ballot_w64((SRC) != 0 && EV_PAIR_VROW(TILE) >= gmin && EV_PAIR_VROW(TILE) < gmax)

    uint4 xr0, xr1, xr2;
    // (TWOB: the rows past the 320-row slab all hold the last needed row -- see xrow2 -- so their chunks may land on row 319 together)
    const int xd0 = swz(tid >> 2, tid & 3), xd1 = swz((tid + 512) >> 2, tid & 3), xd2 = swz(min((tid + 1024) >> 2, XROWS - 1), tid & 3);
    static_assert(XCH == 3, "three 16-B chunks per thread");
    const char* const xgt = xg + (tid & 3) * 16;
    const int xrow2 = min((tid >> 2) + 256, 255 + 2 * h1 + 2 * H2);
    // rows beyond the slab the convs read (256 + 2 h1 + 2 H2 <= 316) re-read the last needed row: a cache hit, not HBM traffic
    // (the first slab row is >= gmin - (H2 + h1) >= gmin - 30, inside the 64 slack rows; only the upper end needs a clamp)
#define EV_PAIR_ROW(G) min((G), gmax + 63)
#define EV_PAIR_GLOAD(TILE)                                                                                \
    {                                                                                                      \
        const int g0_ = (TILE) * BMO - H2 - h1 + (tid >> 2);                                               \
        const int g2_ = (TILE) * BMO - H2 - h1 + xrow2;                                                    \
        xr0 = *reinterpret_cast<const uint4*>(xgt + (long)EV_PAIR_ROW(g0_) * x_pitch);                     \
        xr1 = *reinterpret_cast<const uint4*>(xgt + (long)EV_PAIR_ROW(g0_ + 128) * x_pitch);               \
        xr2 = *reinterpret_cast<const uint4*>(xgt + (long)EV_PAIR_ROW(g2_) * x_pitch);                     \
    }
    // leaky_relu(x, .1) of models.py:51 once per element while staging (not once per tap on the fragments)
#define EV_PAIR_SSTORE(BUF)                                                                                \
    {                                                                                                      \
        *reinterpret_cast<uint4*>(Xa + (BUF) * XBYTES + xd0) = lrelu_h8(xr0, 0.1f);                        \
        *reinterpret_cast<uint4*>(Xa + (BUF) * XBYTES + xd1) = lrelu_h8(xr1, 0.1f);                        \
        *reinterpret_cast<uint4*>(Xa + (BUF) * XBYTES + xd2) = lrelu_h8(xr2, 0.1f);                        \
    }

    int tile = blockIdx.x;                    // grid <= ntiles
    unsigned long long vmask;
    uint4 wf1[K][2], wf2[K][2];
#define EV_PAIR_WFILL()                                                                                        \
    _Pragma("unroll") for (int t = 0; t < K; ++t)                                                              \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                        \
            if constexpr (W1_RESIDENT) wf1[t][a] = *reinterpret_cast<const uint4*>(W1s + swz(t * 32 + a * 16 + fr, fq)); \
            if constexpr (W2_RESIDENT) wf2[t][a] = *reinterpret_cast<const uint4*>(W2s + swz(t * 32 + a * 16 + fr, fq)); \
        }
    {
        uint8_t vb;
        EV_PAIR_GLOAD(tile)
        EV_PAIR_VLOAD(tile, vb)
        EV_PAIR_SSTORE(0)
        vmask = EV_PAIR_VMASK(tile, vb);
    }
    __syncthreads();
    EV_PAIR_WFILL()
#undef EV_PAIR_WFILL
    int cur = 0;
    const int wrow0 = wave * 32 + fr;
    for (; tile < ntiles; tile += gridDim.x) {
        char* const es = (TWOB ? Xa + cur * XBYTES : Es) + wave * EBYTES;      // (TWOB: the slab conv1 has finished with)
        const int next = min(tile + (int)gridDim.x, ntiles - 1);      // clamped: the last prefetch of a block is never used
        const int m0 = tile * BMO;
        const int t_end = min(m0 + BMO, p.M);
        // ---------------- memory requests of this iteration, oldest first
        uint4 resv[2];          // raw residual rows x[t] (the LDS slab holds leaky_relu(x)); an L2 hit: the slab just came through
        float4 accin[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = min(m0 + wave * 32 + it * 16 + er, t_end - 1);
            if (!(EV_PAIR_ABLATE & 8)) resv[it] = *reinterpret_cast<const uint4*>(xg + (long)t * x_pitch + eg * 16); else resv[it] = make_uint4(t, it, lane, 0);
            if constexpr (ACCMODE == 1) {
                const float* ap = e.acc32 + (long)t * e.ldacc + eco;
                accin[it][0] = *reinterpret_cast<const float4*>(ap);
                accin[it][1] = *reinterpret_cast<const float4*>(ap + 4);
            }
            if constexpr (ACCMODE == 2 && !(EV_PAIR_ABLATE & 8)) {        // the same 8 registers per row hold the two fp16 addends
                *reinterpret_cast<uint4*>(&accin[it][0]) = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(e.add16_a) + (long)t * e.ldadd + eco);
                *reinterpret_cast<uint4*>(&accin[it][1]) = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(e.add16_b) + (long)t * e.ldadd + eco);
            }
        }
        uint8_t vb_next;
        if (!(EV_PAIR_ABLATE & 2)) { EV_PAIR_GLOAD(next) }
        EV_PAIR_VLOAD(next, vb_next)
        __builtin_amdgcn_sched_barrier(0);    // hipcc otherwise sinks these requests below conv1, next to their use
        f32x4 acc[2][2];
        // ---------------- conv1 (dilation d): 256 rows, global rows m0 - H2 + r1
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const char* Xb = Xa + cur * XBYTES;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int r0 = wrow0 + t * p.dil;
                const char* xp = Xb + r0 * 64 + ((fq ^ ((r0 >> 1) & 3)) << 4);
                if constexpr (!W1_RESIDENT) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) wf1[t][a] = *reinterpret_cast<const uint4*>(W1s + swz(t * 32 + a * 16 + fr, fq));
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    uint4 xf = *reinterpret_cast<const uint4*>(xp + b * 16 * 64);
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        EV_PAIR_MFMA(acc[a][b], *reinterpret_cast<half8*>(&wf1[t][a]), *reinterpret_cast<half8*>(&xf))
                }
            }
        }
        // bias + leaky-relu + zero outside the utterance (conv2 must see the reference's zero padding) -> Xt (fp16)
        const unsigned xtmask = (unsigned)vmask;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r1 = wrow0 + b * 16;
            const bool valid = (xtmask & (frbit << (b * 16))) != 0u;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const f32x2 lo = lrelu2(f32x2{acc[a][b][0], acc[a][b][1]} + b1v[a][0], slope01);
                const f32x2 hi = lrelu2(f32x2{acc[a][b][2], acc[a][b][3]} + b1v[a][1], slope01);
                uint2 w;
                *reinterpret_cast<half2v*>(&w.x) = __builtin_convertvector(lo, half2v);
                *reinterpret_cast<half2v*>(&w.y) = __builtin_convertvector(hi, half2v);
                w.x = valid ? w.x : 0u; w.y = valid ? w.y : 0u;
                const int co = a * 16 + 4 * fq;
                *reinterpret_cast<uint2*>(Xt + swz(r1, co >> 3) + (co & 7) * 2) = w;
            }
        }
        EV_PAIR_SYNC()
        // ---------------- conv2 (dilation 1): rows m0 + r2, reads Xt rows r2 + t
        if constexpr (!W2_RESIDENT && !TWOB) {
#pragma unroll
            for (int t = 0; t < K; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a) wf2[t][a] = *reinterpret_cast<const uint4*>(W2s + swz(t * 32 + a * 16 + fr, fq));
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int r0 = wrow0 + t;
            const char* xp = Xt + r0 * 64 + ((fq ^ ((r0 >> 1) & 3)) << 4);
            if constexpr (TWOB) {
#pragma unroll
                for (int a = 0; a < 2; ++a) wf2[t][a] = *reinterpret_cast<const uint4*>(W2s + swz(t * 32 + a * 16 + fr, fq));
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                uint4 xf = *reinterpret_cast<const uint4*>(xp + b * 16 * 64);
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    EV_PAIR_MFMA(acc[a][b], *reinterpret_cast<half8*>(&wf2[t][a]), *reinterpret_cast<half8*>(&xf))
            }
        }
        // ---------------- the next tile's slab goes to the idle buffer BEFORE this tile's stores are issued, so that its wait
        // (the oldest requests in flight) never has to drain them
        __builtin_amdgcn_sched_barrier(0);    // ... and hoists their wait into conv2
        EV_PAIR_SSTORE(cur ^ 1)
        const unsigned long long vmask_next = EV_PAIR_VMASK(next, vb_next);
        // ---------------- epilogue: transpose through the wave's scratch, then 16-byte row-contiguous stores
        if constexpr (!TWOB) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    *reinterpret_cast<f32x4*>(es + (b * 16 + fr) * EPITCH + (a * 16 + 4 * fq) * 4) = acc[a][b];
            __builtin_amdgcn_wave_barrier();
        }
        const unsigned outmask = (unsigned)(vmask >> H2);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int lr = it * 16 + er;
            const int t = m0 + wave * 32 + lr;
            const bool rowok = t < t_end;
            const bool valid = (outmask & (erbit << (it * 16))) != 0u;
            if constexpr (TWOB) {             // 16-row passes: row group `it` through the wave's 16-row scratch
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    *reinterpret_cast<f32x4*>(es + fr * EPITCH + (a * 16 + 4 * fq) * 4) = acc[a][it];
                __builtin_amdgcn_wave_barrier();
            }
            const int sr = TWOB ? er : lr;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(es + sr * EPITCH + eg * 32);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(es + sr * EPITCH + eg * 32 + 16);
            f32x2 v[4] = {f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}};
            const half2v* hh = reinterpret_cast<const half2v*>(&resv[it]);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = (v[q] + b2v[q] + __builtin_convertvector(hh[q], f32x2)) * out_scale2;
            if constexpr (ACCMODE == 1) {
                v[0] += f32x2{accin[it][0].x, accin[it][0].y}; v[1] += f32x2{accin[it][0].z, accin[it][0].w};
                v[2] += f32x2{accin[it][1].x, accin[it][1].y}; v[3] += f32x2{accin[it][1].z, accin[it][1].w};
            }
            if constexpr (ACCMODE == 2) {
                const half2v* ha = reinterpret_cast<const half2v*>(&accin[it][0]);
                const half2v* hb = reinterpret_cast<const half2v*>(&accin[it][1]);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += __builtin_convertvector(ha[q], f32x2) + __builtin_convertvector(hb[q], f32x2);
            }
            if (o32a) {
                float* op = rowok ? o32a + (long)t * e.ldo + eco : reinterpret_cast<float*>(trash);
                *reinterpret_cast<float4*>(op) = valid ? make_float4(v[0][0], v[0][1], v[1][0], v[1][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(op + 4) = valid ? make_float4(v[2][0], v[2][1], v[3][0], v[3][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (has_post) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = lrelu2(v[q], post_slope2);
            }
            if (o32b) {
                float* op = rowok ? o32b + (long)t * e.ldo + eco : reinterpret_cast<float*>(trash);
                *reinterpret_cast<float4*>(op) = valid ? make_float4(v[0][0], v[0][1], v[1][0], v[1][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(op + 4) = valid ? make_float4(v[2][0], v[2][1], v[3][0], v[3][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (o16) {
                uint4 o;
                half2v* h = reinterpret_cast<half2v*>(&o);
#pragma unroll
                for (int q = 0; q < 4; ++q) h[q] = __builtin_convertvector(v[q], half2v);
                o.x = valid ? o.x : 0u; o.y = valid ? o.y : 0u; o.z = valid ? o.z : 0u; o.w = valid ? o.w : 0u;
                if (!(EV_PAIR_ABLATE & 4)) *reinterpret_cast<uint4*>(rowok ? reinterpret_cast<char*>(o16 + (long)t * e.ldo + eco) : trash) = o;
                else asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));
            }
        }
        vmask = vmask_next;
        EV_PAIR_SYNC()
        cur ^= 1;
    }
#undef EV_PAIR_GLOAD
#undef EV_PAIR_SSTORE
#undef EV_PAIR_ROW
#undef EV_PAIR_VROW
#undef EV_PAIR_VLOAD
#undef EV_PAIR_VMASK
}

static size_t pair_lds_bytes(int k, bool twob = false) {
    if (twob) return 2 * (size_t)k * 32 * 64 + 2 * 320 * 64 + 272 * 64;
    return 2 * (size_t)k * 32 * 64 + 2 * 384 * 64 + 272 * 64 + 8 * 32 * (32 * 4 + 16);
}
static constexpr size_t PAIR64_LDS_BYTES = 2 * 2 * (size_t)(3 * 64 * 64) + 2 * 2 * (size_t)(272 * 64) + 2 * (size_t)(272 * 64);

// Per-device kernel state.  Kernels that use more than 64 KB of dynamic LDS need the opt-in attribute on EVERY device they are
// launched on, and the persistent kernels size their grid by the device's CU count: both are set up per device (ev_create
// calls init_device_kernels after hipSetDevice; the launchers look the current device up), never in a function-local static.
static int g_n_cu[64];
static bool g_dev_ready[64];
static int current_device() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < 64) ? d : 0; }

static int device_cus() {
    const int d = current_device();
    if (!g_dev_ready[d]) (void)init_device_kernels(d);       // per-kernel test entry points without a handle
    return g_n_cu[d] > 0 ? g_n_cu[d] : 256;
}

void launch_resblock_pair_c32(const ResPairParams& p, hipStream_t s) {
    const int n_cu = device_cus();
    const int h2 = (p.k - 1) / 2, bmo = 256 - 2 * h2;
    const int ntiles = (p.M + bmo - 1) / bmo;
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    static const char* one_env = tuning_env("EV_PAIR_1B");                   // "1": one block per CU for every k (A/B switch)
    const bool twob = p.k <= 3 && !(one_env && one_env[0] == '1') && !(p.epi.reserved0 & 4) && ntiles >= 4 * n_cu;     // (reserved0 bit 2: in-process A/B)
    const size_t lds = pair_lds_bytes(p.k, twob);
    const int grid2 = ntiles < 2 * n_cu ? ntiles : 2 * n_cu;
#define EV_PAIR_LAUNCH(KK, TB, G)                                                                                          \
        if (p.epi.add16_a) hipLaunchKernelGGL((resblock_pair_c32_kernel<KK, 2, TB>), dim3(G), dim3(512), lds, s, p);  \
        else if (p.epi.acc32) hipLaunchKernelGGL((resblock_pair_c32_kernel<KK, 1, TB>), dim3(G), dim3(512), lds, s, p); \
        else hipLaunchKernelGGL((resblock_pair_c32_kernel<KK, 0, TB>), dim3(G), dim3(512), lds, s, p);
    switch (p.k) {
        case 3: if (twob) { EV_PAIR_LAUNCH(3, true, grid2) } else { EV_PAIR_LAUNCH(3, false, grid) } break;
        case 7: EV_PAIR_LAUNCH(7, false, grid) break;
        case 11: EV_PAIR_LAUNCH(11, false, grid) break;
        default: break;
    }
#undef EV_PAIR_LAUNCH
}

// =====================================================================================================================
// Fused ResBlock pair, C = 64, k = 3 (HiFi-GAN stage 2: 128 samples x 64 channels per mel frame).  The k = 3 layers of that
// stage run at the HBM roofline of their layer-wise traffic (conv1 4.4 TB/s, conv2 5.1 TB/s measured), so the only way down is
// not to move the intermediate: same structure as the C = 32 kernel (persistent 8-wave block, conv1 -> xt in LDS -> conv2 ->
// residual / MRF epilogue, straight-line memory stream), with the channel dimension as two 64-byte K-chunk "planes" so that
// all of the swizzle / fragment geometry carries over.  Both weight sets (48 KB) stay in LDS for the whole kernel and are
// read as fragments (4 per (tap, chunk), shared by the wave's two row groups); the transposed-epilogue scratch aliases the
// slab buffer of the tile being finished (dead after conv1), which is what makes 150 KB fit.  k = 7 / 11 do not fit
// (weights 112 / 176 KB) and stay layer-wise: they are MFMA-bound there anyway.
template <int K, int ACCMODE>
__global__ __launch_bounds__(512, 1) void resblock_pair_c64_kernel(const ResPairParams p) {
    constexpr int C = 64, H2 = (K - 1) / 2, BMO = 256 - 2 * H2;
    constexpr int XROWS = 272, PLANE = XROWS * 64, XBYTES = 2 * PLANE;   // slab buffer = 2 K-chunk planes of [rows][64 B]
    constexpr int WPL = K * C * 64, WBYTES = 2 * WPL;                    // per conv: 2 planes of [(tap, co)][64 B]
    constexpr int EPITCH = C * 4 + 16, EBYTES = 16 * EPITCH;             // per-wave transpose scratch: 16 rows x 64 fp32
    constexpr int XCH = (XROWS * 8 + 511) / 512;
    static_assert(8 * EBYTES <= XBYTES && 256 + 2 * 5 * H2 + 2 * H2 <= XROWS && XCH == 5, "LDS aliasing / slab geometry");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* W1s = smem;
    char* W2s = smem + WBYTES;
    char* Xa = smem + 2 * WBYTES;            // [2][2 planes][XROWS][64]
    char* Xt = Xa + 2 * XBYTES;              // [2 planes][XROWS][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int h1 = H2 * p.dil;
    const int x_pitch = p.ldx * 2;
    const char* xg = reinterpret_cast<const char*>(p.x);
    const int ntiles = (p.M + BMO - 1) / BMO;
    const ConvGemmParams& e = p.epi;
    const int gmin = p.gmax ? p.gmin : 0, gmax = p.gmax ? p.gmax : p.M;      // rows of x that exist (sub-range launches)

    for (int c = tid; c < K * C * 8; c += 512) {
        const int row = c >> 3, q = c & 7, tap = row >> 6, co = row & 63, plane = q >> 2, part = q & 3;
        const long off = ((long)(co * K + tap) * C) * 2 + q * 16;
        *reinterpret_cast<uint4*>(W1s + plane * WPL + swz(row, part)) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w1) + off);
        *reinterpret_cast<uint4*>(W2s + plane * WPL + swz(row, part)) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w2) + off);
    }
    f32x2 b1v[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) b1v[a][q] = f32x2{p.b1[a * 16 + 4 * fq + 2 * q], p.b1[a * 16 + 4 * fq + 2 * q + 1]};
    // coalesced side of the epilogue: 8 lanes per row (8 channels each), 8 rows per wave-instruction
    const int er = lane >> 3, eg = lane & 7, eco = eg * 8;
    const unsigned frbit = 1u << fr, erbit = 1u << er;
    f32x2 b2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = e.bias ? f32x2{e.bias[eco + 2 * q], e.bias[eco + 2 * q + 1]} : f32x2{0.f, 0.f};
    const f32x2 out_scale2 = f32x2{e.out_scale, e.out_scale};
    const bool has_post = e.post_lrelu != 0;
    const f32x2 post_slope2 = f32x2{e.post_slope, e.post_slope};
    const f32x2 slope01 = f32x2{0.1f, 0.1f};
    __half* const o16 = reinterpret_cast<__half*>(e.out16);
    float* const o32a = e.out32_before_post ? e.out32 : nullptr;
    float* const o32b = e.out32_before_post ? nullptr : e.out32;
    char* const trash = g_store_trash + lane * 64;

    const uint8_t* vptr = e.row_valid ? e.row_valid : g_row_always_valid;
    const int vshift = e.row_valid ? e.valid_shift : 31;
#define EV_P64_VROW(TILE) ((TILE) * BMO - H2 + wave * 32 + lane)
#define EV_P64_VLOAD(TILE, DST) { const int g_ = EV_P64_VROW(TILE); DST = vptr[min(max(g_, gmin), gmax - 1) >> vshift]; }
#define EV_P64_VMASK(TILE, SRC) __builtin_amdgcn_ballot_w64((SRC) != 0 && EV_P64_VROW(TILE) >= gmin && EV_P64_VROW(TILE) < gmax)

    // slab staging: chunk c = tid + 512 i -> row c >> 3, 16-byte column c & 7 (plane = column >> 2); the fifth chunk only exists
    // for waves 0-1 (272 rows x 8 columns = 4 x 512 + 128)
    uint4 xr[XCH];
    const int xrow = tid >> 3, xq = tid & 7;
    const char* const xgt = xg + xq * 16;
    const int xdst = (xq >> 2) * PLANE + swz(xrow, xq & 3);           // + i * 64 rows: (row + 64 i) keeps the swizzle phase
    const bool has5 = wave < 2;
#define EV_P64_GLOAD(TILE)                                                                                 \
    {                                                                                                      \
        const int g0_ = (TILE) * BMO - H2 - h1 + xrow;                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
            xr[i] = *reinterpret_cast<const uint4*>(xgt + (long)min(g0_ + 64 * i, gmax + 63) * x_pitch);    \
        if (has5) xr[4] = *reinterpret_cast<const uint4*>(xgt + (long)min(g0_ + 256, gmax + 63) * x_pitch); \
    }
#define EV_P64_SSTORE(BUF)                                                                                 \
    {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
            *reinterpret_cast<uint4*>(Xa + (BUF) * XBYTES + xdst + i * 4096) = lrelu_h8(xr[i], 0.1f);      \
        if (has5) *reinterpret_cast<uint4*>(Xa + (BUF) * XBYTES + xdst + 4 * 4096) = lrelu_h8(xr[4], 0.1f); \
    }

    int tile = blockIdx.x;                    // grid <= ntiles
    unsigned long long vmask;
    {
        uint8_t vb;
        EV_P64_GLOAD(tile)
        EV_P64_VLOAD(tile, vb)
        EV_P64_SSTORE(0)
        vmask = EV_P64_VMASK(tile, vb);
    }
    __syncthreads();
    int cur = 0;
    const int wrow0 = wave * 32 + fr;
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = min(tile + (int)gridDim.x, ntiles - 1);
        const int m0 = tile * BMO;
        const int t_end = min(m0 + BMO, p.M);
        // ---------------- memory requests of this iteration, oldest first: residual / MRF rows of THIS tile, slab of the NEXT
        uint4 resv[2][2];
        float4 accin[2][2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int t = min(m0 + wave * 32 + b * 16 + it * 8 + er, t_end - 1);
                resv[b][it] = *reinterpret_cast<const uint4*>(xg + (long)t * x_pitch + eg * 16);
                if constexpr (ACCMODE == 1) {
                    const float* ap = e.acc32 + (long)t * e.ldacc + eco;
                    accin[b][it][0] = *reinterpret_cast<const float4*>(ap);
                    accin[b][it][1] = *reinterpret_cast<const float4*>(ap + 4);
                }
                if constexpr (ACCMODE == 2) {
                    *reinterpret_cast<uint4*>(&accin[b][it][0]) = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(e.add16_a) + (long)t * e.ldadd + eco);
                    *reinterpret_cast<uint4*>(&accin[b][it][1]) = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(e.add16_b) + (long)t * e.ldadd + eco);
                }
            }
        uint8_t vb_next;
        EV_P64_GLOAD(next)
        EV_P64_VLOAD(next, vb_next)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[4][2];
        // ---------------- conv1 (dilation d): 256 rows, global rows m0 - H2 + r1
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const char* Xb = Xa + cur * XBYTES;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                const int r0 = wrow0 + t * p.dil;
                const int xo = r0 * 64 + ((fq ^ ((r0 >> 1) & 3)) << 4);
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    uint4 xf[2], wf[4];
#pragma unroll
                    for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const uint4*>(Xb + kc * PLANE + xo + b * 16 * 64);
#pragma unroll
                    for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const uint4*>(W1s + kc * WPL + swz(t * 64 + a * 16 + fr, fq));
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wf[a]),
                                                                               *reinterpret_cast<half8*>(&xf[b]), acc[a][b], 0, 0, 0);
                }
            }
        }
        // bias + leaky-relu + zero outside the utterance (conv2 must see the reference's zero padding) -> Xt (fp16, 2 planes)
        const unsigned xtmask = (unsigned)vmask;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r1 = wrow0 + b * 16;
            const bool valid = (xtmask & (frbit << (b * 16))) != 0u;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x2 lo = lrelu2(f32x2{acc[a][b][0], acc[a][b][1]} + b1v[a][0], slope01);
                const f32x2 hi = lrelu2(f32x2{acc[a][b][2], acc[a][b][3]} + b1v[a][1], slope01);
                uint2 w;
                *reinterpret_cast<half2v*>(&w.x) = __builtin_convertvector(lo, half2v);
                *reinterpret_cast<half2v*>(&w.y) = __builtin_convertvector(hi, half2v);
                w.x = valid ? w.x : 0u; w.y = valid ? w.y : 0u;
                const int c32 = (a & 1) * 16 + 4 * fq;
                *reinterpret_cast<uint2*>(Xt + (a >> 1) * PLANE + swz(r1, c32 >> 3) + (c32 & 7) * 2) = w;
            }
        }
        __syncthreads();
        // ---------------- conv2 (dilation 1): rows m0 + r2, reads Xt rows r2 + t
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int r0 = wrow0 + t;
            const int xo = r0 * 64 + ((fq ^ ((r0 >> 1) & 3)) << 4);
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                uint4 xf[2], wf[4];
#pragma unroll
                for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const uint4*>(Xt + kc * PLANE + xo + b * 16 * 64);
#pragma unroll
                for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const uint4*>(W2s + kc * WPL + swz(t * 64 + a * 16 + fr, fq));
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wf[a]),
                                                                           *reinterpret_cast<half8*>(&xf[b]), acc[a][b], 0, 0, 0);
            }
        }
        // ---------------- next tile's slab -> idle buffer (its wait comes before this tile's stores are issued)
        __builtin_amdgcn_sched_barrier(0);
        EV_P64_SSTORE(cur ^ 1)
        const unsigned long long vmask_next = EV_P64_VMASK(next, vb_next);
        // ---------------- epilogue: 16 rows at a time through the wave's scratch (inside the finished tile's slab buffer)
        char* es = Xa + cur * XBYTES + wave * EBYTES;
        const unsigned outmask = (unsigned)(vmask >> H2);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int a = 0; a < 4; ++a)
                *reinterpret_cast<f32x4*>(es + fr * EPITCH + (a * 16 + 4 * fq) * 4) = acc[a][b];
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int lr = it * 8 + er;
                const int t = m0 + wave * 32 + b * 16 + lr;
                const bool rowok = t < t_end;
                const bool valid = (outmask & (erbit << (b * 16 + it * 8))) != 0u;
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(es + lr * EPITCH + eg * 32);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(es + lr * EPITCH + eg * 32 + 16);
                f32x2 v[4] = {f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}};
                const half2v* hh = reinterpret_cast<const half2v*>(&resv[b][it]);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (v[q] + b2v[q] + __builtin_convertvector(hh[q], f32x2)) * out_scale2;
                if constexpr (ACCMODE == 1) {
                    v[0] += f32x2{accin[b][it][0].x, accin[b][it][0].y}; v[1] += f32x2{accin[b][it][0].z, accin[b][it][0].w};
                    v[2] += f32x2{accin[b][it][1].x, accin[b][it][1].y}; v[3] += f32x2{accin[b][it][1].z, accin[b][it][1].w};
                }
                if constexpr (ACCMODE == 2) {
                    const half2v* ha = reinterpret_cast<const half2v*>(&accin[b][it][0]);
                    const half2v* hb = reinterpret_cast<const half2v*>(&accin[b][it][1]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += __builtin_convertvector(ha[q], f32x2) + __builtin_convertvector(hb[q], f32x2);
                }
                if (o32a) {
                    float* op = rowok ? o32a + (long)t * e.ldo + eco : reinterpret_cast<float*>(trash);
                    *reinterpret_cast<float4*>(op) = valid ? make_float4(v[0][0], v[0][1], v[1][0], v[1][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(op + 4) = valid ? make_float4(v[2][0], v[2][1], v[3][0], v[3][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (has_post) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = lrelu2(v[q], post_slope2);
                }
                if (o32b) {
                    float* op = rowok ? o32b + (long)t * e.ldo + eco : reinterpret_cast<float*>(trash);
                    *reinterpret_cast<float4*>(op) = valid ? make_float4(v[0][0], v[0][1], v[1][0], v[1][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(op + 4) = valid ? make_float4(v[2][0], v[2][1], v[3][0], v[3][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (o16) {
                    uint4 o;
                    half2v* h = reinterpret_cast<half2v*>(&o);
#pragma unroll
                    for (int q = 0; q < 4; ++q) h[q] = __builtin_convertvector(v[q], half2v);
                    o.x = valid ? o.x : 0u; o.y = valid ? o.y : 0u; o.z = valid ? o.z : 0u; o.w = valid ? o.w : 0u;
                    *reinterpret_cast<uint4*>(rowok ? reinterpret_cast<char*>(o16 + (long)t * e.ldo + eco) : trash) = o;
                }
            }
        }
        vmask = vmask_next;
        __syncthreads();
        cur ^= 1;
    }
#undef EV_P64_GLOAD
#undef EV_P64_SSTORE
#undef EV_P64_VROW
#undef EV_P64_VLOAD
#undef EV_P64_VMASK
}

#include "ev_gemm_mx.h"
#include "ev_gemm_mx64.h"
#include "ev_pair_mx.h"
#include "ev_pair_e5.h"

#include "ev_conv64_mx.h"
#include "ev_pair64_mx.h"

int init_device_kernels(int device) {
    if (device < 0 || device >= 64) return -1;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return -1;
    g_n_cu[device] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    hipError_t e = hipSuccess;
    auto attr = [&](const void* fn, size_t bytes) { const hipError_t r = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes); if (r != hipSuccess) e = r; };
#define EV_PAIR_ATTR(KK, TB)                                                             \
    attr((const void*)resblock_pair_c32_kernel<KK, 0, TB>, pair_lds_bytes(KK, TB));      \
    attr((const void*)resblock_pair_c32_kernel<KK, 1, TB>, pair_lds_bytes(KK, TB));      \
    attr((const void*)resblock_pair_c32_kernel<KK, 2, TB>, pair_lds_bytes(KK, TB));
    EV_PAIR_ATTR(3, false) EV_PAIR_ATTR(7, false) EV_PAIR_ATTR(11, false) EV_PAIR_ATTR(3, true)
#undef EV_PAIR_ATTR
    attr((const void*)resblock_pair_c64_kernel<3, 0>, PAIR64_LDS_BYTES);
    attr((const void*)resblock_pair_c64_kernel<3, 1>, PAIR64_LDS_BYTES);
    attr((const void*)resblock_pair_c64_kernel<3, 2>, PAIR64_LDS_BYTES);
    if (x3_set_attributes() != hipSuccess) e = hipErrorInvalidValue;
    if (phased_set_attributes() != hipSuccess) e = hipErrorInvalidValue;
    if (mx_set_attributes() != hipSuccess) e = hipErrorInvalidValue;
    if (mx64_set_attributes() != hipSuccess) e = hipErrorInvalidValue;
    if (pair_mx_set_attributes() != hipSuccess) e = hipErrorInvalidValue;
    if (pair_e5_set_attributes() != hipSuccess) e = hipErrorInvalidValue;
    if (conv64_mx_set_attributes() != hipSuccess) e = hipErrorInvalidValue;
    if (pair64_mx_set_attributes() != hipSuccess) e = hipErrorInvalidValue;
    g_dev_ready[device] = (e == hipSuccess);
    return e == hipSuccess ? 0 : -1;
}
int launch_resblock_pair_c32_mx(const ResPairParams& p, hipStream_t s) {
    const ConvGemmParams& e = p.epi;
    if (!p.w1_mx || !p.w2_mx || !e.out32 || e.out16 || e.add16_a || e.post_lrelu || e.seq_bias || e.mxo_h || !(p.k == 3 || p.k == 7 || p.k == 11) ||
        (p.k - 1) / 2 * (p.dil + 1) * 2 + 256 > 320 || p.ldx != 32) return -1;
    const int n_cu = device_cus();
    const int h2 = (p.k - 1) / 2;
    // (round 6) E5M2 activation operands in the cross terms, no block maxima (ev_pair_e5.h), where they pay: k = 3, the pairs whose time is the quantisers' and the
    // tile's fixed cost (-8 ... -12 % per launch, -4 % with the accumulate-in; more accurate than fp4 on every shape).  At k = 7 / 11 the same kernel measures +0 ... +5 %:
    // those launches are bound by their conv phases (one LDS fragment read per MFMA), and the bf8 operand doubles the B-fragment bytes and runs the block-scaled MFMA
    // at the fp8 rate (profiles/r6_b_pair_e5_ab.txt) -- they keep the fp4 kernel.  The rule is on the layer's shape only, for EVERY M: an utterance gets the same
    // bits alone and in a batch.  epi.reserved0 bit 4 (16): fp4 everywhere (ev_config.mx_act_format = 1); bit 5 (32): E5M2 at every k (tools/bench_pair_mx.py).
    if (!(e.reserved0 & 16) && (p.k == 3 || (e.reserved0 & 32))) {
        const int bmo2 = 128 - 2 * h2, ntiles2 = (p.M + bmo2 - 1) / bmo2;
        const int grid2 = (ntiles2 + 1) / 2 < n_cu ? (ntiles2 + 1) / 2 : n_cu;
#define EV_PE5_LAUNCH(KK)                                                                                                           \
        if (e.acc32) hipLaunchKernelGGL((resblock_pair_c32_e5_kernel<KK, 1>), dim3(grid2), dim3(512), PairE5Geom<KK>::TOTAL, s, p);   \
        else hipLaunchKernelGGL((resblock_pair_c32_e5_kernel<KK, 0>), dim3(grid2), dim3(512), PairE5Geom<KK>::TOTAL, s, p);
        switch (p.k) {
            case 3: EV_PE5_LAUNCH(3) break;
            case 7: EV_PE5_LAUNCH(7) break;
            default: EV_PE5_LAUNCH(11) break;
        }
#undef EV_PE5_LAUNCH
        return 0;
    }
    // two-group schedule (resblock_pair_c32_mx2_kernel: 128-row tiles, one per 4-wave group) once there are enough tiles to keep every group of every
    // CU busy; epi.reserved0 bit 2: in-process A/B (lock-step kernel).  Which kernel a launch takes depends on M, the result does not (bit-identical).
    {
        const int bmo2 = 128 - 2 * h2, ntiles2 = (p.M + bmo2 - 1) / bmo2;
        if (!(e.reserved0 & 4) && ntiles2 >= 2) {
            const int grid2 = (ntiles2 + 1) / 2 < n_cu ? (ntiles2 + 1) / 2 : n_cu;
#define EV_PMX2_LAUNCH(KK)                                                                                                          \
            if (e.acc32) hipLaunchKernelGGL((resblock_pair_c32_mx2_kernel<KK, 1>), dim3(grid2), dim3(512), PairMx2Geom<KK>::TOTAL, s, p);   \
            else hipLaunchKernelGGL((resblock_pair_c32_mx2_kernel<KK, 0>), dim3(grid2), dim3(512), PairMx2Geom<KK>::TOTAL, s, p);
            switch (p.k) {
                case 3: EV_PMX2_LAUNCH(3) break;
                case 7: EV_PMX2_LAUNCH(7) break;
                default: EV_PMX2_LAUNCH(11) break;
            }
#undef EV_PMX2_LAUNCH
            return 0;
        }
    }
    const int bmo = 256 - 2 * h2;
    const int ntiles = (p.M + bmo - 1) / bmo;
    const int grid = ntiles < n_cu ? ntiles : n_cu;
#define EV_PMX_LAUNCH(KK)                                                                                                         \
        if (e.acc32) hipLaunchKernelGGL((resblock_pair_c32_mx_kernel<KK, 1>), dim3(grid), dim3(512), PairMxGeom<KK>::TOTAL, s, p);   \
        else hipLaunchKernelGGL((resblock_pair_c32_mx_kernel<KK, 0>), dim3(grid), dim3(512), PairMxGeom<KK>::TOTAL, s, p);
    switch (p.k) {
        case 3: EV_PMX_LAUNCH(3) break;
        case 7: EV_PMX_LAUNCH(7) break;
        default: EV_PMX_LAUNCH(11) break;
    }
#undef EV_PMX_LAUNCH
    return 0;
}

void launch_resblock_pair_c64(const ResPairParams& p, hipStream_t s) {
    const size_t bytes = PAIR64_LDS_BYTES;
    const int n_cu = device_cus();
    const int bmo = 256 - 2;
    const int ntiles = (p.M + bmo - 1) / bmo;
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    if (p.epi.add16_a) hipLaunchKernelGGL((resblock_pair_c64_kernel<3, 2>), dim3(grid), dim3(512), bytes, s, p);
    else if (p.epi.acc32) hipLaunchKernelGGL((resblock_pair_c64_kernel<3, 1>), dim3(grid), dim3(512), bytes, s, p);
    else hipLaunchKernelGGL((resblock_pair_c64_kernel<3, 0>), dim3(grid), dim3(512), bytes, s, p);
}

int mx_launch_kind(const ConvGemmParams& p) {
    if (p.dtype != DT_MX) return 0;
    return (mx64_eligible(p) || conv64_mx_eligible(p)) ? 2 : (mx_eligible(p) ? 1 : 0);
}

int splitk_check(const ConvGemmParams& p) {
    if (p.ksplit <= 1) return 0;
    if (p.dtype != DT_F32S || p.ksplit > 16 || p.N % 64 || p.K % 32 || (p.K / 32) % p.ksplit || p.M % 128 || p.add16_a || (!p.out32 && !p.out16)) return -1;
    if (!p.mx_scratch || ((uintptr_t)p.mx_scratch & 15) || p.mx_scratch_size < (size_t)p.ksplit * (size_t)p.M * (size_t)p.N * 4) return -1;
    return 0;
}

void launch_conv_gemm(const ConvGemmParams& p, hipStream_t s) {
    // preconditions are checked by the engine (ev_engine.cpp: check_gemm)
    (void)device_cus();          // per-device large-LDS opt-in for the per-kernel entry points that run without a handle
    if (p.dtype == DT_F16) launch_dt<_Float16>(p, s);
    else if (p.dtype == DT_F32S) launch_split(p, s);
    else if (p.dtype == DT_MX) {
        if (mx64_eligible(p)) launch_mx64(p, s);
        else if (conv64_mx_eligible(p)) launch_conv64_mx(p, s);
        else if (mx_eligible(p)) launch_mx(p, s);
        else { ConvGemmParams q = p; q.dtype = DT_F32S; launch_split(q, s); }      // same operands, three fp16 MFMAs per product
    } else launch_dt<float>(p, s);
}

}  // namespace ev
