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
// Staged rows are 64 B of data + 16 B pad (80 B pitch) so that the 16 rows of a ds_read_b128
// fragment read hit distinct 16-byte LDS slots.
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

#include "ev_kernels.h"

namespace ev {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int ROWB = 80;   // LDS pitch of one staged row: 64 B payload + 16 B pad
static constexpr int MAX_SPAN = 64;

__device__ __forceinline__ uint4 lrelu_h8(uint4 v, float slope) {
    half8 h = *reinterpret_cast<half8*>(&v);
    const _Float16 s = (_Float16)slope;
    h = __builtin_elementwise_max(h, h * s);   // slope in (0,1): max(x, slope*x) == leaky_relu(x)
    return *reinterpret_cast<uint4*>(&h);
}
__device__ __forceinline__ uint4 lrelu_f4(uint4 v, float slope) {
    float* f = reinterpret_cast<float*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = fmaxf(f[i], f[i] * slope);
    return v;
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

template <typename TIn, int BM, int BN, int WT, int WC>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams p) {
    constexpr int ES = sizeof(TIn);
    constexpr int TT = BM / WT, TC = BN / WC, MT = TT / 16, NT = TC / 16;
    constexpr int XCH = ((BM + MAX_SPAN) * 4 + 255) / 256;   // 16-B chunks per thread for the X slab
    constexpr int WCH = (BN * 4 + 255) / 256;
    static_assert(WT * WC == 4, "4 waves per block");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wt = wave / WC, wc = wave % WC;

    const int span = (p.taps - 1) * p.dil;
    const int slab_rows = BM + span;
    const int xbuf_bytes = slab_rows * ROWB;
    char* Xs = smem;
    char* Ws = smem + 2 * xbuf_bytes;

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

    const int nkc = (p.K * ES) >> 6;
    const int steps = nkc * p.taps;
    const char* Ag = reinterpret_cast<const char*>(p.A);
    const char* Wg = reinterpret_cast<const char*>(p.W);
    const long a_pitch = (long)p.lda * ES;
    const long w_tap_pitch = (long)p.K * ES;
    const long w_row_pitch = w_tap_pitch * p.taps;
    const int x_chunks = slab_rows * 4;
    const long a_row0 = (long)m0 - (long)p.center * p.dil;

    uint4 xr[XCH], wr[WCH];

    auto gload_x = [&](int kc) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = tid + i * 256;
            if (c < x_chunks) {
                const int r = c >> 2, part = c & 3;
                xr[i] = *reinterpret_cast<const uint4*>(Ag + (a_row0 + r) * a_pitch + (long)kc * 64 + part * 16);
            }
        }
    };
    auto sstore_x = [&](int buf) {
#pragma unroll
        for (int i = 0; i < XCH; ++i) {
            const int c = tid + i * 256;
            if (c < x_chunks) {
                const int r = c >> 2, part = c & 3;
                uint4 v = xr[i];
                if (p.pro_lrelu) v = (ES == 2) ? lrelu_h8(v, p.pro_slope) : lrelu_f4(v, p.pro_slope);
                *reinterpret_cast<uint4*>(Xs + buf * xbuf_bytes + r * ROWB + part * 16) = v;
            }
        }
    };
    auto gload_w = [&](int kc, int tap) {
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + i * 256;
            if (c < BN * 4) {
                const int r = c >> 2, part = c & 3;
                wr[i] = *reinterpret_cast<const uint4*>(Wg + (long)(n0 + r) * w_row_pitch + (long)tap * w_tap_pitch +
                                                        (long)kc * 64 + part * 16);
            }
        }
    };
    auto sstore_w = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WCH; ++i) {
            const int c = tid + i * 256;
            if (c < BN * 4) {
                const int r = c >> 2, part = c & 3;
                *reinterpret_cast<uint4*>(Ws + buf * (BN * ROWB) + r * ROWB + part * 16) = wr[i];
            }
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    gload_x(0);
    gload_w(0, 0);
    sstore_x(0);
    sstore_w(0);
    __syncthreads();

    const int frag_off = (lane & 15) * ROWB + (lane >> 4) * 16;
    int kc = 0, tap = 0;
    for (int s = 0; s < steps; ++s) {
        int ntap = tap + 1, nkc_ = kc;
        if (ntap == p.taps) { ntap = 0; nkc_ = kc + 1; }
        const bool has_next = (s + 1 < steps);
        const bool next_x = has_next && (ntap == 0);
        if (has_next) gload_w(nkc_, ntap);
        if (next_x) gload_x(nkc_);

        const char* Xb = Xs + (kc & 1) * xbuf_bytes + (wt * TT + tap * p.dil) * ROWB + frag_off;
        const char* Wb = Ws + (s & 1) * (BN * ROWB) + (wc * TC) * ROWB + frag_off;
        uint4 xf[MT], wf[NT];
#pragma unroll
        for (int b = 0; b < MT; ++b) xf[b] = *reinterpret_cast<const uint4*>(Xb + b * 16 * ROWB);
#pragma unroll
        for (int a = 0; a < NT; ++a) wf[a] = *reinterpret_cast<const uint4*>(Wb + a * 16 * ROWB);
#pragma unroll
        for (int a = 0; a < NT; ++a) {
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                if constexpr (ES == 2) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wf[a]),
                                                                       *reinterpret_cast<half8*>(&xf[b]), acc[a][b], 0, 0, 0);
                } else {
                    const float* wa = reinterpret_cast<const float*>(&wf[a]);
                    const float* xb = reinterpret_cast<const float*>(&xf[b]);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], xb[e], acc[a][b], 0, 0, 0);
                }
            }
        }
        if (has_next) sstore_w((s + 1) & 1);
        if (next_x) sstore_x(nkc_ & 1);
        __syncthreads();
        tap = ntap;
        kc = nkc_;
    }

    // ---- epilogue: lane holds D[co0..co0+3][t] for each (a, b) tile
    const int t_base = m0 + wt * TT + (lane & 15);
    const int co_base = n0 + wc * TC + 4 * (lane >> 4);
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int t = t_base + b * 16;
        const bool valid = p.row_valid ? (p.row_valid[t >> p.valid_shift] != 0) : true;
        const float* sb = nullptr;
        if (p.seq_bias && valid) sb = p.seq_bias + (long)p.row_seq[t] * p.ld_seq_bias;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int co = co_base + a * 16;
            float v[4] = {acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]};
            if (valid) {
                if (p.bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(p.bias + co);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (p.act != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], p.act, p.act_slope);
                }
                if (sb) {
                    const float4 bb = *reinterpret_cast<const float4*>(sb + co);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (p.res) {
                    if (p.res_dtype == DT_F16) {
                        const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const __half*>(p.res) + (long)t * p.ldres + co);
                        const __half2* h = reinterpret_cast<const __half2*>(&rr);
                        const float2 f0 = __half22float2(h[0]), f1 = __half22float2(h[1]);
                        v[0] += f0.x; v[1] += f0.y; v[2] += f1.x; v[3] += f1.y;
                    } else {
                        const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + (long)t * p.ldres + co);
                        v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= p.out_scale;
                if (p.acc32) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.acc32 + (long)t * p.ldacc + co);
                    v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                }
                if (p.out32 && p.out32_before_post)
                    *reinterpret_cast<float4*>(p.out32 + (long)t * p.ldo + co) = make_float4(v[0], v[1], v[2], v[3]);
                if (p.post_lrelu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * p.post_slope;
                }
            } else {
                v[0] = v[1] = v[2] = v[3] = 0.f;
                if (p.out32 && p.out32_before_post)
                    *reinterpret_cast<float4*>(p.out32 + (long)t * p.ldo + co) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (p.out32 && !p.out32_before_post)
                *reinterpret_cast<float4*>(p.out32 + (long)t * p.ldo + co) = make_float4(v[0], v[1], v[2], v[3]);
            if (p.out16) {
                uint2 o;
                __half2* h = reinterpret_cast<__half2*>(&o);
                h[0] = __floats2half2_rn(v[0], v[1]);
                h[1] = __floats2half2_rn(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<__half*>(p.out16) + (long)t * p.ldo + co) = o;
            }
        }
    }
}

template <typename TIn, int BM, int BN, int WT, int WC>
static void launch_cfg(const ConvGemmParams& p, hipStream_t s) {
    const int span = (p.taps - 1) * p.dil;
    const size_t lds = 2 * (size_t)(BM + span) * ROWB + 2 * (size_t)BN * ROWB;
    const int grid = (p.M / BM) * (p.N / BN);
    hipLaunchKernelGGL((conv_gemm_kernel<TIn, BM, BN, WT, WC>), dim3(grid), dim3(256), lds, s, p);
}

template <typename TIn>
static void launch_dt(const ConvGemmParams& p, hipStream_t s) {
    if (p.N % 128 == 0) launch_cfg<TIn, 128, 128, 2, 2>(p, s);
    else if (p.N % 64 == 0) launch_cfg<TIn, 256, 64, 4, 1>(p, s);
    else launch_cfg<TIn, 256, 32, 4, 1>(p, s);
}

void launch_conv_gemm(const ConvGemmParams& p, hipStream_t s) {
    // preconditions are checked by the engine (ev_engine.cpp: check_gemm)
    if (p.dtype == DT_F16) launch_dt<_Float16>(p, s);
    else launch_dt<float>(p, s);
}

}  // namespace ev
