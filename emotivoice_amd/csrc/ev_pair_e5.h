// Fused HiFi-GAN ResBlock pair at C = 32 (stage 3) with MAXIMA-FREE cross-term operands (round 6):
//
//        xt  = leaky_relu(c1(leaky_relu(x, .1)) + b1, .1)          c1 = Conv1d(32, 32, k, dilation d)
//        out = epilogue(c2(xt) + b2 + x)                            c2 = Conv1d(32, 32, k, dilation 1)       (models/hifigan/models.py:50-57)
//
// The structure is resblock_pair_c32_mx2_kernel's (ev_pair_mx.h: persistent 8-wave block, two 4-wave groups one barrier apart, both convs' weights
// stationary in LDS, x fp32 in / out, xt never leaves LDS) and so is the product: x.w = xh.wh (fp16 MFMA) + Q(wl).Q(xh) + Q(wh).Q(xl) (block-scaled MFMAs,
// K = 128 = four taps x 32 channels).  What changed is the ACTIVATION side of the two cross terms.  Round 5 found this kernel bound by the SIMDs' issue ports:
// 452 of its 725 instructions per tile and wave were VALU, most of them the block-scaled fp4 quantiser (block maxima over 32 channels -- cross-lane --, scale
// bytes, conversions: ~70 VALU per 8 values, twice per element and pair).  Here
//        Q(xh) = the TOP BYTE of the fp16 hi part: OCP E5M2 by truncation, one v_perm_b32 per four values, no scale (E8M0 127);
//        Q(xl) = E5M2 of xl 2^11 at the CONSTANT block scale 2^-11 (E8M0 116): |x - fp16(x)| <= 2^-11 2^e, so the scaled remainder of every fp16-normal x is inside
//                E5M2's normal range -- one v_cvt_scalef32_pk_bf8_f32 per two values;
// no block maximum, no scale plane, no cross-lane traffic: ~22 VALU per 8 values.  Per-element exponents and two mantissa bits make both operands MORE accurate
// than block-scaled e2m1 (tools/precision_study_mx.py: zero-mean recipe 4.22e-4 -> 4.15e-4 with stage 3 alone, trained-like recipe 5.31e-4 -> 5.15e-4).  The
// weights keep their block-scaled fp4 planes (A operand, `cbsz:4`), the activations are the B operand in bf8 (`blgp:1`): the instruction then runs at the fp8
// rate (tools/mfma_ubench.hip: 4.6 instead of 7.3 PF/s; the 64 f16 + 32 block-scaled mix 1.04 instead of 1.17 PF/s) -- affordable where MFMA-busy is 0.27.
//
// Operand layout of an 8-BIT operand of v_mfma_scale_f32_16x16x128_f8f6f4 (measured: tools/mx_layout_probe.hip, profiles/r6_a_mx_layout_probe.txt): lane group
// j = lane >> 4 holds K elements 16 j .. 16 j + 15 in registers 0-3 and 64 + 16 j .. 64 + 16 j + 15 in registers 4-7 (NOT 32 consecutive ones like fp4), i.e. with
// K = 128 = [tap 4 g + q][32 channels]: registers 0-3 = tap 4 g + (j >> 1), registers 4-7 = tap 4 g + 2 + (j >> 1), each the 16-channel half j & 1.  The code
// planes are therefore kept as two 16-byte-pitch half planes [half][row][16]: a fragment is two ds_read_b128 of the lane's half at its two taps' rows, and the
// lanes a ds_read_b128 serves together (rows r .. r + 3, r + 12 .. r + 15 of one half, r + 4 .. r + 11 of the other) fall on 64 distinct banks.
// The accumulate chains f16 K = 32 <-> fp4 x bf8 block-scaled are exact at 0 wait states in both orders (tools/mfma_chain_check.hip, profiles/r6_a_mfma_chain_check.txt).
//
// LDS: the 32-byte code rows make the k = 11 footprint 157 KB without the epilogue's transposing scratch (18 KB), so that instantiation keeps the MFMA C layout
// for its residual loads and fp32 stores (a lane = 4 consecutive channels of a row; measured equal to the fp4 kernel there); k = 3 / 7 transpose as before -- in the
// C layout a quarter wave touches 16 rows instead of 4, which measured 6 % SLOWER on the accumulate-in launches (profiles/r6_b_pair_e5_ab.txt).
#pragma once

typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

// A = 16 output channels x 128 K as block-scaled fp4 (16 bytes per lane, byte 0 of sa = the lane's E8M0 scale), B = 16 time steps x 128 K as E5M2 (32 bytes per
// lane, layout above), sb = its E8M0 scale (the same for every block here).  In place, as mfma_mx_inplace.
__device__ __forceinline__ void mfma_e5_inplace(f32x4& c, const uint4& a, const u32x8& b, int sa, int sb) {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:1"
                 : "+v"(c) : "v"(*reinterpret_cast<const u32x4*>(&a)), "v"(b), "v"(sa), "v"(sb));
}

static constexpr int E5_SCALE_HI = 127, E5_SCALE_LO = 127 - 11;

// 8 consecutive channels a[0..3] (packed pairs) -> ho = their fp16 hi parts, ch = the E5M2 codes of the hi parts (their top bytes), cl = the E5M2 codes of
// (a - hi) 2^11.  Host statement: emotivoice_amd/mxfp4.py (e5m2_hi_codes / e5m2_lo_codes).
__device__ __forceinline__ void e5_quant8(const f32x2 (&a)[4], uint4& ho, uint2& ch, uint2& cl) {
    typedef short s16x2 __attribute__((ext_vector_type(2)));
    half2v hh[4];
    f32x2 lf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        hh[j] = __builtin_convertvector(a[j], half2v);
        lf[j] = a[j] - __builtin_convertvector(hh[j], f32x2);
    }
    ho.x = *reinterpret_cast<unsigned*>(&hh[0]); ho.y = *reinterpret_cast<unsigned*>(&hh[1]);
    ho.z = *reinterpret_cast<unsigned*>(&hh[2]); ho.w = *reinterpret_cast<unsigned*>(&hh[3]);
    ch.x = __builtin_amdgcn_perm(ho.y, ho.x, 0x07050301u);
    ch.y = __builtin_amdgcn_perm(ho.w, ho.z, 0x07050301u);
    s16x2 c0 = {0, 0}, c1 = {0, 0};
    c0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(c0, lf[0][0], lf[0][1], 0x1p-11f, false);
    c0 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(c0, lf[1][0], lf[1][1], 0x1p-11f, true);
    c1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(c1, lf[2][0], lf[2][1], 0x1p-11f, false);
    c1 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(c1, lf[3][0], lf[3][1], 0x1p-11f, true);
    cl.x = *reinterpret_cast<unsigned*>(&c0); cl.y = *reinterpret_cast<unsigned*>(&c1);
}

// EV_PAIR_TIMING (tuning builds: build.py --variant ptime EV_PAIR_TIMING; tools/bench_pair_mx.py --timing): every wave sums s_memtime deltas per part of an iteration
// and lane 0 writes the nine sums + a marker to ((unsigned*)p.epi.row_seq)[(block * 8 + wave) * 16 + k] at the end: k = 0 issue of the iteration's global loads, 1 conv1,
// 2 xt quantiser + stores, 3 barrier (xt complete), 4 next slab -> LDS (includes the wait for its loads), 5 conv2, 6 epilogue, 7 barrier (slab complete), 8 prologue
#ifdef EV_PAIR_TIMING
#define EV_PT_DECL unsigned long long pt_t = __builtin_readcyclecounter(); unsigned pt_s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define EV_PT(K_) { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_readcyclecounter(); pt_s[K_] += (unsigned)(n_ - pt_t); pt_t = n_; __builtin_amdgcn_sched_barrier(0); }
#define EV_PT_DUMP if (lane == 0 && p.epi.row_seq) { unsigned* o_ = reinterpret_cast<unsigned*>(const_cast<int32_t*>(p.epi.row_seq)) + ((size_t)blockIdx.x * 8 + wave) * 16; \
        for (int k_ = 0; k_ < 9; ++k_) o_[k_] = pt_s[k_]; o_[15] = 0xC0FFEEu; }
#else
#define EV_PT_DECL
#define EV_PT(K_)
#define EV_PT_DUMP
#endif

template <int K>
struct PairE5Geom {
    static constexpr int C = 32, H2 = (K - 1) / 2, GR = 128, BMO = GR - 2 * H2, KG = (K + 3) / 4, KP = KG * 4;
    static constexpr int XR = 192, TR = 144;
    static constexpr int WHB = K * C * 64, WQB = KP * C * 16, WSB = KP * C;
    static constexpr int OFF_WH = 0, OFF_WQ = 2 * WHB, OFF_WS = OFF_WQ + 4 * WQB, OFF_G = OFF_WS + 4 * WSB;
    // one group's slab / xt: fp16 hi plane [rows][64 B, swizzled], code planes [plane][half][rows][16 B]
    static constexpr int XHS = XR * 16, XPS = 2 * XHS, THS = TR * 16, TPS = 2 * THS;
    static constexpr int G_XH = 0, G_XQ = XR * 64, G_TH = G_XQ + 2 * XPS, G_TQ = G_TH + TR * 64;
    static constexpr int GB = G_TQ + 2 * TPS;
    // the epilogue transposes through 16 rows of LDS scratch per wave (a lane then owns 8 consecutive channels of a row: 32-byte accesses, 4 lanes per 128-byte
    // row -- a quarter wave touches 4 rows, not 16) wherever that scratch fits; k = 11 (157 KB without it) keeps the MFMA C layout for its loads and stores
    static constexpr int EPITCH = C * 4 + 16, OFF_ES = OFF_G + 2 * GB;
    static constexpr bool TRANSPOSED_EPI = OFF_ES + 8 * 16 * EPITCH <= 160 * 1024;
    static constexpr int TOTAL = OFF_ES + (TRANSPOSED_EPI ? 8 * 16 * EPITCH : 0);
    static_assert(OFF_G % 256 == 0 && GB % 256 == 0 && G_XQ % 256 == 0 && G_TH % 256 == 0 && G_TQ % 256 == 0 && XHS % 256 == 0 && THS % 256 == 0 &&
                  TOTAL <= 160 * 1024 && GR + K - 1 <= TR, "LDS plan");
};

// ACCMODE: 0 = none, 1 = fp32 accumulate-in (epi.acc32, may alias epi.out32: the running MRF sum)
template <int K, int ACCMODE>
__global__ __launch_bounds__(512, 1) void resblock_pair_c32_e5_kernel(const ResPairParams p) {
    using G = PairE5Geom<K>;
    constexpr int C = G::C, H2 = G::H2, BMO = G::BMO, KG = G::KG, EPITCH = G::EPITCH;
    constexpr bool TEPI = G::TRANSPOSED_EPI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const W1h = smem + G::OFF_WH;
    char* const W2h = W1h + G::WHB;
    char* const Wq = smem + G::OFF_WQ;        // [conv][plane][KP][32][16]
    char* const Wsc = smem + G::OFF_WS;       // [conv][plane][KP][32]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave >> 2, lw = wave & 3, ltid = tid & 255;
    char* const gb = smem + G::OFF_G + gi * G::GB;
    char* const Xh = gb + G::G_XH;
    char* const Xq = gb + G::G_XQ;        // [plane][half][XR][16]
    char* const Th = gb + G::G_TH;
    char* const Tq = gb + G::G_TQ;        // [plane][half][TR][16]
    char* const es = smem + G::OFF_ES + wave * 16 * EPITCH;          // (TEPI only)
    const int fr = lane & 15, fq = lane >> 4;
    const int dil = p.dil, h1 = H2 * dil;
    const int x_pitch = p.ldx * 4;
    const char* xg = reinterpret_cast<const char*>(p.x);
    const int ntiles = (p.M + BMO - 1) / BMO;
    const ConvGemmParams& e = p.epi;
    const int gmin = p.gmax ? p.gmin : 0, gmax = p.gmax ? p.gmax : p.M;

    // ---- weights -> LDS, once per block (all 512 threads)
    for (int c = tid; c < K * C * 4; c += 512) {
        const int row = c >> 2, part = c & 3, tap = row >> 5, co = row & 31;
        const long off = ((long)(co * K + tap) * C) * 2 + part * 16;
        *reinterpret_cast<uint4*>(W1h + swz(row, part)) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w1) + off);
        *reinterpret_cast<uint4*>(W2h + swz(row, part)) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w2) + off);
    }
    for (int c = tid; c < 2 * G::WQB / 16; c += 512) {
        *reinterpret_cast<uint4*>(Wq + c * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w1_mx) + c * 16);
        *reinterpret_cast<uint4*>(Wq + 2 * G::WQB + c * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w2_mx) + c * 16);
    }
    for (int c = tid; c < 2 * G::WSB / 16; c += 512) {
        *reinterpret_cast<uint4*>(Wsc + c * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w1_mx) + 2 * G::WQB + c * 16);
        *reinterpret_cast<uint4*>(Wsc + 2 * G::WSB + c * 16) = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.w2_mx) + 2 * G::WQB + c * 16);
    }
    // the MFMA C layout: lane (fr, fq) holds channels a * 16 + 4 fq .. + 3 of time row fr (of the 16-row tile b)
    f32x2 b1v[2][2], b2v[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int co = a * 16 + 4 * fq + 2 * q;
            b1v[a][q] = f32x2{p.b1[co], p.b1[co + 1]};
            b2v[a][q] = e.bias ? f32x2{e.bias[co], e.bias[co + 1]} : f32x2{0.f, 0.f};
        }
    // the transposed side of the epilogue (TEPI): 4 lanes per row, 16 rows per instruction, 8 channels per lane
    const int er = lane >> 2, eg = lane & 3, eco = eg * 8;
    const unsigned erbit = 1u << er;
    f32x2 b2t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b2t[q] = e.bias ? f32x2{e.bias[eco + 2 * q], e.bias[eco + 2 * q + 1]} : f32x2{0.f, 0.f};
    const unsigned frbit = 1u << fr;
    const f32x2 out_scale2 = f32x2{e.out_scale, e.out_scale};
    const f32x2 slope01 = f32x2{0.1f, 0.1f};
    float* const o32 = e.out32;
    char* const trash = g_store_trash + lane * 64;
    const uint8_t* vptr = e.row_valid ? e.row_valid : g_row_always_valid;
    const int vshift = e.row_valid ? e.valid_shift : 31;
#define EV_PE5_VROW(TILE) ((TILE) * BMO - H2 + lw * 32 + lane)
#define EV_PE5_VLOAD(TILE, DST) { const int g_ = EV_PE5_VROW(TILE); DST = vptr[min(max(g_, gmin), gmax - 1) >> vshift]; }
#define EV_PE5_VMASK(TILE, SRC) __builtin_amdgcn_ballot_w64((SRC) != 0 && EV_PE5_VROW(TILE) >= gmin && EV_PE5_VROW(TILE) < gmax)

    // ---- slab staging of a group: a thread owns three (row, 8-channel quarter) units of the 192-row slab: rows (ltid >> 2) + {0, 64, 128}; the third one only in
    // the waves whose rows the convs' span reaches (ev_pair_mx.h, round 5)
    float4 xr[3][2];
    const int xq = ltid & 3;
    const char* const xgt = xg + xq * 32;
    const int xrow2 = min((ltid >> 2) + 128, G::GR - 1 + 2 * h1 + 2 * H2);
    const int drow[3] = {ltid >> 2, (ltid >> 2) + 64, (ltid >> 2) + 128};
    const bool need3 = lw * 16 < 2 * (h1 + H2);
    const int xq_off = (xq >> 1) * G::XHS + (xq & 1) * 8;          // this quarter's 8 code bytes inside a row of its half plane
#define EV_PE5_ROW(G_) min((G_), gmax + 63)
#define EV_PE5_GLOAD(TILE)                                                                                 \
    {                                                                                                      \
        const int g0_ = (TILE) * BMO - H2 - h1 + (ltid >> 2);                                              \
        const int g2_ = (TILE) * BMO - H2 - h1 + xrow2;                                                    \
        const char* q0_ = xgt + (long)EV_PE5_ROW(g0_) * x_pitch;                                           \
        const char* q1_ = xgt + (long)EV_PE5_ROW(g0_ + 64) * x_pitch;                                      \
        const char* q2_ = xgt + (long)EV_PE5_ROW(g2_) * x_pitch;                                           \
        xr[0][0] = *reinterpret_cast<const float4*>(q0_); xr[0][1] = *reinterpret_cast<const float4*>(q0_ + 16); \
        xr[1][0] = *reinterpret_cast<const float4*>(q1_); xr[1][1] = *reinterpret_cast<const float4*>(q1_ + 16); \
        if (need3) { xr[2][0] = *reinterpret_cast<const float4*>(q2_); xr[2][1] = *reinterpret_cast<const float4*>(q2_ + 16); } \
    }
    // leaky_relu(x, .1) of models.py:51, then the operand planes of the slab
#define EV_PE5_SSTORE()                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) if (i < 2 || need3) {                                    \
        f32x2 a_[4] = {lrelu2(f32x2{xr[i][0].x, xr[i][0].y}, slope01), lrelu2(f32x2{xr[i][0].z, xr[i][0].w}, slope01), \
                       lrelu2(f32x2{xr[i][1].x, xr[i][1].y}, slope01), lrelu2(f32x2{xr[i][1].z, xr[i][1].w}, slope01)}; \
        uint4 ho_; uint2 ch_, cl_;                                                                         \
        e5_quant8(a_, ho_, ch_, cl_);                                                                      \
        *reinterpret_cast<uint4*>(Xh + swz(drow[i], xq)) = ho_;                                            \
        *reinterpret_cast<uint2*>(Xq + xq_off + drow[i] * 16) = ch_;                                       \
        *reinterpret_cast<uint2*>(Xq + G::XPS + xq_off + drow[i] * 16) = cl_;                              \
    }
    // one conv of the pair on the wave's 32 rows: fp16 hi x hi tap by tap, then the two cross terms four taps at a time.
    // XH: the operand's fp16 plane (64-byte rows), XQ: its code planes (plane stride PS, half stride HS), WH: the conv's fp16 weights, CONV: 0 / 1
#define EV_PE5_CONV(XH, XQ, PS, HS, WH, CONV, DIL)                                                         \
    {                                                                                                      \
        _Pragma("unroll") for (int t = 0; t < K; ++t) {                                                    \
            const int r0 = wrow0 + t * (DIL);                                                              \
            const char* xp = (XH) + r0 * 64 + ((fq ^ ((r0 >> 1) & 3)) << 4);                               \
            uint4 wf_[2];                                                                                  \
            _Pragma("unroll") for (int a = 0; a < 2; ++a) wf_[a] = *reinterpret_cast<const uint4*>((WH) + swz(t * 32 + a * 16 + fr, fq)); \
            _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                                \
                uint4 xf_ = *reinterpret_cast<const uint4*>(xp + b * 16 * 64);                             \
                _Pragma("unroll") for (int a = 0; a < 2; ++a)                                              \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wf_[a]), *reinterpret_cast<half8*>(&xf_), acc[a][b], 0, 0, 0); \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < KG; ++g) {                                                   \
            const int tw = 4 * g + fq;                     /* weights (fp4): this lane's K block = tap tw (zero codes beyond K) */ \
            const int ta = 4 * g + (fq >> 1);              /* activations (bf8): registers 0-3 = tap ta, 4-7 = tap ta + 2, channel half fq & 1 */ \
            const char* xa_ = (XQ) + (fq & 1) * (HS) + (wrow0 + min(ta, K - 1) * (DIL)) * 16;               \
            const char* xb_ = (XQ) + (fq & 1) * (HS) + (wrow0 + min(ta + 2, K - 1) * (DIL)) * 16;           \
            _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                             \
                uint4 wq_[2];                                                                              \
                u32x8 xq_[2];                                                                              \
                int ws_[2];                                                                                \
                _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                            \
                    wq_[a] = *reinterpret_cast<const uint4*>(Wq + ((CONV) * 2 + pl) * G::WQB + (tw * 32 + a * 16 + fr) * 16); \
                    ws_[a] = *reinterpret_cast<const uint8_t*>(Wsc + ((CONV) * 2 + pl) * G::WSB + tw * 32 + a * 16 + fr); \
                }                                                                                          \
                _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                            \
                    const u32x4 lo_ = *reinterpret_cast<const u32x4*>(xa_ + pl * (PS) + b * 256);          \
                    const u32x4 hi_ = *reinterpret_cast<const u32x4*>(xb_ + pl * (PS) + b * 256);          \
                    xq_[b] = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                    \
                }                                                                                          \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                              \
                    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                          \
                        mfma_e5_inplace(acc[a][b], wq_[a], xq_[b], ws_[a], pl ? sc_lo : sc_hi);            \
            }                                                                                              \
        }                                                                                                  \
        mfma_asm_fence(acc);          /* the quantiser / the epilogue read the accumulators next */       \
    }
    // block barrier of the main loop: LDS traffic retired, vmcnt left alone (see conv_c64_mx2_kernel)
#define EV_PE5_GROUP_BARRIER()                                   \
    {                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        __builtin_amdgcn_sched_barrier(0);                       \
    }
    int sc_hi = E5_SCALE_HI, sc_lo = E5_SCALE_LO;
    asm volatile("" : "+v"(sc_hi), "+v"(sc_lo));          // (the MFMA's scale operands are VGPRs: materialised once)

    // tile stream of a group: tiles 2 (b + n grid) + gi; both groups run the same number of iterations (a tile index beyond the last recomputes
    // the last tile and stores nothing)
    const int niter = (ntiles + 2 * (int)gridDim.x - 1) / (2 * (int)gridDim.x);
    int tq = 2 * (int)blockIdx.x + gi;
    int tile = min(tq, ntiles - 1);
    unsigned long long vmask;
    EV_PT_DECL
    {
        uint8_t vb;
        EV_PE5_GLOAD(tile)
        EV_PE5_VLOAD(tile, vb)
        EV_PE5_SSTORE()
        vmask = EV_PE5_VMASK(tile, vb);
    }
    __syncthreads();
    if (gi == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0 from here on
    const int wrow0 = lw * 32 + fr;
    EV_PT(8)
    for (int it_ = 0; it_ < niter; ++it_, tq += 2 * (int)gridDim.x) {
        tile = min(tq, ntiles - 1);
        const int next = min(tq + 2 * (int)gridDim.x, ntiles - 1);
        const int m0 = tile * BMO;
        const int t_end = tq < ntiles ? min(m0 + BMO, p.M) : m0;          // (a repeated tile stores nothing)
        // ---------------- memory requests of this iteration, oldest first: raw residual rows (L2 hits: the slab just came through), the accumulate-in rows,
        // then the next tile's slab and row-valid byte
        float4 resv[2][2], accin[2][2];          // TEPI: [16-row pass][16-byte half of the lane's 8 channels]; else [a][b] in the C layout
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int t = max(min(m0 + lw * 32 + b * 16 + (TEPI ? er : fr), t_end - 1), 0);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int co = TEPI ? eco + 4 * a : a * 16 + 4 * fq;
                float4& rd = TEPI ? resv[b][a] : resv[a][b];
                rd = *reinterpret_cast<const float4*>(xg + (long)t * x_pitch + co * 4);
                if constexpr (ACCMODE == 1) {
                    float4& ad = TEPI ? accin[b][a] : accin[a][b];
                    ad = *reinterpret_cast<const float4*>(e.acc32 + (long)t * e.ldacc + co);
                }
            }
        }
        uint8_t vb_next;
        EV_PE5_GLOAD(next)
        EV_PE5_VLOAD(next, vb_next)
        __builtin_amdgcn_sched_barrier(0);
        EV_PT(0)
        f32x4 acc[2][2];
        // ---------------- conv1 (dilation d): the group's 128 rows, global rows m0 - H2 + r1
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        EV_PE5_CONV(Xh, Xq, G::XPS, G::XHS, W1h, 0, dil)
        EV_PT(1)
        // bias + leaky-relu + zero outside the utterance (conv2 must see the reference's zero padding) -> the xt planes: 4 channels of a row per lane and a,
        // i.e. one dword of each code plane -- no cross-lane step
        const unsigned xtmask = (unsigned)vmask;
        const bool xt_masked = xtmask != 0xffffffffu;          // (wave-uniform: only a wave with a row outside the utterances runs the selects)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r1 = wrow0 + b * 16;
            const bool valid = (xtmask & (frbit << (b * 16))) != 0u;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                f32x2 v0 = lrelu2(f32x2{acc[a][b][0], acc[a][b][1]} + b1v[a][0], slope01);
                f32x2 v1 = lrelu2(f32x2{acc[a][b][2], acc[a][b][3]} + b1v[a][1], slope01);
                if (xt_masked) { v0[0] = valid ? v0[0] : 0.f; v0[1] = valid ? v0[1] : 0.f; v1[0] = valid ? v1[0] : 0.f; v1[1] = valid ? v1[1] : 0.f; }
                const half2v h0 = __builtin_convertvector(v0, half2v), h1_ = __builtin_convertvector(v1, half2v);
                const f32x2 l0 = v0 - __builtin_convertvector(h0, f32x2), l1 = v1 - __builtin_convertvector(h1_, f32x2);
                uint2 w;
                w.x = *reinterpret_cast<const unsigned*>(&h0); w.y = *reinterpret_cast<const unsigned*>(&h1_);
                const int co = a * 16 + 4 * fq;
                *reinterpret_cast<uint2*>(Th + swz(r1, co >> 3) + (co & 7) * 2) = w;
                s16x2 cl = {0, 0};
                cl = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(cl, l0[0], l0[1], 0x1p-11f, false);
                cl = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(cl, l1[0], l1[1], 0x1p-11f, true);
                *reinterpret_cast<unsigned*>(Tq + a * G::THS + r1 * 16 + 4 * fq) = __builtin_amdgcn_perm(w.y, w.x, 0x07050301u);
                *reinterpret_cast<unsigned*>(Tq + G::TPS + a * G::THS + r1 * 16 + 4 * fq) = *reinterpret_cast<unsigned*>(&cl);
            }
        }
        EV_PT(2)
        EV_PE5_GROUP_BARRIER()          // every wave of the group is done with the slab; xt is complete
        EV_PT(3)
        EV_PE5_SSTORE()                 // the next tile's slab replaces the current one (conv2 only reads xt)
        const unsigned long long vmask_next = EV_PE5_VMASK(next, vb_next);
        EV_PT(4)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        EV_PE5_CONV(Th, Tq, G::TPS, G::THS, W2h, 1, 1)
        EV_PT(5)
        const unsigned outmask = (unsigned)(vmask >> H2);
        if constexpr (TEPI) {
            // ---------------- epilogue: 16-row passes through the wave's transposing scratch, 32-byte row-contiguous fp32 accesses
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int t = m0 + lw * 32 + it * 16 + er;
                const bool rowok = t < t_end;
                const bool valid = (outmask & (erbit << (it * 16))) != 0u;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int a = 0; a < 2; ++a) *reinterpret_cast<f32x4*>(es + fr * EPITCH + (a * 16 + 4 * fq) * 4) = acc[a][it];
                __builtin_amdgcn_wave_barrier();
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(es + er * EPITCH + eg * 32);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(es + er * EPITCH + eg * 32 + 16);
                f32x2 v[4] = {f32x2{v0[0], v0[1]}, f32x2{v0[2], v0[3]}, f32x2{v1[0], v1[1]}, f32x2{v1[2], v1[3]}};
                const f32x2 rr[4] = {f32x2{resv[it][0].x, resv[it][0].y}, f32x2{resv[it][0].z, resv[it][0].w},
                                     f32x2{resv[it][1].x, resv[it][1].y}, f32x2{resv[it][1].z, resv[it][1].w}};
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = (v[q] + b2t[q] + rr[q]) * out_scale2;
                if constexpr (ACCMODE == 1) {
                    v[0] += f32x2{accin[it][0].x, accin[it][0].y}; v[1] += f32x2{accin[it][0].z, accin[it][0].w};
                    v[2] += f32x2{accin[it][1].x, accin[it][1].y}; v[3] += f32x2{accin[it][1].z, accin[it][1].w};
                }
                float* op = rowok ? o32 + (long)t * e.ldo + eco : reinterpret_cast<float*>(trash);
                if (outmask == 0xffffffffu) {          // (wave-uniform) every output row of the wave is inside an utterance: no selects
                    *reinterpret_cast<float4*>(op) = make_float4(v[0][0], v[0][1], v[1][0], v[1][1]);
                    *reinterpret_cast<float4*>(op + 4) = make_float4(v[2][0], v[2][1], v[3][0], v[3][1]);
                } else {
                    *reinterpret_cast<float4*>(op) = valid ? make_float4(v[0][0], v[0][1], v[1][0], v[1][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(op + 4) = valid ? make_float4(v[2][0], v[2][1], v[3][0], v[3][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        } else {
            // ---------------- epilogue in the C layout: rows m0 + lw * 32 + b * 16 + fr
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int t = m0 + lw * 32 + b * 16 + fr;
                const bool rowok = t < t_end;
                const bool valid = (outmask & (frbit << (b * 16))) != 0u;
                float* const orow = rowok ? o32 + (long)t * e.ldo + 4 * fq : reinterpret_cast<float*>(trash);
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    f32x2 v0 = (f32x2{acc[a][b][0], acc[a][b][1]} + b2v[a][0] + f32x2{resv[a][b].x, resv[a][b].y}) * out_scale2;
                    f32x2 v1 = (f32x2{acc[a][b][2], acc[a][b][3]} + b2v[a][1] + f32x2{resv[a][b].z, resv[a][b].w}) * out_scale2;
                    if constexpr (ACCMODE == 1) { v0 += f32x2{accin[a][b].x, accin[a][b].y}; v1 += f32x2{accin[a][b].z, accin[a][b].w}; }
                    float* const op = rowok ? orow + a * 16 : orow;
                    if (outmask == 0xffffffffu) *reinterpret_cast<float4*>(op) = make_float4(v0[0], v0[1], v1[0], v1[1]);          // (wave-uniform)
                    else *reinterpret_cast<float4*>(op) = valid ? make_float4(v0[0], v0[1], v1[0], v1[1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        vmask = vmask_next;
        EV_PT(6)
        EV_PE5_GROUP_BARRIER()          // the group's new slab is complete; xt may be overwritten
        EV_PT(7)
    }
    if (gi == 0) __builtin_amdgcn_s_barrier();          // barrier counts of the two groups match
    EV_PT_DUMP
#undef EV_PE5_GROUP_BARRIER
#undef EV_PE5_CONV
#undef EV_PE5_SSTORE
#undef EV_PE5_GLOAD
#undef EV_PE5_ROW
#undef EV_PE5_VROW
#undef EV_PE5_VLOAD
#undef EV_PE5_VMASK
}

template <int K>
static hipError_t pair_e5_attr() {
    hipError_t e = hipSuccess, r;
    r = hipFuncSetAttribute((const void*)resblock_pair_c32_e5_kernel<K, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PairE5Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)resblock_pair_c32_e5_kernel<K, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, PairE5Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    return e;
}
static hipError_t pair_e5_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = pair_e5_attr<3>(); if (r != hipSuccess) e = r;
    r = pair_e5_attr<7>(); if (r != hipSuccess) e = r;
    r = pair_e5_attr<11>(); if (r != hipSuccess) e = r;
    return e;
}
