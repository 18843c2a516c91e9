// Fused HiFi-GAN ResBlock pair at C = 32 (stage 3) in the "MX" arithmetic of ev_gemm_mx.h:
//
//        xt  = leaky_relu(c1(leaky_relu(x, .1)) + b1, .1)          c1 = Conv1d(32, 32, k, dilation d)
//        out = epilogue(c2(xt) + b2 + x)                            c2 = Conv1d(32, 32, k, dilation 1)       (models/hifigan/models.py:50-57)
//
// with x and out fp32 in HBM (the raw residual stream of the split-precision data flow) and every product evaluated as
// xh.wh (one fp16 MFMA per tap: K = 32 channels) + Q(xh).Q(wl) + Q(xl).Q(wh) (block-scaled fp4 MFMAs, K = 128 = FOUR TAPS x 32 channels
// per instruction: k-block q of a lane is tap 4 g + q, so one E8M0 scale per lane is one scale per (row, tap) = per 32 channels).
// Layer-wise the split-precision mode moves 5 fp32 tensor passes per pair (5.4 GB at B = 32 x 1024 frames) and issues 3 MFMAs per product;
// here x crosses HBM once in and once out (2.2 GB) and a product costs 1 + 2/4 MFMA-equivalents.
//
// One persistent 8-wave block per CU (the structure of resblock_pair_c32_kernel):
//   * both convs' weights stay in LDS: fp16 hi parts (swizzled 64-byte rows), fp4 code planes [plane][tap][co][16 B] of the lo / hi parts and
//     their scale bytes (host layout: emotivoice_amd/mxfp4.py pack_pair_weight_planes; taps padded to a multiple of four with zero codes);
//   * the fp32 slab of the NEXT tile (256 + (k-1)(d+1) rows) is requested into registers at the top of a tile; after conv1 has finished with
//     the current slab (the mid-tile barrier) it is turned into the operand planes in place: leaky-relu, fp16 hi plane, fp4 codes of the hi and
//     of the remainder, scale bytes (mx_quant8: the same quantiser as the conv-GEMM epilogue) -- one slab buffer, not two;
//   * conv1's result gets bias + leaky-relu + sequence-edge zeroing in registers and goes to LDS as the same kind of plane set (the block maximum
//     of a row is spread over the four k-groups of lanes: two cross-lane maxima per row group);
//   * conv2, then the residual (raw fp32 rows re-read from L2) / scale / fp32 accumulate-in epilogue through a 16-row transposing scratch.
#pragma once

template <int K>
struct PairMxGeom {
    static constexpr int C = 32, H2 = (K - 1) / 2, BMO = 256 - 2 * H2, KG = (K + 3) / 4, KP = KG * 4;
    static constexpr int XROWS = 320, TROWS = 272, EPITCH = C * 4 + 16;
    static constexpr int WHB = K * C * 64;             // fp16 weights of one conv
    static constexpr int WQB = KP * C * 16;            // one fp4 code plane of one conv
    static constexpr int WSB = KP * C;                 // its scale bytes
    static constexpr int OFF_WH = 0, OFF_WQ = 2 * WHB, OFF_WS = OFF_WQ + 4 * WQB, OFF_XH = OFF_WS + 4 * WSB;
    static constexpr int OFF_XQ = OFF_XH + XROWS * 64, OFF_XS = OFF_XQ + 2 * XROWS * 16, OFF_TH = OFF_XS + 2 * XROWS;
    static constexpr int OFF_TQ = OFF_TH + TROWS * 64, OFF_TS = OFF_TQ + 2 * TROWS * 16, OFF_ES = OFF_TS + 2 * TROWS;
    static constexpr int TOTAL = OFF_ES + 8 * 16 * EPITCH;
    static_assert(OFF_XH % 16 == 0 && OFF_TH % 16 == 0 && OFF_ES % 16 == 0 && TOTAL <= 160 * 1024, "LDS plan");
};

// ACCMODE: 0 = none, 1 = fp32 accumulate-in (epi.acc32, may alias epi.out32: the running MRF sum)
template <int K, int ACCMODE>
__global__ __launch_bounds__(512, 1) void resblock_pair_c32_mx_kernel(const ResPairParams p) {
    using G = PairMxGeom<K>;
    constexpr int C = G::C, H2 = G::H2, BMO = G::BMO, KG = G::KG, XROWS = G::XROWS, TROWS = G::TROWS, EPITCH = G::EPITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const W1h = smem + G::OFF_WH;
    char* const W2h = W1h + G::WHB;
    char* const Wq = smem + G::OFF_WQ;        // [conv][plane][KP][32][16]
    char* const Wsc = smem + G::OFF_WS;       // [conv][plane][KP][32]
    char* const Xh = smem + G::OFF_XH;
    char* const Xq = smem + G::OFF_XQ;        // [plane][XROWS][16]
    char* const Xsc = smem + G::OFF_XS;       // [plane][XROWS]
    char* const Th = smem + G::OFF_TH;
    char* const Tq = smem + G::OFF_TQ;        // [plane][TROWS][16]
    char* const Tsc = smem + G::OFF_TS;       // [plane][TROWS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* const es = smem + G::OFF_ES + wave * 16 * EPITCH;
    const int fr = lane & 15, fq = lane >> 4;
    const int dil = p.dil, h1 = H2 * dil;
    const int x_pitch = p.ldx * 4;
    const char* xg = reinterpret_cast<const char*>(p.x);
    const int ntiles = (p.M + BMO - 1) / BMO;
    const ConvGemmParams& e = p.epi;
    const int gmin = p.gmax ? p.gmin : 0, gmax = p.gmax ? p.gmax : p.M;

    // ---- weights -> LDS, once per block
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
    f32x2 b1v[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) b1v[a][q] = f32x2{p.b1[a * 16 + 4 * fq + 2 * q], p.b1[a * 16 + 4 * fq + 2 * q + 1]};
    const int er = lane >> 2, eg = lane & 3, eco = eg * 8;          // coalesced side of the epilogue: 4 lanes per row, 16 rows per instruction
    const unsigned frbit = 1u << fr, erbit = 1u << er;
    f32x2 b2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = e.bias ? f32x2{e.bias[eco + 2 * q], e.bias[eco + 2 * q + 1]} : f32x2{0.f, 0.f};
    const f32x2 out_scale2 = f32x2{e.out_scale, e.out_scale};
    const f32x2 slope01 = f32x2{0.1f, 0.1f};
    float* const o32 = e.out32;
    char* const trash = g_store_trash + lane * 64;
    const uint8_t* vptr = e.row_valid ? e.row_valid : g_row_always_valid;
    const int vshift = e.row_valid ? e.valid_shift : 31;
#define EV_PMX_VROW(TILE) ((TILE) * BMO - H2 + wave * 32 + lane)
#define EV_PMX_VLOAD(TILE, DST) { const int g_ = EV_PMX_VROW(TILE); DST = vptr[min(max(g_, gmin), gmax - 1) >> vshift]; }
#define EV_PMX_VMASK(TILE, SRC) __builtin_amdgcn_ballot_w64((SRC) != 0 && EV_PMX_VROW(TILE) >= gmin && EV_PMX_VROW(TILE) < gmax)

    // ---- slab staging: a thread owns three (row, 8-channel quarter) units of the 320-row slab: rows (tid >> 2) + {0, 128, 256}
    float4 xr[3][2];
    const int xq = tid & 3;
    const char* const xgt = xg + xq * 32;
    const int xrow2 = min((tid >> 2) + 256, 255 + 2 * h1 + 2 * H2);      // rows beyond the convs' span re-read the last needed row (a cache hit)
    const int drow[3] = {tid >> 2, (tid >> 2) + 128, min((tid >> 2) + 256, XROWS - 1)};
#define EV_PMX_ROW(G_) min((G_), gmax + 63)
#define EV_PMX_GLOAD(TILE)                                                                                 \
    {                                                                                                      \
        const int g0_ = (TILE) * BMO - H2 - h1 + (tid >> 2);                                               \
        const int g2_ = (TILE) * BMO - H2 - h1 + xrow2;                                                    \
        const char* q0_ = xgt + (long)EV_PMX_ROW(g0_) * x_pitch;                                           \
        const char* q1_ = xgt + (long)EV_PMX_ROW(g0_ + 128) * x_pitch;                                     \
        const char* q2_ = xgt + (long)EV_PMX_ROW(g2_) * x_pitch;                                           \
        xr[0][0] = *reinterpret_cast<const float4*>(q0_); xr[0][1] = *reinterpret_cast<const float4*>(q0_ + 16); \
        xr[1][0] = *reinterpret_cast<const float4*>(q1_); xr[1][1] = *reinterpret_cast<const float4*>(q1_ + 16); \
        xr[2][0] = *reinterpret_cast<const float4*>(q2_); xr[2][1] = *reinterpret_cast<const float4*>(q2_ + 16); \
    }
    // leaky_relu(x, .1) of models.py:51, then the operand planes of the slab (the four threads of a row are one quad)
#define EV_PMX_SSTORE()                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                        \
        f32x2 a_[4] = {lrelu2(f32x2{xr[i][0].x, xr[i][0].y}, slope01), lrelu2(f32x2{xr[i][0].z, xr[i][0].w}, slope01), \
                       lrelu2(f32x2{xr[i][1].x, xr[i][1].y}, slope01), lrelu2(f32x2{xr[i][1].z, xr[i][1].w}, slope01)}; \
        uint4 ho_; unsigned ch_, cl_, bh_, bl_;                                                            \
        mx_quant8(a_, ho_, ch_, cl_, bh_, bl_);                                                            \
        *reinterpret_cast<uint4*>(Xh + swz(drow[i], xq)) = ho_;                                            \
        *reinterpret_cast<unsigned*>(Xq + drow[i] * 16 + xq * 4) = ch_;                                    \
        *reinterpret_cast<unsigned*>(Xq + XROWS * 16 + drow[i] * 16 + xq * 4) = cl_;                       \
        if (xq == 0) { Xsc[drow[i]] = (char)bh_; Xsc[XROWS + drow[i]] = (char)bl_; }                       \
    }

    // one conv of the pair on the wave's 32 rows: fp16 hi x hi tap by tap, then the two fp4 cross terms four taps at a time.
    // XH / XQ / XS: the operand's planes (row pitch 64 / 16 / 1 bytes, XQ / XS plane stride NR rows), WH: the conv's fp16 weights, CONV: 0 / 1
#define EV_PMX_CONV(XH, XQ, XS, NR, WH, CONV, DIL)                                                         \
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
            const int tw = 4 * g + fq;                     /* this lane's tap (weights: zero codes beyond K) */ \
            const int rq = wrow0 + min(tw, K - 1) * (DIL); /* ... and its operand row (a real row for the padded taps) */ \
            _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                             \
                uint4 wq_[2], xq_[2];                                                                      \
                int ws_[2], xs_[2];                                                                        \
                _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                            \
                    wq_[a] = *reinterpret_cast<const uint4*>(Wq + ((CONV) * 2 + pl) * G::WQB + (tw * 32 + a * 16 + fr) * 16); \
                    ws_[a] = *reinterpret_cast<const uint8_t*>(Wsc + ((CONV) * 2 + pl) * G::WSB + tw * 32 + a * 16 + fr); \
                }                                                                                          \
                _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                            \
                    xq_[b] = *reinterpret_cast<const uint4*>((XQ) + pl * (NR) * 16 + (rq + b * 16) * 16);  \
                    xs_[b] = *reinterpret_cast<const uint8_t*>((XS) + pl * (NR) + rq + b * 16);            \
                }                                                                                          \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                              \
                    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                          \
                        mfma_mx_inplace(acc[a][b], wq_[a], xq_[b], ws_[a], xs_[b]);                          \
            }                                                                                              \
        }                                                                                                  \
        mfma_asm_fence(acc);          /* the quantiser / the epilogue read the accumulators next */       \
    }

    int tile = blockIdx.x;                    // grid <= ntiles
    unsigned long long vmask;
    {
        uint8_t vb;
        EV_PMX_GLOAD(tile)
        EV_PMX_VLOAD(tile, vb)
        EV_PMX_SSTORE()
        vmask = EV_PMX_VMASK(tile, vb);
    }
    __syncthreads();
    const int wrow0 = wave * 32 + fr;
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = min(tile + (int)gridDim.x, ntiles - 1);      // clamped: the last prefetch of a block is never used
        const int m0 = tile * BMO;
        const int t_end = min(m0 + BMO, p.M);
        // ---------------- memory requests of this iteration, oldest first: raw residual rows (L2 hits: the slab just came through),
        // the accumulate-in rows, then the next tile's slab and row-valid byte
        float4 resv[2][2], accin[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = min(m0 + wave * 32 + it * 16 + er, t_end - 1);
            const char* rp = xg + (long)t * x_pitch + eg * 32;
            resv[it][0] = *reinterpret_cast<const float4*>(rp);
            resv[it][1] = *reinterpret_cast<const float4*>(rp + 16);
            if constexpr (ACCMODE == 1) {
                const float* ap = e.acc32 + (long)t * e.ldacc + eco;
                accin[it][0] = *reinterpret_cast<const float4*>(ap);
                accin[it][1] = *reinterpret_cast<const float4*>(ap + 4);
            }
        }
        uint8_t vb_next;
        EV_PMX_GLOAD(next)
        EV_PMX_VLOAD(next, vb_next)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][2];
        // ---------------- conv1 (dilation d): 256 rows, global rows m0 - H2 + r1
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        EV_PMX_CONV(Xh, Xq, Xsc, XROWS, W1h, 0, dil)
        // bias + leaky-relu + zero outside the utterance (conv2 must see the reference's zero padding) -> the xt plane set
        const unsigned xtmask = (unsigned)vmask;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r1 = wrow0 + b * 16;
            const bool valid = (xtmask & (frbit << (b * 16))) != 0u;
            f32x2 v[2][2];
            half2v hh[2][2];
            f32x2 hf[2][2], lf[2][2];
            float mh = 0.f, ml = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                v[a][0] = lrelu2(f32x2{acc[a][b][0], acc[a][b][1]} + b1v[a][0], slope01);
                v[a][1] = lrelu2(f32x2{acc[a][b][2], acc[a][b][3]} + b1v[a][1], slope01);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    v[a][q][0] = valid ? v[a][q][0] : 0.f; v[a][q][1] = valid ? v[a][q][1] : 0.f;
                    hh[a][q] = __builtin_convertvector(v[a][q], half2v);
                    hf[a][q] = __builtin_convertvector(hh[a][q], f32x2);
                    lf[a][q] = v[a][q] - hf[a][q];
                    mh = max3_abs_raw(hf[a][q][0], hf[a][q][1], mh);
                    ml = max3_abs_raw(lf[a][q][0], lf[a][q][1], ml);
                }
            }
            // a row's 32 channels sit in the four k-groups of lanes (fr fixed): maxima across lanes l, l ^ 16, l ^ 32, l ^ 48
            mh = max_xor16_raw(max_xor32_raw(mh));
            ml = max_xor16_raw(max_xor32_raw(ml));
            const unsigned bh = mx_scale_byte(mh), bl = mx_scale_byte(ml);
            const float sh = __uint_as_float(bh << 23), sl = __uint_as_float(bl << 23);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                uint2 w;
                w.x = *reinterpret_cast<unsigned*>(&hh[a][0]); w.y = *reinterpret_cast<unsigned*>(&hh[a][1]);
                const int co = a * 16 + 4 * fq;
                *reinterpret_cast<uint2*>(Th + swz(r1, co >> 3) + (co & 7) * 2) = w;
                unsigned ch = 0, cl = 0;
                ch = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ch, hf[a][0][0], hf[a][0][1], sh, 0);
                ch = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ch, hf[a][1][0], hf[a][1][1], sh, 1);
                cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[a][0][0], lf[a][0][1], sl, 0);
                cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[a][1][0], lf[a][1][1], sl, 1);
                *reinterpret_cast<unsigned short*>(Tq + r1 * 16 + (co >> 1)) = (unsigned short)ch;
                *reinterpret_cast<unsigned short*>(Tq + TROWS * 16 + r1 * 16 + (co >> 1)) = (unsigned short)cl;
            }
            if (fq == 0) { Tsc[r1] = (char)bh; Tsc[TROWS + r1] = (char)bl; }
        }
        __syncthreads();          // every wave is done with the slab; xt is complete
        // ---------------- the next tile's slab replaces the current one (conv2 only reads xt)
        EV_PMX_SSTORE()
        const unsigned long long vmask_next = EV_PMX_VMASK(next, vb_next);
        // ---------------- conv2 (dilation 1): rows m0 + r2, reads xt rows r2 + t
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        EV_PMX_CONV(Th, Tq, Tsc, TROWS, W2h, 1, 1)
        // ---------------- epilogue: 16-row passes through the wave's transposing scratch, 32-byte row-contiguous fp32 stores
        const unsigned outmask = (unsigned)(vmask >> H2);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = m0 + wave * 32 + it * 16 + er;
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
            for (int q = 0; q < 4; ++q) v[q] = (v[q] + b2v[q] + rr[q]) * out_scale2;
            if constexpr (ACCMODE == 1) {
                v[0] += f32x2{accin[it][0].x, accin[it][0].y}; v[1] += f32x2{accin[it][0].z, accin[it][0].w};
                v[2] += f32x2{accin[it][1].x, accin[it][1].y}; v[3] += f32x2{accin[it][1].z, accin[it][1].w};
            }
            float* op = rowok ? o32 + (long)t * e.ldo + eco : reinterpret_cast<float*>(trash);
            *reinterpret_cast<float4*>(op) = valid ? make_float4(v[0][0], v[0][1], v[1][0], v[1][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(op + 4) = valid ? make_float4(v[2][0], v[2][1], v[3][0], v[3][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        vmask = vmask_next;
        __syncthreads();          // the new slab is complete; xt may be overwritten
    }
#undef EV_PMX_CONV
#undef EV_PMX_SSTORE
#undef EV_PMX_GLOAD
#undef EV_PMX_ROW
#undef EV_PMX_VROW
#undef EV_PMX_VLOAD
#undef EV_PMX_VMASK
}


// ---- Round 4: the same pair with the block's eight waves split into TWO GROUPS one barrier apart (see conv_c64_mx2_kernel).  Waves 0-3 and waves 4-7
// (SIMD partners) each own a 128-row tile of their own (BMO = 128 - 2 H2 output rows), with private slab and xt plane sets; only the read-only weights are
// shared.  Both groups run the same two barriers per tile, group 1 one barrier late: while one wave of a SIMD is in [requests, conv1, xt -> LDS] its partner
// is in [next slab -> LDS, conv2, epilogue], so the latency chains of one group (global loads, LDS round trips of the quantisers, store issue) sit beside the
// other group's matrix work instead of beside an identical copy of themselves.  Same arithmetic per output element: bit-identical to the lock-step kernel.
template <int K>
struct PairMx2Geom {
    static constexpr int C = 32, H2 = (K - 1) / 2, GR = 128, BMO = GR - 2 * H2, KG = (K + 3) / 4, KP = KG * 4;
    static constexpr int XR = 192, TR = 144, EPITCH = C * 4 + 16;
    static constexpr int WHB = K * C * 64, WQB = KP * C * 16, WSB = KP * C;
    static constexpr int OFF_WH = 0, OFF_WQ = 2 * WHB, OFF_WS = OFF_WQ + 4 * WQB, OFF_G = OFF_WS + 4 * WSB;
    static constexpr int G_XH = 0, G_XQ = XR * 64, G_XS = G_XQ + 2 * XR * 16, G_TH = G_XS + 2 * XR, G_TQ = G_TH + TR * 64, G_TS = G_TQ + 2 * TR * 16;
    static constexpr int GB = G_TS + 2 * TR;            // one group's slab + xt plane sets
    static constexpr int OFF_ES = OFF_G + 2 * GB;
    static constexpr int TOTAL = OFF_ES + 8 * 16 * EPITCH;
    static_assert(OFF_G % 16 == 0 && GB % 16 == 0 && G_TH % 16 == 0 && OFF_ES % 16 == 0 && TOTAL <= 160 * 1024 && GR + K - 1 <= TR, "LDS plan");
};

template <int K, int ACCMODE>
__global__ __launch_bounds__(512, 1) void resblock_pair_c32_mx2_kernel(const ResPairParams p) {
    using G = PairMx2Geom<K>;
    constexpr int C = G::C, H2 = G::H2, BMO = G::BMO, KG = G::KG, XR = G::XR, TR = G::TR, EPITCH = G::EPITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const W1h = smem + G::OFF_WH;
    char* const W2h = W1h + G::WHB;
    char* const Wq = smem + G::OFF_WQ;
    char* const Wsc = smem + G::OFF_WS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave >> 2, lw = wave & 3, ltid = tid & 255;
    char* const gb = smem + G::OFF_G + gi * G::GB;
    char* const Xh = gb + G::G_XH;
    char* const Xq = gb + G::G_XQ;        // [plane][XR][16]
    char* const Xsc = gb + G::G_XS;       // [plane][XR]
    char* const Th = gb + G::G_TH;
    char* const Tq = gb + G::G_TQ;        // [plane][TR][16]
    char* const Tsc = gb + G::G_TS;       // [plane][TR]
    char* const es = smem + G::OFF_ES + wave * 16 * EPITCH;
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
    f32x2 b1v[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) b1v[a][q] = f32x2{p.b1[a * 16 + 4 * fq + 2 * q], p.b1[a * 16 + 4 * fq + 2 * q + 1]};
    const int er = lane >> 2, eg = lane & 3, eco = eg * 8;
    const unsigned frbit = 1u << fr, erbit = 1u << er;
    f32x2 b2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = e.bias ? f32x2{e.bias[eco + 2 * q], e.bias[eco + 2 * q + 1]} : f32x2{0.f, 0.f};
    const f32x2 out_scale2 = f32x2{e.out_scale, e.out_scale};
    const f32x2 slope01 = f32x2{0.1f, 0.1f};
    float* const o32 = e.out32;
    char* const trash = g_store_trash + lane * 64;
    const uint8_t* vptr = e.row_valid ? e.row_valid : g_row_always_valid;
    const int vshift = e.row_valid ? e.valid_shift : 31;
#define EV_PMX_VROW(TILE) ((TILE) * BMO - H2 + lw * 32 + lane)
#define EV_PMX_VLOAD(TILE, DST) { const int g_ = EV_PMX_VROW(TILE); DST = vptr[min(max(g_, gmin), gmax - 1) >> vshift]; }
#define EV_PMX_VMASK(TILE, SRC) __builtin_amdgcn_ballot_w64((SRC) != 0 && EV_PMX_VROW(TILE) >= gmin && EV_PMX_VROW(TILE) < gmax)

    // ---- slab staging of a group: a thread owns three (row, 8-channel quarter) units of the 192-row slab: rows (ltid >> 2) + {0, 64, 128}
    float4 xr[3][2];
    const int xq = ltid & 3;
    const char* const xgt = xg + xq * 32;
    const int xrow2 = min((ltid >> 2) + 128, G::GR - 1 + 2 * h1 + 2 * H2);      // rows beyond the convs' span re-read the last needed row
    const int drow[3] = {ltid >> 2, (ltid >> 2) + 64, (ltid >> 2) + 128};
    // (round 5) the third unit of a thread is slab row 128 + lw * 16 + lane / 4: the convs read rows < 128 + 2 (h1 + H2) only, so for the short spans (k = 3: 4-12 rows,
    // k = 7 at d = 1: 12) whole waves have no third unit to stage -- a wave-uniform skip of its two loads and ~75 VALU of quantiser per tile.  This kernel is bound by the
    // SIMDs' issue ports, not by HBM or latency: instructions not issued are time saved.  Rows nobody reads stay unwritten; the outputs keep their bits.
    const bool r4_paths = (e.reserved0 & 8) != 0;          // in-process A/B (tools/bench_pair_mx.py --dbg 8): round 4's instruction stream
    const bool need3 = r4_paths || lw * 16 < 2 * (h1 + H2);
#define EV_PMX_ROW(G_) min((G_), gmax + 63)
#define EV_PMX_GLOAD(TILE)                                                                                 \
    {                                                                                                      \
        const int g0_ = (TILE) * BMO - H2 - h1 + (ltid >> 2);                                              \
        const int g2_ = (TILE) * BMO - H2 - h1 + xrow2;                                                    \
        const char* q0_ = xgt + (long)EV_PMX_ROW(g0_) * x_pitch;                                           \
        const char* q1_ = xgt + (long)EV_PMX_ROW(g0_ + 64) * x_pitch;                                      \
        const char* q2_ = xgt + (long)EV_PMX_ROW(g2_) * x_pitch;                                           \
        xr[0][0] = *reinterpret_cast<const float4*>(q0_); xr[0][1] = *reinterpret_cast<const float4*>(q0_ + 16); \
        xr[1][0] = *reinterpret_cast<const float4*>(q1_); xr[1][1] = *reinterpret_cast<const float4*>(q1_ + 16); \
        if (need3) { xr[2][0] = *reinterpret_cast<const float4*>(q2_); xr[2][1] = *reinterpret_cast<const float4*>(q2_ + 16); } \
    }
#define EV_PMX_SSTORE()                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) if (i < 2 || need3) {                                    \
        f32x2 a_[4] = {lrelu2(f32x2{xr[i][0].x, xr[i][0].y}, slope01), lrelu2(f32x2{xr[i][0].z, xr[i][0].w}, slope01), \
                       lrelu2(f32x2{xr[i][1].x, xr[i][1].y}, slope01), lrelu2(f32x2{xr[i][1].z, xr[i][1].w}, slope01)}; \
        uint4 ho_; unsigned ch_, cl_, bh_, bl_;                                                            \
        mx_quant8(a_, ho_, ch_, cl_, bh_, bl_);                                                            \
        *reinterpret_cast<uint4*>(Xh + swz(drow[i], xq)) = ho_;                                            \
        *reinterpret_cast<unsigned*>(Xq + drow[i] * 16 + xq * 4) = ch_;                                    \
        *reinterpret_cast<unsigned*>(Xq + XR * 16 + drow[i] * 16 + xq * 4) = cl_;                          \
        if (xq == 0) { Xsc[drow[i]] = (char)bh_; Xsc[XR + drow[i]] = (char)bl_; }                          \
    }
#define EV_PMX_CONV(XH, XQ, XS, NR, WH, CONV, DIL)                                                         \
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
            const int tw = 4 * g + fq;                                                                     \
            const int rq = wrow0 + min(tw, K - 1) * (DIL);                                                 \
            _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                             \
                uint4 wq_[2], xq_[2];                                                                      \
                int ws_[2], xs_[2];                                                                        \
                _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                            \
                    wq_[a] = *reinterpret_cast<const uint4*>(Wq + ((CONV) * 2 + pl) * G::WQB + (tw * 32 + a * 16 + fr) * 16); \
                    ws_[a] = *reinterpret_cast<const uint8_t*>(Wsc + ((CONV) * 2 + pl) * G::WSB + tw * 32 + a * 16 + fr); \
                }                                                                                          \
                _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                            \
                    xq_[b] = *reinterpret_cast<const uint4*>((XQ) + pl * (NR) * 16 + (rq + b * 16) * 16);  \
                    xs_[b] = *reinterpret_cast<const uint8_t*>((XS) + pl * (NR) + rq + b * 16);            \
                }                                                                                          \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                              \
                    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                          \
                        mfma_mx_inplace(acc[a][b], wq_[a], xq_[b], ws_[a], xs_[b]);                          \
            }                                                                                              \
        }                                                                                                  \
        mfma_asm_fence(acc);          /* the quantiser / the epilogue read the accumulators next */       \
    }
    // block barrier of the main loop: LDS traffic retired, vmcnt left alone (see conv_c64_mx2_kernel)
#define EV_PMX_GROUP_BARRIER()                                   \
    {                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        __builtin_amdgcn_sched_barrier(0);                       \
    }

    // tile stream of a group: tiles 2 (b + n grid) + gi; both groups run the same number of iterations (a tile index beyond the last recomputes
    // the last tile and stores nothing)
    const int niter = (ntiles + 2 * (int)gridDim.x - 1) / (2 * (int)gridDim.x);
    int tq = 2 * (int)blockIdx.x + gi;
    int tile = min(tq, ntiles - 1);
    unsigned long long vmask;
    {
        uint8_t vb;
        EV_PMX_GLOAD(tile)
        EV_PMX_VLOAD(tile, vb)
        EV_PMX_SSTORE()
        vmask = EV_PMX_VMASK(tile, vb);
    }
    __syncthreads();
    if (gi == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0 from here on
    const int wrow0 = lw * 32 + fr;
    for (int it_ = 0; it_ < niter; ++it_, tq += 2 * (int)gridDim.x) {
        tile = min(tq, ntiles - 1);
        const int next = min(tq + 2 * (int)gridDim.x, ntiles - 1);
        const int m0 = tile * BMO;
        const int t_end = tq < ntiles ? min(m0 + BMO, p.M) : m0;          // (a repeated tile stores nothing)
        float4 resv[2][2], accin[2][2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int t = max(min(m0 + lw * 32 + it * 16 + er, t_end - 1), 0);
            const char* rp = xg + (long)t * x_pitch + eg * 32;
            resv[it][0] = *reinterpret_cast<const float4*>(rp);
            resv[it][1] = *reinterpret_cast<const float4*>(rp + 16);
            if constexpr (ACCMODE == 1) {
                const float* ap = e.acc32 + (long)t * e.ldacc + eco;
                accin[it][0] = *reinterpret_cast<const float4*>(ap);
                accin[it][1] = *reinterpret_cast<const float4*>(ap + 4);
            }
        }
        uint8_t vb_next;
        EV_PMX_GLOAD(next)
        EV_PMX_VLOAD(next, vb_next)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][2];
        // ---------------- conv1 (dilation d): the group's 128 rows, global rows m0 - H2 + r1
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        EV_PMX_CONV(Xh, Xq, Xsc, XR, W1h, 0, dil)
        const unsigned xtmask = (unsigned)vmask;
        const bool xt_masked = r4_paths || xtmask != 0xffffffffu;          // (round 5) the 16 selects per tile that zero xt outside the utterances run only where there is such a row
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r1 = wrow0 + b * 16;
            const bool valid = (xtmask & (frbit << (b * 16))) != 0u;
            f32x2 v[2][2];
            half2v hh[2][2];
            f32x2 hf[2][2], lf[2][2];
            float mh = 0.f, ml = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                v[a][0] = lrelu2(f32x2{acc[a][b][0], acc[a][b][1]} + b1v[a][0], slope01);
                v[a][1] = lrelu2(f32x2{acc[a][b][2], acc[a][b][3]} + b1v[a][1], slope01);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (xt_masked) { v[a][q][0] = valid ? v[a][q][0] : 0.f; v[a][q][1] = valid ? v[a][q][1] : 0.f; }      // (wave-uniform: only a wave with a row outside the utterances)
                    hh[a][q] = __builtin_convertvector(v[a][q], half2v);
                    hf[a][q] = __builtin_convertvector(hh[a][q], f32x2);
                    lf[a][q] = v[a][q] - hf[a][q];
                    mh = max3_abs_raw(hf[a][q][0], hf[a][q][1], mh);
                    ml = max3_abs_raw(lf[a][q][0], lf[a][q][1], ml);
                }
            }
            mh = max_xor16_raw(max_xor32_raw(mh));
            ml = max_xor16_raw(max_xor32_raw(ml));
            const unsigned bh = mx_scale_byte(mh), bl = mx_scale_byte(ml);
            const float sh = __uint_as_float(bh << 23), sl = __uint_as_float(bl << 23);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                uint2 w;
                w.x = *reinterpret_cast<unsigned*>(&hh[a][0]); w.y = *reinterpret_cast<unsigned*>(&hh[a][1]);
                const int co = a * 16 + 4 * fq;
                *reinterpret_cast<uint2*>(Th + swz(r1, co >> 3) + (co & 7) * 2) = w;
                unsigned ch = 0, cl = 0;
                ch = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ch, hf[a][0][0], hf[a][0][1], sh, 0);
                ch = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(ch, hf[a][1][0], hf[a][1][1], sh, 1);
                cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[a][0][0], lf[a][0][1], sl, 0);
                cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[a][1][0], lf[a][1][1], sl, 1);
                *reinterpret_cast<unsigned short*>(Tq + r1 * 16 + (co >> 1)) = (unsigned short)ch;
                *reinterpret_cast<unsigned short*>(Tq + TR * 16 + r1 * 16 + (co >> 1)) = (unsigned short)cl;
            }
            if (fq == 0) { Tsc[r1] = (char)bh; Tsc[TR + r1] = (char)bl; }
        }
        EV_PMX_GROUP_BARRIER()          // every wave of the group is done with the slab; xt is complete
        EV_PMX_SSTORE()                 // the next tile's slab replaces the current one (conv2 only reads xt)
        const unsigned long long vmask_next = EV_PMX_VMASK(next, vb_next);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        EV_PMX_CONV(Th, Tq, Tsc, TR, W2h, 1, 1)
        const unsigned outmask = (unsigned)(vmask >> H2);
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
            for (int q = 0; q < 4; ++q) v[q] = (v[q] + b2v[q] + rr[q]) * out_scale2;
            if constexpr (ACCMODE == 1) {
                v[0] += f32x2{accin[it][0].x, accin[it][0].y}; v[1] += f32x2{accin[it][0].z, accin[it][0].w};
                v[2] += f32x2{accin[it][1].x, accin[it][1].y}; v[3] += f32x2{accin[it][1].z, accin[it][1].w};
            }
            float* op = rowok ? o32 + (long)t * e.ldo + eco : reinterpret_cast<float*>(trash);
            if (outmask == 0xffffffffu && !r4_paths) {          // (round 5, wave-uniform) every output row of the wave is inside an utterance: no selects
                *reinterpret_cast<float4*>(op) = make_float4(v[0][0], v[0][1], v[1][0], v[1][1]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[2][0], v[2][1], v[3][0], v[3][1]);
            } else {
                *reinterpret_cast<float4*>(op) = valid ? make_float4(v[0][0], v[0][1], v[1][0], v[1][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(op + 4) = valid ? make_float4(v[2][0], v[2][1], v[3][0], v[3][1]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        vmask = vmask_next;
        EV_PMX_GROUP_BARRIER()          // the group's new slab is complete; xt may be overwritten
    }
    if (gi == 0) __builtin_amdgcn_s_barrier();          // barrier counts of the two groups match
#undef EV_PMX_GROUP_BARRIER
#undef EV_PMX_CONV
#undef EV_PMX_SSTORE
#undef EV_PMX_GLOAD
#undef EV_PMX_ROW
#undef EV_PMX_VROW
#undef EV_PMX_VLOAD
#undef EV_PMX_VMASK
}

template <int K>
static hipError_t pair_mx_attr() {
    hipError_t e = hipSuccess, r;
    r = hipFuncSetAttribute((const void*)resblock_pair_c32_mx_kernel<K, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PairMxGeom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)resblock_pair_c32_mx_kernel<K, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, PairMxGeom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)resblock_pair_c32_mx2_kernel<K, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PairMx2Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)resblock_pair_c32_mx2_kernel<K, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, PairMx2Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    return e;
}
static hipError_t pair_mx_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = pair_mx_attr<3>(); if (r != hipSuccess) e = r;
    r = pair_mx_attr<7>(); if (r != hipSuccess) e = r;
    r = pair_mx_attr<11>(); if (r != hipSuccess) e = r;
    return e;
}
