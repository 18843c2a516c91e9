// conv_gemm_mx_kernel's pipeline at C = 64 (HiFi-GAN stage 2, k = 7 / 11): the MX arithmetic of ev_gemm_mx.h -- xh.wh as fp16 MFMAs, Q(xh).Q(wl) + Q(xl).Q(wh) as
// block-scaled fp4 MFMAs -- on 256-row x 64-channel tiles, one block per tile, two blocks per CU, weights STREAMED through the four-slot LDS ring instead of
// stationary in a persistent block.
//
// Why a second kernel for these layers.  conv_c64_mx_kernel (ev_conv64_mx.h) keeps the weights of 32 output channels in LDS for the whole launch (k = 11: 70 KB),
// which leaves room for ONE 8-wave block per CU, 32 x 32 outputs per wave (one 16-byte LDS fragment read per MFMA), the slab read once per channel half, and a
// lock-step item (stage, MFMAs, epilogue) that nothing else on the CU overlaps: 0.76-0.88 ms per k = 11 launch at MFMA-busy 0.30, against ~0.2 ms on either roof.
// Here a wave owns 32 rows x all 64 channels (0.75 fragment reads per MFMA), the slab is read once, the epilogue of one resident block runs beside the K loop of
// the other, and a retired block's stores drain behind a fresh block's opening loads (why the dispatcher's back-filling beats a
// persistent loop).  The price: every tile re-reads the conv's weights from L2 (k = 11: 132 KB per 256 x 64 tile, 0.8 of the tile's own operand bytes).
//
// Geometry (everything else -- LDS-DMA staging, swizzle, phase groups, two barriers per step, epilogues -- is conv_gemm_mx_kernel's):
//   * K = 64 is two fp16 chunks of 32 channels; ONE block-scaled MFMA covers TWO TAPS x 64 channels (k-block q of a lane = tap 2 g + (q >> 1), channel half q & 1),
//     as in conv_c64_mx_kernel, so the host planes are the same (mxfp4.pack_c64_weight_planes: taps padded to an even count with zero codes).
//   * a step is 16 MFMAs per wave: fp16 passes: one tap PAIR of one chunk (2 taps x 4 channel tiles x 2 row tiles; the phantom tap of an odd k is skipped);
//     fp4 passes: two tap pairs (2 x 4 x 2).  A step's weight tile is 8 KB in both: 128 LDS rows of 64 B = (tap of the pair | pair of the step) x 64 output channels.
//   * four slab loads per tile: fp16 chunk 0, fp16 chunk 1 (384 rows of 64 B), then the fp4 codes of the hi parts and of the remainders (rows of 32 B: 12 KB of a
//     24-KB buffer) with their activation / weight scale runs in the third piece of waves 4-7, as in the parent kernel.
//   * the whole step sequence of a tile is unrolled at compile time (k = 7: 12 steps, k = 11: 18): request targets, ring slots and every vmcnt immediate are
//     constants derived from the issue order below (mx64_wait_imm).
// Same accumulation order for every output element whatever M is: the kernel is chosen by the layer's shape only.
#pragma once

template <int TAPS>
struct Mx64Sched {
    static constexpr int KG = (TAPS + 1) / 2, KH = (KG + 1) / 2, KP = KG * 2, NS = 2 * KG + 2 * KH;
    static constexpr int chunk_of(int s) { return s < KG ? 0 : (s < 2 * KG ? 1 : (s < 2 * KG + KH ? 2 : 3)); }
    static constexpr int first_of(int c) { return c == 0 ? 0 : (c == 1 ? KG : (c == 2 ? 2 * KG : 2 * KG + KH)); }
    static constexpr bool starts_chunk(int s) { return s == first_of(chunk_of(s)); }
    // requests a wave issues inside step t, in order: [3 slab pieces of the NEXT chunk, if t opens a chunk that has a successor], [the weight tile of step t + 3]
    static constexpr int x_issued(int t) { return (starts_chunk(t) && chunk_of(t) < 3) ? 3 : 0; }
    static constexpr int w_issued(int t) { return t + 3 < NS ? 1 : 0; }
    static constexpr int issued_before(int t) {          // prologue: 3 slab pieces of chunk 0, weight tiles of steps 0-2
        int n = 6;
        for (int j = 0; j < t; ++j) n += x_issued(j) + w_issued(j);
        return n;
    }
    static constexpr int idx_w(int m) { return m < 3 ? 3 + m : issued_before(m - 3) + x_issued(m - 3); }
    static constexpr int idx_xlast(int c) { return c == 0 ? 2 : issued_before(first_of(c - 1)) + 2; }
    // vmcnt immediate of step s's wait (in front of its first barrier): everything step s + 1 reads must have landed -- its weight tile, and its slab if it opens
    // a chunk -- while whatever was issued after the newest of those may stay in flight.  -1: nothing to wait for (the last step).
    static constexpr int wait_imm(int s) {
        if (s + 1 >= NS) return -1;
        int need = idx_w(s + 1);
        if (starts_chunk(s + 1)) { const int x = idx_xlast(chunk_of(s + 1)); need = x > need ? x : need; }
        return issued_before(s) - 1 - need;
    }
};
static_assert(Mx64Sched<11>::NS == 18 && Mx64Sched<7>::NS == 12 && Mx64Sched<11>::wait_imm(0) == 1 && Mx64Sched<11>::wait_imm(1) == 4 && Mx64Sched<11>::wait_imm(2) == 1, "schedule");

template <int N> __device__ __forceinline__ void mx64_wait() {
    if constexpr (N >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int TAPS, int EPI>
__global__ __launch_bounds__(512, 4) void conv_gemm_mx64_kernel(const ConvGemmParams p) {
    using SC = Mx64Sched<TAPS>;
    constexpr int BM = PH_BM, XBUF = PH_XBUF, WBUF = PH_WBUF, MT = 2, NT = 4, KG = SC::KG, KP = SC::KP, NS = SC::NS;
    static_assert(PH_SLABR >= BM + MAX_SPAN + 64 && KP * 128 <= 2048 && (TAPS == 7 || TAPS == 11), "slab rows / weight-scale run / tap counts");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Xs = smem;                 // [2][24 KB]: fp16 chunk = 384 rows x 64 B; fp4 chunk = rows x 32 B, scales at MX_XS_OFF / MX_WS_OFF
    char* const Ws = smem + 2 * XBUF;      // [4][8 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 2;              // phase group: waves w and w + 4 share a SIMD
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    int bid = blockIdx.x;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = bid * BM;
    const int dil = p.dil;
    const int row0 = m0 - p.center * dil;                  // first slab row
    const int row0a = row0 & ~3, soff = row0 - row0a;
    const int last_row = BM + (TAPS - 1) * dil - 1;        // last slab row a fragment reads (the phantom tap of an odd k reads nothing)

    // ---- request geometry (per lane, fixed for the tile)
    const int prow = lane >> 2;
    const unsigned pp16 = (unsigned)(((lane & 3) ^ ((lane >> 3) & 3)) << 4);          // 16-byte part of a 64-byte LDS row after the swizzle
    // weight tiles: LDS row of the lane = wave * 16 + prow = (sel, co): sel = wave >> 2 (tap of the pair / pair of the step), co = (wave & 3) * 16 + prow
    const int wco = (wave & 3) * 16 + prow;
    const unsigned w16_lane = (unsigned)wco * (unsigned)(TAPS * 128) + pp16;                                            // fp16 [64][TAPS][64]
    const unsigned wq = (unsigned)((lane & 3) ^ ((lane >> 3) & 3));                                                     // k-block the lane's slot holds
    const unsigned w4_lane = (unsigned)(((wco >> 5) * KP + (int)(wq >> 1)) * 1024 + (wco & 31) * 32 + (int)(wq & 1) * 16);   // codes [half][KP][32 co][32 B]
    // slab pieces: fp16 rows of 128 B in the hi plane (64-byte chunk c at + 64 c); fp4 rows of 32 B
    unsigned x16_lane[3], x4_lane[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) x16_lane[i] = (unsigned)min((wave + 8 * i) * 16 + prow, last_row) * 128u + pp16;
#pragma unroll
    for (int i = 0; i < 2; ++i) x4_lane[i] = (unsigned)min((wave + 8 * i) * 32 + (lane >> 1), last_row) * 32u + (unsigned)(lane & 1) * 16u;
    // third piece of an fp4 chunk: waves 4, 5 the activation scale run (4 B per slab row from row0a), waves 6, 7 the plane's weight scales ([half][KP][32][2]);
    // waves 0-3 repeat their second code piece
    const int xs_run = ((BM + (TAPS - 1) * dil + soff) * 4 + 15) & ~15;
    unsigned sc_lane;
    if (wave == 4) sc_lane = min(lane * 16, xs_run - 16);
    else if (wave == 5) sc_lane = min(1024 + lane * 16, xs_run - 16);
    else if (wave == 6) sc_lane = min(lane * 16, KP * 128 - 16);
    else sc_lane = min(1024 + lane * 16, KP * 128 - 16);

    const char* const wmx = reinterpret_cast<const char*>(p.W_mx);
    constexpr int WQB = KP * 32 * 32 * 2, WSB = KP * 32 * 2 * 2;          // one code plane, one scale plane (both channel halves)
    const char* const xb16 = uniform_ptr(reinterpret_cast<const char*>(p.A) + (long)row0 * 128L);
    const char* const xb4[2] = {uniform_ptr(reinterpret_cast<const char*>(p.mx_x4[0]) + (long)row0 * 32L), uniform_ptr(reinterpret_cast<const char*>(p.mx_x4[1]) + (long)row0 * 32L)};
    const char* const xsb[2] = {uniform_ptr(reinterpret_cast<const char*>(p.mx_xs[0]) + (long)row0a * 4L), uniform_ptr(reinterpret_cast<const char*>(p.mx_xs[1]) + (long)row0a * 4L)};
    const char* const wb16 = uniform_ptr(reinterpret_cast<const char*>(p.W));
    const unsigned xdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024), wdst = xdst + 2 * XBUF;

    // slab of chunk C (0 / 1: fp16 channels 32 C ..; 2: codes of the hi parts, their scales, Q(wl)'s scales; 3: codes of the remainders, their scales, Q(wh)'s scales)
    auto issue_x = [&](auto cc) {
        constexpr int C_ = decltype(cc)::value;
        constexpr unsigned dst = (unsigned)((C_ & 1) * XBUF);
        if constexpr (C_ < 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16(xb16 + C_ * 64, x16_lane[i], xdst + dst + i * 8192);
        } else {
            constexpr int P_ = C_ - 2;                       // activation plane: 0 = hi codes, 1 = remainder codes; weight plane P_: 0 = Q(wl), 1 = Q(wh)
            glds16(xb4[P_], x4_lane[0], xdst + dst);
            glds16(xb4[P_], x4_lane[1], xdst + dst + 8192);
            const bool sc = wave >= 4;
            const char* const sbase = wave < 6 ? xsb[P_] : wmx + 2 * WQB + P_ * WSB;
            glds16(sc ? sbase : xb4[P_], sc ? sc_lane : x4_lane[1], xdst + dst + 2 * 8192);
        }
    };
    // weight tile of step S -> ring slot S & 3
    auto issue_w = [&](auto ss) {
        constexpr int S_ = decltype(ss)::value;
        constexpr int C_ = SC::chunk_of(S_), U_ = S_ - SC::first_of(C_);
        const int sel = wave >> 2;
        if constexpr (C_ < 2) {
            const int t = min(2 * U_ + sel, TAPS - 1);       // (the phantom tap of an odd k: its rows are loaded from the last real tap and never multiplied)
            glds16(wb16 + C_ * 64 + (unsigned)t * 128u, w16_lane, wdst + (S_ & 3) * WBUF);
        } else {
            const int g = min(2 * U_ + sel, KG - 1);
            glds16(wmx + (C_ - 2) * WQB + (unsigned)(2 * g) * 1024u, w4_lane, wdst + (S_ & 3) * WBUF);
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;

    bool tile_live = true;
    if (p.row_valid) {
        const uint8_t* vp = p.row_valid;
        const int r4 = m0 + lane * 4, vs = p.valid_shift;
        const unsigned any = vp[r4 >> vs] | vp[(r4 + 1) >> vs] | vp[(r4 + 2) >> vs] | vp[(r4 + 3) >> vs];
        tile_live = __builtin_amdgcn_ballot_w64(any != 0) != 0ull;
    }
    if (tile_live) {
        issue_x(std::integral_constant<int, 0>{});
        issue_w(std::integral_constant<int, 0>{});
        issue_w(std::integral_constant<int, 1>{});
        issue_w(std::integral_constant<int, 2>{});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wc == 1) __builtin_amdgcn_s_barrier();

        // fragment offsets.  fp16: slab row r at r * 64 with the 16-byte part swizzled by (r >> 1) & 3; fp4: slab row r at r * 32 + (q & 1) * 16, and the row a lane reads
        // is shifted by (q >> 1) taps (its k-block belongs to the second tap of the pair); weight tile row (sel, co) at swz(sel * 64 + co, q)
        const int xrow = wave * 32 + fr;                                    // + b * 16 + tap * dil
        const int x4row = xrow + (fq >> 1) * dil;                           // + b * 16 + 2 g * dil
        const int xs_base = MX_XS_OFF + soff * 4 + (fq & 1);                // + row * 4
        // weight scale of (co, tap 2 g + (q >> 1), half q & 1): [half co >> 5][KP][co & 31][2]
        int ws_lane[NT];
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int co = a * 16 + fr;
            ws_lane[a] = MX_WS_OFF + (((co >> 5) * KP + (fq >> 1)) * 32 + (co & 31)) * 2 + (fq & 1);          // + 2 g * 64
        }

#define EV_MX64_STEP(S)                                                                                                              \
    if constexpr ((S) < NS) {                                                                                                        \
        constexpr int C_ = SC::chunk_of(S), U_ = (S) - SC::first_of(C_);                                                             \
        const char* const Xb = Xs + (C_ & 1) * XBUF;                                                                                 \
        const char* const Wb = Ws + ((S) & 3) * WBUF;                                                                                \
        uint4 xf[2][MT], wf[2][NT];                                                                                                  \
        int xsc[2][MT], wsc[2][NT];                                                                                                  \
        constexpr int NSEL = (C_ < 2) ? ((2 * U_ + 1 < TAPS) ? 2 : 1) : ((2 * U_ + 1 < KG) ? 2 : 1);                                 \
        int dil_ = dil;                                                                                                              \
        asm volatile("" : "+s"(dil_));                                                                                               \
        _Pragma("unroll") for (int sel = 0; sel < NSEL; ++sel) {                                                                     \
            _Pragma("unroll") for (int a = 0; a < NT; ++a) wf[sel][a] = *reinterpret_cast<const uint4*>(Wb + swz(sel * 64 + a * 16 + fr, fq)); \
            if constexpr (C_ < 2) {                                                                                                  \
                const int r_ = xrow + (2 * U_ + sel) * dil_;                                                                         \
                _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                                     \
                    const int rb_ = r_ + b * 16;                                                                                     \
                    xf[sel][b] = *reinterpret_cast<const uint4*>(Xb + rb_ * 64 + ((fq ^ ((rb_ >> 1) & 3)) << 4));                    \
                }                                                                                                                    \
            } else {                                                                                                                 \
                const int g_ = 2 * U_ + sel;                                                                                         \
                const int r_ = x4row + 2 * g_ * dil_;                                                                                \
                _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                                     \
                    xf[sel][b] = *reinterpret_cast<const uint4*>(Xb + (r_ + b * 16) * 32 + ((fq & 1) << 4));                         \
                    xsc[sel][b] = *reinterpret_cast<const uint8_t*>(Xb + xs_base + (r_ + b * 16) * 4);                               \
                }                                                                                                                    \
                _Pragma("unroll") for (int a = 0; a < NT; ++a) wsc[sel][a] = *reinterpret_cast<const uint8_t*>(Xb + ws_lane[a] + g_ * 128); \
            }                                                                                                                        \
        }                                                                                                                            \
        mx64_wait<SC::wait_imm(S)>();                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                                               \
        _Pragma("unroll") for (int sel = 0; sel < NSEL; ++sel)                                                                       \
            _Pragma("unroll") for (int a = 0; a < NT; ++a)                                                                           \
                _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                                     \
                    if constexpr (C_ < 2) mfma_inplace(acc[a][b], *reinterpret_cast<half8*>(&wf[sel][a]), *reinterpret_cast<half8*>(&xf[sel][b])); \
                    else mfma_mx_inplace(acc[a][b], wf[sel][a], xf[sel][b], wsc[sel][a], xsc[sel][b]);                               \
                    const int idx = (sel * NT + a) * MT + b;                                                                         \
                    if constexpr (SC::x_issued(S) != 0) { if (idx == 2) issue_x(std::integral_constant<int, (C_ < 3 ? C_ + 1 : 3)>{}); } \
                    if constexpr (SC::w_issued(S) != 0) { if (idx == 5) issue_w(std::integral_constant<int, ((S) + 3 < NS ? (S) + 3 : 0)>{}); } \
                }                                                                                                                    \
        __builtin_amdgcn_s_setprio(0);                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
        __builtin_amdgcn_s_barrier();                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                                           \
    }
        EV_MX64_STEP(0) EV_MX64_STEP(1) EV_MX64_STEP(2) EV_MX64_STEP(3) EV_MX64_STEP(4) EV_MX64_STEP(5)
        EV_MX64_STEP(6) EV_MX64_STEP(7) EV_MX64_STEP(8) EV_MX64_STEP(9) EV_MX64_STEP(10) EV_MX64_STEP(11)
        EV_MX64_STEP(12) EV_MX64_STEP(13) EV_MX64_STEP(14) EV_MX64_STEP(15) EV_MX64_STEP(16) EV_MX64_STEP(17)
#undef EV_MX64_STEP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (wc == 0) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        mfma_asm_fence(acc);
    }       // tile_live
    if constexpr (EPI == EPI_GENERIC) gemm_epilogue_lds<MT, NT>(p, acc, smem + wave * epi_wave_bytes<64>(), m0 + wave * 32, 0);
    else {
        EV_TRACE_EPI_DUMMY
        gemm_epilogue_fast<MT, NT, EPI, 16>(p, acc, smem + wave * 4096, m0 + wave * 32, 0 EV_TRACE_EPI_ARGS);
    }
}

template <int TAPS, int EPI>
static void launch_mx64_epi(const ConvGemmParams& p, hipStream_t s) {
    hipLaunchKernelGGL((conv_gemm_mx64_kernel<TAPS, EPI>), dim3(p.M / PH_BM), dim3(512), PH_LDS, s, p);
}
template <int TAPS>
static void launch_mx64_taps(const ConvGemmParams& p, int e, hipStream_t s) {
    switch (e) {
#define EV_MX64_CASE(E) case (E): launch_mx64_epi<TAPS, (E)>(p, s); break;
        EV_MX_VARIANTS(EV_MX64_CASE)
#undef EV_MX64_CASE
        default: break;
    }
}
template <int TAPS>
static hipError_t mx64_attr_taps() {
    hipError_t e = hipSuccess, r;
#define EV_MX64_ATTR(E) r = hipFuncSetAttribute((const void*)conv_gemm_mx64_kernel<TAPS, (E)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    EV_MX_VARIANTS(EV_MX64_ATTR)
#undef EV_MX64_ATTR
    return e;
}
static hipError_t mx64_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = mx64_attr_taps<7>(); if (r != hipSuccess) e = r;
    r = mx64_attr_taps<11>(); if (r != hipSuccess) e = r;
    return e;
}
// DT_MX call with N = K = 64, k = 7 / 11, plane sets in (the generator's stage-2 ResBlock convs): the streamed kernel.  A function of the layer's shape and of the
// epilogue form only; reserved0 bit 3 = in-process A/B switch of the op tests / tools (the persistent conv_c64_mx_kernel).
// Measured at the stage-2 size of configs[1] (4.2 M rows).  Isolated (tools/bench_c64.py --ab): k = 11 conv1 727 against 772 us, conv2 (residual from planes) 720
// against 764; k = 7 (12 steps per tile) 594-606 against 593-604.  In the forward (bench.py per-launch table): k = 11 conv1 / conv2 / conv2 + fp32 MRF sum
// 0.68-0.70 / 0.73-0.77 / 0.85 ms (persistent: 0.75-0.80 / 0.85-0.88 / 1.06), k = 7 0.53 / 0.62 / 0.77 (0.50-0.55 / 0.63-0.65 / 0.84): the family 9.05 -> 8.37 ms.
// What neither kernel escapes is a tile's fixed cost -- opening round trip, plane-set epilogue: ~10 of the 18-22 us a 256 x 64 tile lives.
static bool mx64_eligible(const ConvGemmParams& p) {
    return p.W_mx && p.N == 64 && p.K == 64 && p.lda == 64 && (p.taps == 11 || p.taps == 7) && p.M % PH_BM == 0 && (p.taps - 1) * p.dil <= MAX_SPAN &&
           p.center * 2 == p.taps - 1 && p.mx_x4[0] && p.mx_x4[1] && p.mx_xs[0] && p.mx_xs[1] && !p.pro_lrelu && !(p.reserved0 & 8) && mx_epi_variant(p) >= 0;
}
static void launch_mx64(const ConvGemmParams& p, hipStream_t s) {
    if (p.taps == 7) launch_mx64_taps<7>(p, mx_epi_variant(p), s); else launch_mx64_taps<11>(p, mx_epi_variant(p), s);
}
