// Conv1d at C = 64 (HiFi-GAN stage 2: ResBlock convs, k = 3 / 7 / 11, any dilation; the stage-3 up-conv as its 3-tap polyphase form) in the "MX"
// arithmetic of ev_gemm_mx.h, activations as MX plane sets (+ fp32 where a residual stream needs them):
//
//        out = epilogue( sum_{tap, k} a[m + (tap - center) dil, k] * W[n][tap][k] + bias[n] ),     a = the input plane set's tensor
//
// every product as xh.wh (fp16 MFMA: K = 32 channels, two per tap) + Q(xh).Q(wl) + Q(xl).Q(wh) (block-scaled fp4 MFMAs, K = 128 = TWO TAPS x 64
// channels per instruction: k-block q of a lane is tap 2 g + (q >> 1), channel half q & 1; one E8M0 scale per lane = per 32 channels).
// The split-precision kernel runs these layers at 0.14-0.29 PF/s algorithmic (three MFMAs per product, the fp32 slab split in registers every K-chunk);
// the conv-GEMM MX kernel needs K % 128 == 0.  Here, as in the fused C = 32 pair kernel (ev_pair_mx.h):
//   * one persistent 8-wave block per CU owns 32 of the 64 output channels: the conv's weights for them stay in LDS for the whole launch
//     (fp16 hi parts <= 44 KB, fp4 planes + scales <= 26 KB; host layout: mxfp4.pack_c64_weight_planes).  The two channel halves of a row
//     tile are separate work items scheduled eight blocks apart, i.e. on the same XCD: the second read of the slab is an L2 hit;
//   * operands travel as PLANE SETS written by the producing layer's epilogue (the same contract as conv_gemm_mx_kernel's: fp16 hi plane
//     [rows][64], fp4 code planes of the hi / remainder parts [rows][32 B], E8M0 scale planes [rows][4 B] with two bytes used; the consumer's
//     leaky-relu already applied), so an element is quantised ONCE, in the transposed epilogue where a quad of lanes holds exactly one 32-channel
//     block -- the first version took fp32 rows and quantised every slab in both channel-half work items: ~480 VALU instructions per thread and
//     item against 40-136 MFMAs, 0.7 ms per conv whatever k.  The slab of the next work item (256 + (k - 1) d rows: 62 KB of planes) is requested
//     into registers at the top of an item and copied to LDS (hi plane as two swizzled 64-byte K-chunks) after the current item's MFMAs, in place;
//   * epilogue through a 16-row transposing scratch: bias, leaky-relu (conv1) or fp32 residual / scale / fp32 accumulate-in (conv2), row mask;
//     outputs: fp32 rows (32-byte row-contiguous stores) and / or the plane set of lrelu(result, mxo_slope) for the next conv.
#pragma once

template <int K>
struct Conv64MxGeom {
    static constexpr int C = 64, NB = 32, KG = (K + 1) / 2, KP = KG * 2;
    static constexpr int XROWS = 320, EPITCH = NB * 4 + 16;
    static constexpr int XHC = XROWS * 64 + 64;        // one K-chunk of the slab's hi plane (+ 64 B: the two chunks a staging store writes land in different banks)
    static constexpr int WHB = K * NB * 64;            // one 32-channel K-chunk of the fp16 weights (two chunks)
    static constexpr int WQB = KP * NB * 32;           // one fp4 code plane
    static constexpr int WSB = KP * NB * 2;            // its scale bytes
    static constexpr int OFF_WH = 0, OFF_WQ = 2 * WHB, OFF_WS = OFF_WQ + 2 * WQB, OFF_XH = OFF_WS + 2 * WSB;
    static constexpr int OFF_XQ = OFF_XH + 2 * XHC, OFF_XS = OFF_XQ + 2 * XROWS * 32, OFF_ES = OFF_XS + 2 * XROWS * 4;
    static constexpr int TOTAL = OFF_ES + 8 * 16 * EPITCH;
    static_assert(OFF_XH % 16 == 0 && OFF_XQ % 16 == 0 && OFF_ES % 16 == 0 && TOTAL <= 160 * 1024, "LDS plan");
};

// MODE: 0 = plain (bias + optional leaky-relu), 1 = + residual and out_scale, 2 = + fp32 accumulate-in (may alias out32)
// RPL (MODE >= 1): the residual is the plane set conv1 of the pair read (ConvGemmParams::res_x4 ...: fp16 hi plane + fp4 remainder codes + scale
// bytes, 2.5 bytes per element) instead of an fp32 tensor
template <int K, int MODE, bool RPL = false>
__global__ __launch_bounds__(512, 1) void conv_c64_mx_kernel(const ConvGemmParams p) {
    using G = Conv64MxGeom<K>;
    constexpr int NB = G::NB, KG = G::KG, XROWS = G::XROWS, EPITCH = G::EPITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Wh = smem + G::OFF_WH;        // [chunk][tap * 32 + co][64 B] swizzled
    char* const Wq = smem + G::OFF_WQ;        // [plane][tap * 32 + co][32 B]
    char* const Wsc = smem + G::OFF_WS;       // [plane][tap * 32 + co][2]
    char* const Xh = smem + G::OFF_XH;        // [chunk][row][64 B] swizzled
    char* const Xq = smem + G::OFF_XQ;        // [plane][row][32 B]
    // (code planes, 32-byte rows: a ds_read_b128 is served in lane groups {0-3, 12-15, 20-27} ... = fragment rows {0-3, 12-15} of channel half 0 with
    // rows {4-11} of half 1 (and vice versa): rows r and r + 8 share a 32-byte bank range but always ask for different halves, so the plain layout
    // is conflict-free.  The first version swapped the halves of every second group of 8 rows -- right for 16 consecutive lanes, wrong for the real
    // groups: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.26 on this kernel, LDS busy half of the time)
    char* const Xsc = smem + G::OFF_XS;       // [plane][row][4] (two bytes used: the row's two 32-channel blocks)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    char* const es = smem + G::OFF_ES + wave * 16 * EPITCH;
    const int fr = lane & 15, fq = lane >> 4;
    const int dil = p.dil, hlo = p.center * dil;
    const int ntiles = p.M / 256, nitems = ((ntiles + 7) >> 3) << 4;       // items come in groups of 16 = 8 row tiles x 2 channel halves
    // item i -> (row tile, channel half): the two halves of a tile are items i and i + 8 (persistent blocks b, b + 8: one XCD, one L2);
    // items whose tile does not exist (the last group of a tile count that is not a multiple of 8) recompute the last tile and store nothing
#define EV_C64_TILE(I) min((((I) >> 4) << 3) + ((I) & 7), ntiles - 1)
#define EV_C64_OK(I) (((((I) >> 4) << 3) + ((I) & 7)) < ntiles)
#define EV_C64_HALF(I) (((I) >> 3) & 1)
    int item = blockIdx.x;
    if (item >= nitems) return;
    const int n0 = EV_C64_HALF(item) * NB;          // fixed per block: gridDim.x is a multiple of 16 (or covers every item)

    // ---- this block's half of the weights -> LDS
    {
        const char* w16 = reinterpret_cast<const char*>(p.W);
        for (int c = tid; c < 2 * K * NB * 4; c += 512) {
            const int part = c & 3, row = (c >> 2) % (K * NB), ch = c / (4 * K * NB), tap = row >> 5, co = row & 31;
            const long off = ((long)((n0 + co) * K + tap) * 64 + ch * 32) * 2 + part * 16;
            *reinterpret_cast<uint4*>(Wh + ch * G::WHB + swz(row, part)) = *reinterpret_cast<const uint4*>(w16 + off);
        }
        // planes: host layout [plane][half n0/32][KP][32 co][32 B], then scales [plane][half][KP][32][2]
        const char* wm = reinterpret_cast<const char*>(p.W_mx);
        for (int c = tid; c < 2 * G::WQB / 16; c += 512) {
            const int pl = c / (G::WQB / 16), r = c % (G::WQB / 16), row = r >> 1, hf = r & 1;
            *reinterpret_cast<uint4*>(Wq + pl * G::WQB + row * 32 + (hf << 4)) =
                *reinterpret_cast<const uint4*>(wm + (size_t)(pl * 2 + (n0 >> 5)) * G::WQB + (size_t)r * 16);
        }
        for (int c = tid; c < 2 * G::WSB / 16; c += 512) {
            const int pl = c / (G::WSB / 16), r = c % (G::WSB / 16);
            *reinterpret_cast<uint4*>(Wsc + pl * G::WSB + r * 16) =
                *reinterpret_cast<const uint4*>(wm + (size_t)4 * G::WQB + (size_t)(pl * 2 + (n0 >> 5)) * G::WSB + (size_t)r * 16);
        }
    }
    const int er = lane >> 2, eg = lane & 3, eco = n0 + eg * 8;          // coalesced side of the epilogue: 4 lanes per row, 16 rows per instruction
    const unsigned erbit = 1u << er;
    f32x2 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = p.bias ? f32x2{p.bias[eco + 2 * q], p.bias[eco + 2 * q + 1]} : f32x2{0.f, 0.f};
    const f32x2 out_scale2 = f32x2{p.out_scale, p.out_scale};
    const bool act_lrelu = p.act == ACT_LRELU;
    const f32x2 act_slope2 = f32x2{p.act_slope, p.act_slope};
    const bool has_planes = p.mxo_h != nullptr;
    // tuning switches (tools/bench_c64.py; results are garbage by design): reserved0 bit 4 no plane stores, 5 no fp32 stores, 6 no slab
    // requests after the first, 7 no MFMAs, 8 no LDS slab writes after the first, 9 no plane quantisation at all
    const int abl = p.reserved0 >> 4;
    const f32x2 mxo_slope2 = f32x2{p.mxo_slope, p.mxo_slope};
    const bool mxo_act = p.mxo_slope != 1.0f, scaled = p.out_scale != 1.0f;          // (x * 1 is exact)
    const f32x2 res_inv2 = f32x2{p.res_inv_slope, p.res_inv_slope};
    float* const o32 = p.out32;
    const uint8_t* vptr = p.row_valid ? p.row_valid : g_row_always_valid;
    const int vshift = p.row_valid ? p.valid_shift : 31;

    // ---- slab staging (plain copies of the producer's planes): per thread five 16-byte units of the hi plane (unit u = tid + 512 i -> row u >> 3,
    // part u & 7), three of the code planes (u -> plane u / 640, row (u % 640) >> 1, half u & 1) and, threads < 160, one of the scale planes
    // (plane tid / 80, rows 4 (tid % 80) ..)
    u32x4 xh[5], xc[3], xs4;          // (native vectors: HIP's uint4 struct arrays stayed in scratch here)
    const int last_row = 255 + (K - 1) * dil;                            // last slab row the conv reads
    // lane offsets are fixed for the whole launch (32-bit, relative to the item's first slab row); per item only the scalar bases move
    // (no run-time index into the kernel argument's pointer arrays either: that copies the whole argument struct to scratch)
    unsigned hoff[5], coff[3];
    int cpl[3], crow[3];
#pragma unroll
    for (int i = 0; i < 5; ++i) hoff[i] = (unsigned)min((tid >> 3) + 64 * i, last_row) * 128u + (tid & 7) * 16u;   // rows beyond the span re-read its last row
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = min(tid + 512 * j, 2 * XROWS * 2 - 1);
        cpl[j] = u / (2 * XROWS); crow[j] = (u % (2 * XROWS)) >> 1;
        coff[j] = (unsigned)min(crow[j], last_row) * 32u + (tid & 1) * 16u;
    }
    const int chalf = tid & 1;
    // (threads >= 160 duplicate unit 159: a load whose result some lanes never consume leaves a pending write on its registers, and the next
    // item's request for them then waits -- vmcnt is in order -- for every store issued since: 2.5 k cycles per item in the first version)
    const int su = min(tid, 159);
    const int spl = su / 80, srow = (su % 80) * 4;
    const unsigned soff = (unsigned)srow * 4u;
    const char* const hbase = reinterpret_cast<const char*>(p.A);
    const char* const cbase0 = reinterpret_cast<const char*>(p.mx_x4[0]);
    const char* const cbase1 = reinterpret_cast<const char*>(p.mx_x4[1]);
    const char* const sbase = reinterpret_cast<const char*>(spl ? p.mx_xs[1] : p.mx_xs[0]);
#define EV_C64_GLOAD(TILE)                                                                                 \
    {                                                                                                      \
        const long r00_ = (long)(TILE) * 256 - hlo;                                                        \
        const char* hb_ = hbase + r00_ * 128;                                                              \
        const char* c0_ = cbase0 + r00_ * 32;                                                              \
        const char* c1_ = cbase1 + r00_ * 32;                                                              \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) xh[i] = *reinterpret_cast<const u32x4*>(hb_ + hoff[i]); \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) xc[j] = *reinterpret_cast<const u32x4*>((cpl[j] ? c1_ : c0_) + coff[j]); \
        xs4 = *reinterpret_cast<const u32x4*>(sbase + r00_ * 4 + soff);                                    \
    }
#define EV_C64_SSTORE()                                                                                    \
    {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                    \
            const int r_ = (tid >> 3) + 64 * i;                                                            \
            *reinterpret_cast<u32x4*>(Xh + ((tid >> 2) & 1) * G::XHC + swz(r_, tid & 3)) = xh[i];      \
        }                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                      \
            *reinterpret_cast<u32x4*>(Xq + cpl[j] * XROWS * 32 + crow[j] * 32 + (chalf << 4)) = xc[j];                           \
        *reinterpret_cast<u32x4*>(Xsc + spl * XROWS * 4 + srow * 4) = xs4;                                 \
    }

    EV_C64_GLOAD(EV_C64_TILE(item))
    EV_C64_SSTORE()
    __syncthreads();
    const int wrow0 = wave * 32 + fr;
    // EV_C64_TIMING (tuning builds: build.py --variant c64t EV_C64_TIMING): wave 0 of block 0 sums the cycles of each section of an item and writes
    // them over out32[0..6] at the end (tools/bench_c64.py); every s_memtime is a scalar-memory round trip, so the product build has none
#ifdef EV_C64_TIMING
    unsigned long long tk_prev = __builtin_readcyclecounter();
    unsigned tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define EV_C64_TICK(I) { const unsigned long long t_ = __builtin_readcyclecounter(); tk[I] += (unsigned)(t_ - tk_prev); tk_prev = t_; }
#else
#define EV_C64_TICK(I)
#endif
    for (; item < nitems; item += gridDim.x) {
        const int tile = EV_C64_TILE(item);
        const bool item_ok = EV_C64_OK(item);
        const int nitem = item + (int)gridDim.x;
        const int ntile = EV_C64_TILE(min(nitem, nitems - 1));          // clamped: the last prefetch of a block is never used
        const int m0 = tile * 256;
        // ---------------- memory requests of this item, oldest first
        uint8_t vb = vptr[(m0 + wave * 32 + (lane & 31)) >> vshift];
        float4 resv[2][2], accin[2][2];
        u32x4 rph[2];
        unsigned rpc[2], rps[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const long t = m0 + wave * 32 + it * 16 + er;
            if constexpr (MODE >= 1 && RPL) {
                rph[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.res) + (t * 64 + eco) * 2);
                rpc[it] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(p.res_x4) + t * 32 + (n0 >> 5) * 16 + eg * 4);
                rps[it] = reinterpret_cast<const uint8_t*>(p.res_xs)[t * 4 + (n0 >> 5)];
            } else if constexpr (MODE >= 1) {
                const float* rp = reinterpret_cast<const float*>(p.res) + t * p.ldres + eco;
                resv[it][0] = *reinterpret_cast<const float4*>(rp); resv[it][1] = *reinterpret_cast<const float4*>(rp + 4);
            }
            if constexpr (MODE == 2) {
                const float* ap = p.acc32 + t * p.ldacc + eco;
                accin[it][0] = *reinterpret_cast<const float4*>(ap); accin[it][1] = *reinterpret_cast<const float4*>(ap + 4);
            }
        }
        if (!(abl & 4)) { EV_C64_GLOAD(ntile) }
        __builtin_amdgcn_sched_barrier(0);
        EV_C64_TICK(0)
        f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---------------- fp16 hi x hi: tap by tap, two 32-channel chunks each
        if (!(abl & 8)) {
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int r0 = wrow0 + t * dil;
            const int xo = r0 * 64 + ((fq ^ ((r0 >> 1) & 3)) << 4);
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 wf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const uint4*>(Wh + ch * G::WHB + swz(t * 32 + a * 16 + fr, fq));
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    uint4 xf = *reinterpret_cast<const uint4*>(Xh + ch * G::XHC + xo + b * 16 * 64);
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wf[a]), *reinterpret_cast<half8*>(&xf), acc[a][b], 0, 0, 0);
                }
            }
        }
        // ---------------- the two fp4 cross terms, two taps per MFMA
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int tw = 2 * g + (fq >> 1), hf = fq & 1;          // this lane's tap (weights: zero codes beyond K) and channel half
            const int rq = wrow0 + min(tw, K - 1) * dil;            // its operand row (a real row for the padded tap)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                uint4 wq[2], xq[2];
                int ws[2], xs[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int wr = tw * 32 + a * 16 + fr;
                    wq[a] = *reinterpret_cast<const uint4*>(Wq + pl * G::WQB + wr * 32 + (hf << 4));
                    ws[a] = *reinterpret_cast<const uint8_t*>(Wsc + pl * G::WSB + wr * 2 + hf);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int rr = rq + b * 16;
                    xq[b] = *reinterpret_cast<const uint4*>(Xq + pl * XROWS * 32 + rr * 32 + (hf << 4));
                    xs[b] = *reinterpret_cast<const uint8_t*>(Xsc + pl * XROWS * 4 + rr * 4 + hf);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int a = 0; a < 2; ++a) mfma_mx_inplace(acc[a][b], wq[a], xq[b], ws[a], xs[b]);
            }
        }
        mfma_asm_fence(acc);
        }
        EV_C64_TICK(1)
        __syncthreads();          // every wave is done with the slab
        EV_C64_TICK(2)
        if (!(abl & 16)) { EV_C64_SSTORE() }          // the next item's slab replaces it
        EV_C64_TICK(3)
        // ---------------- epilogue: 16-row passes through the wave's transposing scratch
        const unsigned vmask = (unsigned)__builtin_amdgcn_ballot_w64(vb != 0 && lane < 32);
        f32x2 vv[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {          // the two 16-row transposes first (LDS only) ...
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int a = 0; a < 2; ++a) *reinterpret_cast<f32x4*>(es + fr * EPITCH + (a * 16 + 4 * fq) * 4) = acc[a][it];
            __builtin_amdgcn_wave_barrier();
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(es + er * EPITCH + eg * 32);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(es + er * EPITCH + eg * 32 + 16);
            vv[it][0] = f32x2{v0[0], v0[1]}; vv[it][1] = f32x2{v0[2], v0[3]}; vv[it][2] = f32x2{v1[0], v1[1]}; vv[it][3] = f32x2{v1[2], v1[3]};
        }
        // ... then ALL the arithmetic (it consumes the residual / accumulate-in rows: once a store has been issued, a wait for an older load
        // is a wait for that store too, and with a run-time number of stores per row group it is vmcnt(0)), and only then the stores
        uint4 pho[2];
        unsigned pch[2], pcl[2], pbh[2], pbl[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool valid = (vmask & (erbit << (it * 16))) != 0u;
            f32x2* v = vv[it];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += bv[q];
            if (act_lrelu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = lrelu2(v[q], act_slope2);
            }
            if constexpr (MODE >= 1) {
                f32x2 rr[4];
                if constexpr (RPL) {          // x = lrelu^-1(hi + code * scale) = min(a, a * inv)
                    const half2v* h = reinterpret_cast<const half2v*>(&rph[it]);
                    const float sc = __uint_as_float(rps[it] << 23);
                    rr[0] = __builtin_convertvector(h[0], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(rpc[it], sc, 0);
                    rr[1] = __builtin_convertvector(h[1], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(rpc[it], sc, 1);
                    rr[2] = __builtin_convertvector(h[2], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(rpc[it], sc, 2);
                    rr[3] = __builtin_convertvector(h[3], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(rpc[it], sc, 3);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 t2 = rr[q] * res_inv2;
                        rr[q] = f32x2{min_raw(rr[q][0], t2[0]), min_raw(rr[q][1], t2[1])};
                    }
                } else {
                    rr[0] = f32x2{resv[it][0].x, resv[it][0].y}; rr[1] = f32x2{resv[it][0].z, resv[it][0].w};
                    rr[2] = f32x2{resv[it][1].x, resv[it][1].y}; rr[3] = f32x2{resv[it][1].z, resv[it][1].w};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += rr[q];
                if (scaled) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= out_scale2;
                }
            }
            if constexpr (MODE == 2) {
                v[0] += f32x2{accin[it][0].x, accin[it][0].y}; v[1] += f32x2{accin[it][0].z, accin[it][0].w};
                v[2] += f32x2{accin[it][1].x, accin[it][1].y}; v[3] += f32x2{accin[it][1].z, accin[it][1].w};
            }
            if (vmask != 0xffffffffu) {            // (wave-uniform: a 32-row group with rows outside the utterances)
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q][0] = valid ? v[q][0] : 0.f; v[q][1] = valid ? v[q][1] : 0.f; }
            }
            if (has_planes && !(abl & 32)) {            // the next conv's operand: its leaky-relu, then the planes (this quad = one 32-channel block of the row)
                f32x2 am[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) am[q] = v[q];
                if (mxo_act) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) am[q] = lrelu2(v[q], mxo_slope2);
                }
                mx_quant8(am, pho[it], pch[it], pcl[it], pbh[it], pbl[it]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const long t = m0 + wave * 32 + it * 16 + er;
            const f32x2* v = vv[it];
            if (item_ok && o32 && !(abl & 2)) {
                float* op = o32 + t * p.ldo + eco;
                *reinterpret_cast<float4*>(op) = make_float4(v[0][0], v[0][1], v[1][0], v[1][1]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[2][0], v[2][1], v[3][0], v[3][1]);
            }
            if (has_planes && item_ok && !(abl & 33)) {
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.mxo_h) + (t * 64 + eco) * 2) = pho[it];
                *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.mxo_q4[0]) + t * 32 + (n0 >> 5) * 16 + eg * 4) = pch[it];
                *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.mxo_q4[1]) + t * 32 + (n0 >> 5) * 16 + eg * 4) = pcl[it];
                if (eg == 0) {
                    reinterpret_cast<uint8_t*>(p.mxo_qs[0])[t * 4 + (n0 >> 5)] = (uint8_t)pbh[it];
                    reinterpret_cast<uint8_t*>(p.mxo_qs[1])[t * 4 + (n0 >> 5)] = (uint8_t)pbl[it];
                }
            }
        }
        EV_C64_TICK(4)
        __syncthreads();          // the new slab is complete
        EV_C64_TICK(5)
#ifdef EV_C64_TIMING
        tk[6] += 1;
#endif
    }
#ifdef EV_C64_TIMING
    if ((abl & 64) && blockIdx.x == 0 && tid == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) reinterpret_cast<unsigned*>(p.out32)[i] = tk[i];
    }
#endif
#undef EV_C64_TICK
#undef EV_C64_GLOAD
#undef EV_C64_SSTORE
#undef EV_C64_TILE
#undef EV_C64_HALF
#undef EV_C64_OK
}


// ---- Round 4: the same conv with the block's eight waves split into TWO GROUPS one barrier apart (the phased conv-GEMM kernel's schedule applied to
// the persistent kernel).  conv_c64_mx_kernel runs its eight waves in lock-step: [requests] [MFMAs] barrier [slab -> LDS] [epilogue] barrier, and an item
// costs the SUM of the phases (per 256-row item ~11 k cycles against 1.3-4.3 k of matrix work; MFMA-busy 0.24).  Here waves 0-3 (one per SIMD) own rows
// [0, 128) of the block's 256-row item and waves 4-7 (their SIMD partners) rows [128, 256), each group with its OWN slab buffer; nothing but the read-only
// weights is shared, so the groups need no ordering between them -- but s_barrier is block-wide, and both run the same two barriers per item: group 1 enters
// its loop one barrier late, so between any two barriers one wave of every SIMD is in its matrix phase (LDS fragment reads + MFMAs) while its partner
// requests / copies the next slab and runs the VALU + store work of its epilogue.  Same arithmetic per output element as conv_c64_mx_kernel (bit-identical
// results); a slab is 128 + (K - 1) d rows per group (<= 192; K = 11: d <= 5, 184 rows -- 163 456 of the CU's 163 840 LDS bytes).
template <int K>
struct Conv64Mx2Geom {
    static constexpr int C = 64, NB = 32, KG = (K + 1) / 2, KP = KG * 2;
    static constexpr int GR = 128, XR = (K == 11) ? 184 : 192, MAXSPAN = XR - GR, EPITCH = NB * 4 + 16;
    static constexpr int XHC = XR * 64 + 64;           // one K-chunk of a group's hi plane
    static constexpr int WHB = K * NB * 64, WQB = KP * NB * 32, WSB = KP * NB * 2;
    static constexpr int OFF_WH = 0, OFF_WQ = 2 * WHB, OFF_WS = OFF_WQ + 2 * WQB, OFF_X = OFF_WS + 2 * WSB;
    static constexpr int XG = 2 * XHC + 2 * XR * 32 + 2 * XR * 4;       // one group's slab: hi (2 chunks), codes (2 planes), scales (2 planes)
    static constexpr int OFF_ES = OFF_X + 2 * XG;
    static constexpr int TOTAL = OFF_ES + 8 * 16 * EPITCH;
    static constexpr int HI_N = (XR * 8 + 255) / 256;   // 16-byte units per thread: hi plane / code planes (4 XR units) / scale planes (8 XR bytes)
    static_assert(OFF_X % 16 == 0 && XG % 16 == 0 && OFF_ES % 16 == 0 && TOTAL <= 160 * 1024 && 4 * XR <= 768 && 8 * XR / 16 <= 256, "LDS plan");
};

template <int K, int MODE, bool RPL = false>
__global__ __launch_bounds__(512, 1) void conv_c64_mx2_kernel(const ConvGemmParams p) {
    using G = Conv64Mx2Geom<K>;
    constexpr int NB = G::NB, KG = G::KG, XR = G::XR, EPITCH = G::EPITCH, HI_N = G::HI_N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Wh = smem + G::OFF_WH;
    char* const Wq = smem + G::OFF_WQ;
    char* const Wsc = smem + G::OFF_WS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave >> 2, lw = wave & 3, ltid = tid & 255;          // group, wave inside the group, thread inside the group
    char* const Xg = smem + G::OFF_X + gi * G::XG;                       // this group's slab
    char* const Xh = Xg;                                                 // [chunk][row][64 B] swizzled
    char* const Xq = Xg + 2 * G::XHC;                                    // [plane][row][32 B]
    char* const Xsc = Xq + 2 * XR * 32;                                  // [plane][row][4]
    char* const es = smem + G::OFF_ES + wave * 16 * EPITCH;
    const int fr = lane & 15, fq = lane >> 4;
    const int dil = p.dil, hlo = p.center * dil;
    const int ntiles = p.M / 256, nitems = ((ntiles + 7) >> 3) << 4;
#define EV_C64_TILE(I) min((((I) >> 4) << 3) + ((I) & 7), ntiles - 1)
#define EV_C64_OK(I) (((((I) >> 4) << 3) + ((I) & 7)) < ntiles)
#define EV_C64_HALF(I) (((I) >> 3) & 1)
    int item = blockIdx.x;
    if (item >= nitems) return;
    const int n0 = EV_C64_HALF(item) * NB;

    // ---- this block's half of the weights -> LDS (all 512 threads)
    {
        const char* w16 = reinterpret_cast<const char*>(p.W);
        for (int c = tid; c < 2 * K * NB * 4; c += 512) {
            const int part = c & 3, row = (c >> 2) % (K * NB), ch = c / (4 * K * NB), tap = row >> 5, co = row & 31;
            const long off = ((long)((n0 + co) * K + tap) * 64 + ch * 32) * 2 + part * 16;
            *reinterpret_cast<uint4*>(Wh + ch * G::WHB + swz(row, part)) = *reinterpret_cast<const uint4*>(w16 + off);
        }
        const char* wm = reinterpret_cast<const char*>(p.W_mx);
        for (int c = tid; c < 2 * G::WQB / 16; c += 512) {
            const int pl = c / (G::WQB / 16), r = c % (G::WQB / 16), row = r >> 1, hf = r & 1;
            *reinterpret_cast<uint4*>(Wq + pl * G::WQB + row * 32 + (hf << 4)) =
                *reinterpret_cast<const uint4*>(wm + (size_t)(pl * 2 + (n0 >> 5)) * G::WQB + (size_t)r * 16);
        }
        for (int c = tid; c < 2 * G::WSB / 16; c += 512) {
            const int pl = c / (G::WSB / 16), r = c % (G::WSB / 16);
            *reinterpret_cast<uint4*>(Wsc + pl * G::WSB + r * 16) =
                *reinterpret_cast<const uint4*>(wm + (size_t)4 * G::WQB + (size_t)(pl * 2 + (n0 >> 5)) * G::WSB + (size_t)r * 16);
        }
    }
    const int er = lane >> 2, eg = lane & 3, eco = n0 + eg * 8;
    const unsigned erbit = 1u << er;
    f32x2 bv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[q] = p.bias ? f32x2{p.bias[eco + 2 * q], p.bias[eco + 2 * q + 1]} : f32x2{0.f, 0.f};
    const f32x2 out_scale2 = f32x2{p.out_scale, p.out_scale};
    const bool act_lrelu = p.act == ACT_LRELU;
    const f32x2 act_slope2 = f32x2{p.act_slope, p.act_slope};
    const bool has_planes = p.mxo_h != nullptr;
    const f32x2 mxo_slope2 = f32x2{p.mxo_slope, p.mxo_slope};
    const bool mxo_act = p.mxo_slope != 1.0f, scaled = p.out_scale != 1.0f;
    const f32x2 res_inv2 = f32x2{p.res_inv_slope, p.res_inv_slope};
    float* const o32 = p.out32;
    const uint8_t* vptr = p.row_valid ? p.row_valid : g_row_always_valid;
    const int vshift = p.row_valid ? p.valid_shift : 31;

    // ---- slab staging of a group (256 threads, plain copies of the producer's planes): HI_N units of the hi plane per thread (unit u = ltid + 256 i ->
    // row u >> 3, part u & 7), three of the code planes (u -> plane u / (2 XR), row (u % (2 XR)) >> 1, half u & 1), one of the scale planes (16 bytes = 4 rows).
    // Units beyond the slab duplicate its last unit, rows beyond the conv's span re-read the span's last row (same data to the same address: harmless, and
    // every lane consumes what it loads -- an un-consumed load leaves a pending write on its registers).
    u32x4 xh[HI_N], xc[3], xs4;
    const int last_row = G::GR - 1 + (K - 1) * dil;
    unsigned hoff[HI_N], coff[3];
    int hdst[HI_N], cdst[3];
#pragma unroll
    for (int i = 0; i < HI_N; ++i) {
        const int r = min((ltid >> 3) + 32 * i, XR - 1);
        hoff[i] = (unsigned)min(r, last_row) * 128u + (ltid & 7) * 16u;
        hdst[i] = ((ltid >> 2) & 1) * G::XHC + swz(r, ltid & 3);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = min(ltid + 256 * j, 4 * XR - 1);
        const int pl = u / (2 * XR), row = (u % (2 * XR)) >> 1, hf = u & 1;
        coff[j] = (unsigned)min(row, last_row) * 32u + hf * 16u + (pl ? 0x80000000u : 0u);          // (bit 31: the remainder's code plane)
        cdst[j] = pl * XR * 32 + row * 32 + (hf << 4);
    }
    const int su = min(ltid, 8 * XR / 16 - 1);
    const int spl = su / (XR / 4), srow = (su % (XR / 4)) * 4;
    const char* const hbase = reinterpret_cast<const char*>(p.A);
    const char* const cbase0 = reinterpret_cast<const char*>(p.mx_x4[0]);
    const char* const cbase1 = reinterpret_cast<const char*>(p.mx_x4[1]);
    const char* const sbase = reinterpret_cast<const char*>(spl ? p.mx_xs[1] : p.mx_xs[0]) + srow * 4;
    const int grow = gi * G::GR;                                     // the group's first row inside the block's 256-row item
#define EV_C64_GLOAD(TILE)                                                                                 \
    {                                                                                                      \
        const long r00_ = (long)(TILE) * 256 + grow - hlo;                                                 \
        const char* hb_ = hbase + r00_ * 128;                                                              \
        const char* c0_ = cbase0 + r00_ * 32;                                                              \
        const char* c1_ = cbase1 + r00_ * 32;                                                              \
        _Pragma("unroll") for (int i = 0; i < HI_N; ++i) xh[i] = *reinterpret_cast<const u32x4*>(hb_ + hoff[i]); \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) xc[j] = *reinterpret_cast<const u32x4*>(((coff[j] >> 31) ? c1_ : c0_) + (coff[j] & 0x7fffffffu)); \
        xs4 = *reinterpret_cast<const u32x4*>(sbase + r00_ * 4);                                           \
    }
#define EV_C64_SSTORE()                                                                                    \
    {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < HI_N; ++i) *reinterpret_cast<u32x4*>(Xh + hdst[i]) = xh[i];  \
        _Pragma("unroll") for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4*>(Xq + cdst[j]) = xc[j];     \
        *reinterpret_cast<u32x4*>(Xsc + spl * XR * 4 + srow * 4) = xs4;                                    \
    }

    // block barrier of the main loop: LDS traffic retired, but NOT vmcnt (a __syncthreads() here would wait for the epilogue's global stores and for
    // the next slab's requests at every phase boundary; what a phase needs from memory the compiler's own register waits provide)
#define EV_C64_GROUP_BARRIER()                                   \
    {                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        __builtin_amdgcn_sched_barrier(0);                       \
    }
    EV_C64_GLOAD(EV_C64_TILE(item))
    EV_C64_SSTORE()
    __syncthreads();
    if (gi == 1) __builtin_amdgcn_s_barrier();          // group 1 runs one barrier behind group 0 from here on
    const int wrow0 = lw * 32 + fr;
    for (; item < nitems; item += gridDim.x) {
        const int tile = EV_C64_TILE(item);
        const bool item_ok = EV_C64_OK(item);
        const int nitem = item + (int)gridDim.x;
        const int ntile = EV_C64_TILE(min(nitem, nitems - 1));
        const int m0 = tile * 256 + grow;
        // ---------------- memory requests of this item, oldest first
        uint8_t vb = vptr[(m0 + lw * 32 + (lane & 31)) >> vshift];
        float4 resv[2][2], accin[2][2];
        u32x4 rph[2];
        unsigned rpc[2], rps[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const long t = m0 + lw * 32 + it * 16 + er;
            if constexpr (MODE >= 1 && RPL) {
                rph[it] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.res) + (t * 64 + eco) * 2);
                rpc[it] = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(p.res_x4) + t * 32 + (n0 >> 5) * 16 + eg * 4);
                rps[it] = reinterpret_cast<const uint8_t*>(p.res_xs)[t * 4 + (n0 >> 5)];
            } else if constexpr (MODE >= 1) {
                const float* rp = reinterpret_cast<const float*>(p.res) + t * p.ldres + eco;
                resv[it][0] = *reinterpret_cast<const float4*>(rp); resv[it][1] = *reinterpret_cast<const float4*>(rp + 4);
            }
            if constexpr (MODE == 2) {
                const float* ap = p.acc32 + t * p.ldacc + eco;
                accin[it][0] = *reinterpret_cast<const float4*>(ap); accin[it][1] = *reinterpret_cast<const float4*>(ap + 4);
            }
        }
        EV_C64_GLOAD(ntile)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---------------- matrix phase: fp16 hi x hi tap by tap (two 32-channel chunks each), then the two fp4 cross terms, two taps per MFMA
#pragma unroll
        for (int t = 0; t < K; ++t) {
            const int r0 = wrow0 + t * dil;
            const int xo = r0 * 64 + ((fq ^ ((r0 >> 1) & 3)) << 4);
#pragma unroll
            for (int ch = 0; ch < 2; ++ch) {
                uint4 wf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const uint4*>(Wh + ch * G::WHB + swz(t * 32 + a * 16 + fr, fq));
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    uint4 xf = *reinterpret_cast<const uint4*>(Xh + ch * G::XHC + xo + b * 16 * 64);
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wf[a]), *reinterpret_cast<half8*>(&xf), acc[a][b], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int tw = 2 * g + (fq >> 1), hf = fq & 1;
            const int rq = wrow0 + min(tw, K - 1) * dil;
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                uint4 wq[2], xq[2];
                int ws[2], xs[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int wr = tw * 32 + a * 16 + fr;
                    wq[a] = *reinterpret_cast<const uint4*>(Wq + pl * G::WQB + wr * 32 + (hf << 4));
                    ws[a] = *reinterpret_cast<const uint8_t*>(Wsc + pl * G::WSB + wr * 2 + hf);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int rr = rq + b * 16;
                    xq[b] = *reinterpret_cast<const uint4*>(Xq + pl * XR * 32 + rr * 32 + (hf << 4));
                    xs[b] = *reinterpret_cast<const uint8_t*>(Xsc + pl * XR * 4 + rr * 4 + hf);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int a = 0; a < 2; ++a) mfma_mx_inplace(acc[a][b], wq[a], xq[b], ws[a], xs[b]);
            }
        }
        mfma_asm_fence(acc);
        EV_C64_GROUP_BARRIER()                 // every wave of the group is done with the slab (the other group: its new slab is complete)
        EV_C64_SSTORE()                        // the next item's slab replaces it
        // ---------------- epilogue: 16-row passes through the wave's transposing scratch
        const unsigned vmask = (unsigned)__builtin_amdgcn_ballot_w64(vb != 0 && lane < 32);
        f32x2 vv[2][4];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int a = 0; a < 2; ++a) *reinterpret_cast<f32x4*>(es + fr * EPITCH + (a * 16 + 4 * fq) * 4) = acc[a][it];
            __builtin_amdgcn_wave_barrier();
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(es + er * EPITCH + eg * 32);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(es + er * EPITCH + eg * 32 + 16);
            vv[it][0] = f32x2{v0[0], v0[1]}; vv[it][1] = f32x2{v0[2], v0[3]}; vv[it][2] = f32x2{v1[0], v1[1]}; vv[it][3] = f32x2{v1[2], v1[3]};
        }
        uint4 pho[2];
        unsigned pch[2], pcl[2], pbh[2], pbl[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const bool valid = (vmask & (erbit << (it * 16))) != 0u;
            f32x2* v = vv[it];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += bv[q];
            if (act_lrelu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = lrelu2(v[q], act_slope2);
            }
            if constexpr (MODE >= 1) {
                f32x2 rr[4];
                if constexpr (RPL) {
                    const half2v* h = reinterpret_cast<const half2v*>(&rph[it]);
                    const float sc = __uint_as_float(rps[it] << 23);
                    rr[0] = __builtin_convertvector(h[0], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(rpc[it], sc, 0);
                    rr[1] = __builtin_convertvector(h[1], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(rpc[it], sc, 1);
                    rr[2] = __builtin_convertvector(h[2], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(rpc[it], sc, 2);
                    rr[3] = __builtin_convertvector(h[3], f32x2) + __builtin_amdgcn_cvt_scalef32_pk_f32_fp4(rpc[it], sc, 3);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x2 t2 = rr[q] * res_inv2;
                        rr[q] = f32x2{min_raw(rr[q][0], t2[0]), min_raw(rr[q][1], t2[1])};
                    }
                } else {
                    rr[0] = f32x2{resv[it][0].x, resv[it][0].y}; rr[1] = f32x2{resv[it][0].z, resv[it][0].w};
                    rr[2] = f32x2{resv[it][1].x, resv[it][1].y}; rr[3] = f32x2{resv[it][1].z, resv[it][1].w};
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += rr[q];
                if (scaled) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= out_scale2;
                }
            }
            if constexpr (MODE == 2) {
                v[0] += f32x2{accin[it][0].x, accin[it][0].y}; v[1] += f32x2{accin[it][0].z, accin[it][0].w};
                v[2] += f32x2{accin[it][1].x, accin[it][1].y}; v[3] += f32x2{accin[it][1].z, accin[it][1].w};
            }
            if (vmask != 0xffffffffu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q][0] = valid ? v[q][0] : 0.f; v[q][1] = valid ? v[q][1] : 0.f; }
            }
            if (has_planes) {
                f32x2 am[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) am[q] = v[q];
                if (mxo_act) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) am[q] = lrelu2(v[q], mxo_slope2);
                }
                mx_quant8(am, pho[it], pch[it], pcl[it], pbh[it], pbl[it]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const long t = m0 + lw * 32 + it * 16 + er;
            const f32x2* v = vv[it];
            if (item_ok && o32) {
                float* op = o32 + t * p.ldo + eco;
                *reinterpret_cast<float4*>(op) = make_float4(v[0][0], v[0][1], v[1][0], v[1][1]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[2][0], v[2][1], v[3][0], v[3][1]);
            }
            if (has_planes && item_ok) {
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.mxo_h) + (t * 64 + eco) * 2) = pho[it];
                *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.mxo_q4[0]) + t * 32 + (n0 >> 5) * 16 + eg * 4) = pch[it];
                *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(p.mxo_q4[1]) + t * 32 + (n0 >> 5) * 16 + eg * 4) = pcl[it];
                if (eg == 0) {
                    reinterpret_cast<uint8_t*>(p.mxo_qs[0])[t * 4 + (n0 >> 5)] = (uint8_t)pbh[it];
                    reinterpret_cast<uint8_t*>(p.mxo_qs[1])[t * 4 + (n0 >> 5)] = (uint8_t)pbl[it];
                }
            }
        }
        EV_C64_GROUP_BARRIER()                 // the group's new slab is complete (the other group: done with its old one)
    }
    if (gi == 0) __builtin_amdgcn_s_barrier();          // barrier counts of the two groups match
#undef EV_C64_GROUP_BARRIER
#undef EV_C64_GLOAD
#undef EV_C64_SSTORE
#undef EV_C64_TILE
#undef EV_C64_HALF
#undef EV_C64_OK
}

template <int K>
static hipError_t conv64_mx_attr() {
    hipError_t e = hipSuccess, r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx_kernel<K, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64MxGeom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx_kernel<K, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64MxGeom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx_kernel<K, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64MxGeom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx_kernel<K, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64MxGeom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx_kernel<K, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64MxGeom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx2_kernel<K, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64Mx2Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx2_kernel<K, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64Mx2Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx2_kernel<K, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64Mx2Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx2_kernel<K, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64Mx2Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_c64_mx2_kernel<K, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, Conv64Mx2Geom<K>::TOTAL); if (r != hipSuccess) e = r;
    return e;
}
static hipError_t conv64_mx_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = conv64_mx_attr<3>(); if (r != hipSuccess) e = r;
    r = conv64_mx_attr<7>(); if (r != hipSuccess) e = r;
    r = conv64_mx_attr<11>(); if (r != hipSuccess) e = r;
    return e;
}
// DT_MX call with N = K = 64: A = the input plane set's fp16 hi plane (lda = 64), mx_x4 / mx_xs its code / scale planes, W = fp16 hi parts
// [64][taps][64], W_mx = mxfp4.pack_c64_weight_planes; outputs: out32 and / or the plane set mxo_* (mxo_logC = 6)
static bool conv64_mx_eligible(const ConvGemmParams& p) {
    const bool rare_act = p.act != ACT_NONE && p.act != ACT_LRELU;
    return p.W_mx && p.N == 64 && p.K == 64 && p.lda == 64 && (p.taps == 3 || p.taps == 7 || p.taps == 11) && p.M % 256 == 0 && (p.taps - 1) * p.dil <= MAX_SPAN &&
           p.center * 2 == p.taps - 1 && p.mx_x4[0] && p.mx_x4[1] && p.mx_xs[0] && p.mx_xs[1] && !p.pro_lrelu && (p.out32 || p.mxo_h) && !p.out16 && !p.seq_bias &&
           !p.add16_a && !p.post_lrelu && !p.out32_before_post && !rare_act && (!p.out32 || p.ldo == 64) && !p.acc_h && !p.mxo_partial &&
           (!p.mxo_h || (p.mxo_logC == 6 && p.mxo_q4[0] && p.mxo_q4[1] && p.mxo_qs[0] && p.mxo_qs[1] && p.mxo_slope >= 0.f && p.mxo_slope <= 1.f)) &&
           (!p.res || p.res_dtype == DT_F32 || (p.res_dtype == DT_MX && p.res_x4 && p.res_xs && p.ldres == 64 && p.res_inv_slope >= 1.0f)) &&
           (!p.acc32 || p.res) && (!p.res || p.act == ACT_NONE) && (p.res || p.out_scale == 1.0f) &&        // (out_scale lives in the residual branch)
           !(p.act == ACT_LRELU && !(p.act_slope >= 0.f && p.act_slope <= 1.f));
}
static void launch_conv64_mx(const ConvGemmParams& p, hipStream_t s) {
    const int n_cu = device_cus();
    const int nitems = ((p.M / 256 + 7) / 8) * 16;
    const int grid = nitems <= n_cu ? nitems : (n_cu / 16) * 16;           // a multiple of 16 keeps a block's channel half fixed
    const int mode = p.acc32 ? 2 : (p.res ? 1 : 0);
    const bool rpl = p.res && p.res_dtype == DT_MX;
    // two-group schedule (conv_c64_mx2_kernel) whenever the group's slab holds the conv's span; reserved0 bit 2: in-process A/B (lock-step kernel)
#define EV_C64_LAUNCH2(KK)                                                                                                        \
        if (mode == 2 && rpl) hipLaunchKernelGGL((conv_c64_mx2_kernel<KK, 2, true>), dim3(grid), dim3(512), Conv64Mx2Geom<KK>::TOTAL, s, p);      \
        else if (mode == 1 && rpl) hipLaunchKernelGGL((conv_c64_mx2_kernel<KK, 1, true>), dim3(grid), dim3(512), Conv64Mx2Geom<KK>::TOTAL, s, p); \
        else if (mode == 2) hipLaunchKernelGGL((conv_c64_mx2_kernel<KK, 2>), dim3(grid), dim3(512), Conv64Mx2Geom<KK>::TOTAL, s, p); \
        else if (mode == 1) hipLaunchKernelGGL((conv_c64_mx2_kernel<KK, 1>), dim3(grid), dim3(512), Conv64Mx2Geom<KK>::TOTAL, s, p); \
        else hipLaunchKernelGGL((conv_c64_mx2_kernel<KK, 0>), dim3(grid), dim3(512), Conv64Mx2Geom<KK>::TOTAL, s, p);
    const int span = (p.taps - 1) * p.dil;
    if (!(p.reserved0 & 4) && !(p.reserved0 >> 4)) {
        if (p.taps == 3 && span <= Conv64Mx2Geom<3>::MAXSPAN) { EV_C64_LAUNCH2(3) return; }
        if (p.taps == 7 && span <= Conv64Mx2Geom<7>::MAXSPAN) { EV_C64_LAUNCH2(7) return; }
        if (p.taps == 11 && span <= Conv64Mx2Geom<11>::MAXSPAN) { EV_C64_LAUNCH2(11) return; }
    }
#undef EV_C64_LAUNCH2
#define EV_C64_LAUNCH(KK)                                                                                                        \
        if (mode == 2 && rpl) hipLaunchKernelGGL((conv_c64_mx_kernel<KK, 2, true>), dim3(grid), dim3(512), Conv64MxGeom<KK>::TOTAL, s, p);      \
        else if (mode == 1 && rpl) hipLaunchKernelGGL((conv_c64_mx_kernel<KK, 1, true>), dim3(grid), dim3(512), Conv64MxGeom<KK>::TOTAL, s, p); \
        else if (mode == 2) hipLaunchKernelGGL((conv_c64_mx_kernel<KK, 2>), dim3(grid), dim3(512), Conv64MxGeom<KK>::TOTAL, s, p); \
        else if (mode == 1) hipLaunchKernelGGL((conv_c64_mx_kernel<KK, 1>), dim3(grid), dim3(512), Conv64MxGeom<KK>::TOTAL, s, p); \
        else hipLaunchKernelGGL((conv_c64_mx_kernel<KK, 0>), dim3(grid), dim3(512), Conv64MxGeom<KK>::TOTAL, s, p);
    switch (p.taps) {
        case 3: EV_C64_LAUNCH(3) break;
        case 7: EV_C64_LAUNCH(7) break;
        default: EV_C64_LAUNCH(11) break;
    }
#undef EV_C64_LAUNCH
}
