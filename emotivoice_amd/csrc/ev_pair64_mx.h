// Fused HiFi-GAN ResBlock pair at C = 64, k = 3 (stage 2) in the "MX" arithmetic, plane sets in / plane sets out (round 4):
//
//        xt  = leaky_relu(c1(a) + b1, .1)             a = the input plane set's tensor = leaky_relu(x, .1); c1 = Conv1d(64, 64, 3, dilation d)
//        out = epilogue(c2(xt) + b2 + x)              c2 = Conv1d(64, 64, 3, dilation 1); x rebuilt from the plane set (hi + Q4(lo), inverse leaky-relu)
//
// Layer-wise (conv_c64_mx_kernel twice) this pair moves 14.8 bytes per element through HBM -- conv1 planes in / planes out, conv2 planes + residual planes in,
// planes out -- and its two launches run at 3.8-4.5 TB/s: the k = 3 chain of stage 2 is HBM-bound (2.8 ms of the 11.6 ms stage at B = 32 x 1024 frames).
// Fused, xt never leaves LDS and the residual comes from the slab that conv1 reads anyway: 6.1 bytes per element.  k = 7 / 11 do not fit (both convs' weights
// for all 64 output channels: 164 / 253 KB), and they are LDS- / MFMA-bound layer-wise, not HBM-bound.
//
// One persistent 8-wave block per CU, 128-row tiles (BMO = 126 output rows):
//   * waves = 4 row groups (32 rows) x 2 output-channel halves (32 channels = one MX scale block); both convs' weights stay in LDS for the whole launch
//     (fp16 hi parts 48 KB, fp4 planes + scales 34 KB; host layout of the planes: mxfp4.pack_c64_weight_planes, both halves);
//   * the slab of the next tile (128 + 2 + 2 d rows of planes: plain copies of what the producer's epilogue wrote) is requested into registers at the top of a
//     tile and replaces the current one after conv1 (the mid-tile barrier); the residual rows of the CURRENT tile are lifted from the slab into registers
//     before that;
//   * conv1's result gets bias + leaky-relu + sequence-edge zeroing in registers and goes to LDS as a plane set (the quantiser of ev_pair_mx.h: a wave's 32
//     channels are exactly one scale block);
//   * conv2, then the epilogue of conv_c64_mx_kernel: 16-row transposing scratch, residual, out_scale, optional fp32 accumulate-in, row mask; outputs: the plane
//     set of leaky_relu(result, mxo_slope) and / or fp32 rows.
// Same products, same accumulation order, same quantisers as the two layer-wise launches: bit-identical outputs (tests/test_gpu_ops.py).
#pragma once

struct Pair64MxGeom {
    static constexpr int K = 3, C = 64, NB = 32, H2 = 1, GR = 128, BMO = GR - 2 * H2, KG = 2, KP = 4;
    static constexpr int XR = 144, TR = 136, EPITCH = NB * 4 + 16, MAXDIL = (XR - GR) / 2;
    static constexpr int WHB = K * NB * 64;            // one 32-channel K-chunk of one output-channel half of one conv's fp16 weights
    static constexpr int WQB = KP * NB * 32;           // one fp4 code plane of one half
    static constexpr int WSB = KP * NB * 2;            // its scale bytes
    // weights: [conv][half][chunk] fp16, [conv][half][plane] codes, [conv][half][plane] scales
    static constexpr int OFF_WH = 0, OFF_WQ = 8 * WHB, OFF_WS = OFF_WQ + 8 * WQB, OFF_XH = OFF_WS + 8 * WSB;
    static constexpr int XHC = XR * 64 + 64, THC = TR * 64 + 64;       // one K-chunk of a hi plane (+ 64 B: the two chunks land in different banks)
    static constexpr int OFF_XQ = OFF_XH + 2 * XHC, OFF_XS = OFF_XQ + 2 * XR * 32, OFF_TH = OFF_XS + 2 * XR * 4;
    static constexpr int OFF_TQ = OFF_TH + 2 * THC, OFF_TS = OFF_TQ + 2 * TR * 32, OFF_ES = OFF_TS + 2 * TR * 4;
    static constexpr int TOTAL = OFF_ES + 8 * 16 * EPITCH;
    static_assert(OFF_XH % 16 == 0 && OFF_XQ % 16 == 0 && OFF_TH % 16 == 0 && OFF_TQ % 16 == 0 && OFF_ES % 16 == 0 && TOTAL <= 160 * 1024 && GR + K - 1 <= TR,
                  "LDS plan");
};

// ACC: fp32 accumulate-in (epi.acc32, may alias epi.out32: the running MRF sum)
template <bool ACC>
__global__ __launch_bounds__(512, 1) void resblock_pair_c64_mx_kernel(const ResPairParams p) {
    using G = Pair64MxGeom;
    constexpr int K = G::K, NB = G::NB, H2 = G::H2, BMO = G::BMO, KG = G::KG, XR = G::XR, TR = G::TR, EPITCH = G::EPITCH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave & 3, ch = wave >> 2;           // row group (32 rows), output-channel half (also: which 32-channel chunk of xt this wave writes)
    char* const Xh = smem + G::OFF_XH;                 // [chunk][row][64 B] swizzled
    char* const Xq = smem + G::OFF_XQ;                 // [plane][row][32 B]
    char* const Xsc = smem + G::OFF_XS;                // [plane][row][4] (two bytes used)
    char* const Th = smem + G::OFF_TH;
    char* const Tq = smem + G::OFF_TQ;
    char* const Tsc = smem + G::OFF_TS;
    char* const es = smem + G::OFF_ES + wave * 16 * EPITCH;
    const int fr = lane & 15, fq = lane >> 4;
    const int dil = p.dil, h1 = H2 * dil;
    const ConvGemmParams& e = p.epi;
    const int ntiles = (p.M + BMO - 1) / BMO;

    // ---- both convs' weights -> LDS, once per block
    {
        for (int c = tid; c < 2 * 2 * 2 * K * NB * 4; c += 512) {            // [conv][half][chunk][tap * 32 + co][4 parts]
            const int part = c & 3, row = (c >> 2) % (K * NB), rest = (c >> 2) / (K * NB), chk = rest & 1, hf = (rest >> 1) & 1, cv = rest >> 2;
            const int tap = row >> 5, co = row & 31;
            const char* w16 = reinterpret_cast<const char*>(cv ? p.w2 : p.w1);
            const long off = ((long)((hf * NB + co) * K + tap) * 64 + chk * 32) * 2 + part * 16;
            *reinterpret_cast<uint4*>(smem + G::OFF_WH + ((cv * 2 + hf) * 2 + chk) * G::WHB + swz(row, part)) = *reinterpret_cast<const uint4*>(w16 + off);
        }
        for (int c = tid; c < 2 * 2 * 2 * G::WQB / 16; c += 512) {            // [conv][half][plane] x WQB / 16 units; host: [plane][half][KP][32][32 B]
            const int u = c % (G::WQB / 16), rest = c / (G::WQB / 16), pl = rest & 1, hf = (rest >> 1) & 1, cv = rest >> 2;
            const char* wm = reinterpret_cast<const char*>(cv ? p.w2_mx : p.w1_mx);
            *reinterpret_cast<uint4*>(smem + G::OFF_WQ + ((cv * 2 + hf) * 2 + pl) * G::WQB + u * 16) =
                *reinterpret_cast<const uint4*>(wm + (size_t)(pl * 2 + hf) * G::WQB + (size_t)u * 16);
        }
        for (int c = tid; c < 2 * 2 * 2 * G::WSB / 16; c += 512) {
            const int u = c % (G::WSB / 16), rest = c / (G::WSB / 16), pl = rest & 1, hf = (rest >> 1) & 1, cv = rest >> 2;
            const char* wm = reinterpret_cast<const char*>(cv ? p.w2_mx : p.w1_mx);
            *reinterpret_cast<uint4*>(smem + G::OFF_WS + ((cv * 2 + hf) * 2 + pl) * G::WSB + u * 16) =
                *reinterpret_cast<const uint4*>(wm + (size_t)4 * G::WQB + (size_t)(pl * 2 + hf) * G::WSB + (size_t)u * 16);
        }
    }
    f32x2 b1v[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 2; ++q) b1v[a][q] = f32x2{p.b1[ch * NB + a * 16 + 4 * fq + 2 * q], p.b1[ch * NB + a * 16 + 4 * fq + 2 * q + 1]};
    const int er = lane >> 2, eg = lane & 3, n0 = ch * NB, eco = n0 + eg * 8;          // coalesced side of the epilogue: 4 lanes per row, 16 rows per instruction
    const unsigned frbit = 1u << fr, erbit = 1u << er;
    f32x2 b2v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) b2v[q] = e.bias ? f32x2{e.bias[eco + 2 * q], e.bias[eco + 2 * q + 1]} : f32x2{0.f, 0.f};
    const f32x2 out_scale2 = f32x2{e.out_scale, e.out_scale};
    const f32x2 slope01 = f32x2{0.1f, 0.1f};
    const bool scaled = e.out_scale != 1.0f;
    const bool has_planes = e.mxo_h != nullptr;
    const bool part_out = e.mxo_partial != 0;          // the running MRF sum as a PARTIAL plane set (hi plane + remainder codes + their scales): no hi codes / hi scales
    const f32x2 mxo_slope2 = f32x2{e.mxo_slope, e.mxo_slope};
    const bool mxo_act = e.mxo_slope != 1.0f;
    const f32x2 res_inv2 = f32x2{e.res_inv_slope, e.res_inv_slope};
    float* const o32 = e.out32;
    const uint8_t* vptr = e.row_valid ? e.row_valid : g_row_always_valid;
    const int vshift = e.row_valid ? e.valid_shift : 31;
#define EV_P64_VROW(TILE) ((TILE) * BMO - H2 + rg * 32 + lane)
#define EV_P64_VLOAD(TILE, DST) { const int g_ = EV_P64_VROW(TILE); DST = vptr[min(max(g_, 0), p.M - 1) >> vshift]; }
#define EV_P64_VMASK(TILE, SRC) __builtin_amdgcn_ballot_w64((SRC) != 0 && EV_P64_VROW(TILE) >= 0 && EV_P64_VROW(TILE) < p.M)

    // ---- slab staging (plain copies of the producer's planes; rows beyond the convs' span re-read its last row, units beyond the slab duplicate its last unit):
    // hi plane 3 units per thread (u = tid + 512 i -> row u >> 3, part u & 7), code planes 2 (u -> plane u / (2 XR), row (u % (2 XR)) >> 1, half u & 1), scales 1
    u32x4 xh[3], xc[2], xs4;
    const int last_row = G::GR - 1 + (K - 1) * dil;                   // last slab row conv1 reads
    int hrow[3], crow[2], hdst[3], cdst[2];
    unsigned hpart[3], cpart[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = min((tid >> 3) + 64 * i, XR - 1);
        hrow[i] = min(r, last_row);
        hpart[i] = (tid & 7) * 16u;
        hdst[i] = ((tid >> 2) & 1) * G::XHC + swz(r, tid & 3);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = min(tid + 512 * j, 4 * XR - 1);
        const int pl = u / (2 * XR), row = (u % (2 * XR)) >> 1, hf = u & 1;
        crow[j] = min(row, last_row);
        cpart[j] = hf * 16u + (pl ? 0x80000000u : 0u);          // (bit 31: the remainder's code plane)
        cdst[j] = pl * XR * 32 + row * 32 + (hf << 4);
    }
    const int su = min(tid, 8 * XR / 16 - 1);
    const int spl = su / (XR / 4), srow = (su % (XR / 4)) * 4;
    const char* const hbase = reinterpret_cast<const char*>(p.x);
    const char* const cbase0 = reinterpret_cast<const char*>(e.mx_x4[0]);
    const char* const cbase1 = reinterpret_cast<const char*>(e.mx_x4[1]);
    const char* const sbase = reinterpret_cast<const char*>(spl ? e.mx_xs[1] : e.mx_xs[0]);
    // (tiles are 126 rows, not a divisor of M: the last tile's slab may reach past the planes' 64 slack rows -- such rows are clamped to the last readable
    // one; they only feed output rows >= M, which are not stored)
    const long row_max = (long)p.M + 60;
#define EV_P64_GLOAD(TILE)                                                                                 \
    {                                                                                                      \
        const long r00_ = (long)(TILE) * BMO - H2 - h1;                                                    \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) xh[i] = *reinterpret_cast<const u32x4*>(hbase + min(r00_ + hrow[i], row_max) * 128 + hpart[i]); \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
            xc[j] = *reinterpret_cast<const u32x4*>(((cpart[j] >> 31) ? cbase1 : cbase0) + min(r00_ + crow[j], row_max) * 32 + (cpart[j] & 0x7fffffffu)); \
        xs4 = *reinterpret_cast<const u32x4*>(sbase + min(r00_ + srow, row_max) * 4);                      \
    }
#define EV_P64_SSTORE()                                                                                    \
    {                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(Xh + hdst[i]) = xh[i];     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) *reinterpret_cast<u32x4*>(Xq + cdst[j]) = xc[j];     \
        *reinterpret_cast<u32x4*>(Xsc + spl * XR * 4 + srow * 4) = xs4;                                    \
    }
    // one conv of the pair on the wave's 32 rows x 32 output channels: fp16 hi x hi tap by tap (two 32-channel chunks each), then the two fp4 cross terms,
    // two taps per MFMA -- the instruction order of conv_c64_mx_kernel.  XH / XQ / XS: the operand's planes (chunk stride HC, NR rows), CONV: 0 / 1
#define EV_P64_CONV(XH, HC, XQ, XS, NR, CONV, DIL)                                                         \
    {                                                                                                      \
        const char* const Wh_ = smem + G::OFF_WH + ((CONV) * 2 + ch) * 2 * G::WHB;                         \
        const char* const Wq_ = smem + G::OFF_WQ + ((CONV) * 2 + ch) * 2 * G::WQB;                         \
        const char* const Ws_ = smem + G::OFF_WS + ((CONV) * 2 + ch) * 2 * G::WSB;                         \
        _Pragma("unroll") for (int t = 0; t < K; ++t) {                                                    \
            const int r0 = wrow0 + t * (DIL);                                                              \
            const int xo = r0 * 64 + ((fq ^ ((r0 >> 1) & 3)) << 4);                                        \
            _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_) {                                             \
                uint4 wf_[2];                                                                              \
                _Pragma("unroll") for (int a = 0; a < 2; ++a) wf_[a] = *reinterpret_cast<const uint4*>(Wh_ + c_ * G::WHB + swz(t * 32 + a * 16 + fr, fq)); \
                _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                            \
                    uint4 xf_ = *reinterpret_cast<const uint4*>((XH) + c_ * (HC) + xo + b * 16 * 64);      \
                    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                          \
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<half8*>(&wf_[a]), *reinterpret_cast<half8*>(&xf_), acc[a][b], 0, 0, 0); \
                }                                                                                          \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < KG; ++g) {                                                   \
            const int tw = 2 * g + (fq >> 1), hf = fq & 1;                                                 \
            const int rq = wrow0 + min(tw, K - 1) * (DIL);                                                 \
            _Pragma("unroll") for (int pl = 0; pl < 2; ++pl) {                                             \
                uint4 wq_[2], xq_[2];                                                                      \
                int ws_[2], xs_[2];                                                                        \
                _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                            \
                    const int wr = tw * 32 + a * 16 + fr;                                                  \
                    wq_[a] = *reinterpret_cast<const uint4*>(Wq_ + pl * G::WQB + wr * 32 + (hf << 4));     \
                    ws_[a] = *reinterpret_cast<const uint8_t*>(Ws_ + pl * G::WSB + wr * 2 + hf);           \
                }                                                                                          \
                _Pragma("unroll") for (int b = 0; b < 2; ++b) {                                            \
                    const int rr = rq + b * 16;                                                            \
                    xq_[b] = *reinterpret_cast<const uint4*>((XQ) + pl * (NR) * 32 + rr * 32 + (hf << 4)); \
                    xs_[b] = *reinterpret_cast<const uint8_t*>((XS) + pl * (NR) * 4 + rr * 4 + hf);        \
                }                                                                                          \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                              \
                    _Pragma("unroll") for (int a = 0; a < 2; ++a)                                          \
                        mfma_mx_inplace(acc[a][b], wq_[a], xq_[b], ws_[a], xs_[b]);                          \
            }                                                                                              \
        }                                                                                                  \
        mfma_asm_fence(acc);                                                                               \
    }
#define EV_P64_BARRIER()                                         \
    {                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);                       \
        __builtin_amdgcn_s_barrier();                            \
        __builtin_amdgcn_sched_barrier(0);                       \
    }

    int tile = blockIdx.x;                    // grid <= ntiles
    unsigned long long vmask;
    {
        uint8_t vb;
        EV_P64_GLOAD(tile)
        EV_P64_VLOAD(tile, vb)
        EV_P64_SSTORE()
        vmask = EV_P64_VMASK(tile, vb);
    }
    __syncthreads();
    const int wrow0 = rg * 32 + fr;
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = min(tile + (int)gridDim.x, ntiles - 1);      // clamped: the last prefetch of a block is never used
        const int m0 = tile * BMO;
        const int t_end = min(m0 + BMO, p.M);
        // ---------------- memory requests of this tile, oldest first: accumulate-in rows, the next tile's slab and row-valid byte
        float4 accin[2][2];
        if constexpr (ACC) {
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int t = min(m0 + rg * 32 + it * 16 + er, t_end - 1);
                const float* ap = e.acc32 + (long)t * e.ldacc + eco;
                accin[it][0] = *reinterpret_cast<const float4*>(ap);
                accin[it][1] = *reinterpret_cast<const float4*>(ap + 4);
            }
        }
        uint8_t vb_next;
        EV_P64_GLOAD(next)
        EV_P64_VLOAD(next, vb_next)
        // the residual rows of this tile, lifted out of the slab before the next one replaces it: output row r2 is slab row r2 + H2 + h1
        u32x4 rph[2];
        unsigned rpc[2], rps[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int sr = rg * 32 + it * 16 + er + H2 + h1;          // (<= 127 + 1 + dil < XR)
            rph[it] = *reinterpret_cast<const u32x4*>(Xh + ch * G::XHC + swz(sr, eg));
            rpc[it] = *reinterpret_cast<const unsigned*>(Xq + XR * 32 + sr * 32 + ch * 16 + eg * 4);
            rps[it] = *reinterpret_cast<const uint8_t*>(Xsc + XR * 4 + sr * 4 + ch);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[2][2];
        // ---------------- conv1 (dilation d): 128 rows, global rows m0 - H2 + r1
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        EV_P64_CONV(Xh, G::XHC, Xq, Xsc, XR, 0, dil)
        // bias + leaky-relu + zero outside the utterance (conv2 must see the reference's zero padding) -> the xt plane set (this wave's 32 channels = one block)
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
            mh = max_xor16_raw(max_xor32_raw(mh));
            ml = max_xor16_raw(max_xor32_raw(ml));
            const unsigned bh = mx_scale_byte(mh), bl = mx_scale_byte(ml);
            const float sh = __uint_as_float(bh << 23), sl = __uint_as_float(bl << 23);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                uint2 w;
                w.x = *reinterpret_cast<unsigned*>(&hh[a][0]); w.y = *reinterpret_cast<unsigned*>(&hh[a][1]);
                const int co = a * 16 + 4 * fq;               // channel inside this wave's 32-channel chunk
                *reinterpret_cast<uint2*>(Th + ch * G::THC + swz(r1, co >> 3) + (co & 7) * 2) = w;
                unsigned cc = 0, cl = 0;
                cc = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cc, hf[a][0][0], hf[a][0][1], sh, 0);
                cc = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cc, hf[a][1][0], hf[a][1][1], sh, 1);
                cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[a][0][0], lf[a][0][1], sl, 0);
                cl = __builtin_amdgcn_cvt_scalef32_pk_fp4_f32(cl, lf[a][1][0], lf[a][1][1], sl, 1);
                *reinterpret_cast<unsigned short*>(Tq + r1 * 32 + ch * 16 + (co >> 1)) = (unsigned short)cc;
                *reinterpret_cast<unsigned short*>(Tq + TR * 32 + r1 * 32 + ch * 16 + (co >> 1)) = (unsigned short)cl;
            }
            if (fq == 0) { Tsc[r1 * 4 + ch] = (char)bh; Tsc[TR * 4 + r1 * 4 + ch] = (char)bl; }
        }
        EV_P64_BARRIER()          // every wave is done with the slab; xt is complete
        EV_P64_SSTORE()           // the next tile's slab replaces the current one (conv2 only reads xt)
        const unsigned long long vmask_next = EV_P64_VMASK(next, vb_next);
        // ---------------- conv2 (dilation 1): rows m0 + r2, reads xt rows r2 + t
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        EV_P64_CONV(Th, G::THC, Tq, Tsc, TR, 1, 1)
        // ---------------- epilogue (conv_c64_mx_kernel's): the two 16-row transposes first, then all the arithmetic, then the stores
        const unsigned outmask = (unsigned)(vmask >> H2);
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
            const bool valid = (outmask & (erbit << (it * 16))) != 0u;
            f32x2* v = vv[it];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] += b2v[q];
            {           // x = lrelu^-1(hi + code * scale) = min(a, a * inv)
                f32x2 rr[4];
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
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] += rr[q];
                if (scaled) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] *= out_scale2;
                }
            }
            if constexpr (ACC) {
                v[0] += f32x2{accin[it][0].x, accin[it][0].y}; v[1] += f32x2{accin[it][0].z, accin[it][0].w};
                v[2] += f32x2{accin[it][1].x, accin[it][1].y}; v[3] += f32x2{accin[it][1].z, accin[it][1].w};
            }
            if (outmask != 0xffffffffu) {
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
            const long t = m0 + rg * 32 + it * 16 + er;
            const bool rowok = t < t_end;
            const f32x2* v = vv[it];
            if (rowok && o32) {
                float* op = o32 + t * e.ldo + eco;
                *reinterpret_cast<float4*>(op) = make_float4(v[0][0], v[0][1], v[1][0], v[1][1]);
                *reinterpret_cast<float4*>(op + 4) = make_float4(v[2][0], v[2][1], v[3][0], v[3][1]);
            }
            if (has_planes && rowok) {
                *reinterpret_cast<uint4*>(reinterpret_cast<char*>(e.mxo_h) + (t * 64 + eco) * 2) = pho[it];
                if (!part_out) *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e.mxo_q4[0]) + t * 32 + ch * 16 + eg * 4) = pch[it];
                *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e.mxo_q4[1]) + t * 32 + ch * 16 + eg * 4) = pcl[it];
                if (eg == 0) {
                    if (!part_out) reinterpret_cast<uint8_t*>(e.mxo_qs[0])[t * 4 + ch] = (uint8_t)pbh[it];
                    reinterpret_cast<uint8_t*>(e.mxo_qs[1])[t * 4 + ch] = (uint8_t)pbl[it];
                }
            }
        }
        vmask = vmask_next;
        EV_P64_BARRIER()          // the new slab is complete; xt may be overwritten
    }
#undef EV_P64_BARRIER
#undef EV_P64_CONV
#undef EV_P64_SSTORE
#undef EV_P64_GLOAD
#undef EV_P64_VROW
#undef EV_P64_VLOAD
#undef EV_P64_VMASK
}

static hipError_t pair64_mx_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = hipFuncSetAttribute((const void*)resblock_pair_c64_mx_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, Pair64MxGeom::TOTAL); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)resblock_pair_c64_mx_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, Pair64MxGeom::TOTAL); if (r != hipSuccess) e = r;
    return e;
}
// x = the input plane set's fp16 hi plane [rows][64] (ldx == 64), epi.mx_x4 / mx_xs its code / scale planes ([rows][32 B] / [rows][4]); w1 / w2 fp16 hi
// parts [64][3][64], w1_mx / w2_mx = mxfp4.pack_c64_weight_planes; epi: bias (= b2), res_inv_slope (the residual IS the input plane set), out_scale, acc32
// (optional, may alias out32), row_valid, outputs out32 (ldo == 64) and / or the plane set mxo_* (mxo_logC == 6; mxo_partial: the MRF sum's partial plane set --
// hi plane, remainder codes and their scales of the raw scaled result, planes only).  0, or -1 for an unsupported call.
int launch_resblock_pair_c64_mx(const ResPairParams& p, hipStream_t s) {
    const ConvGemmParams& e = p.epi;
    if (p.k != 3 || p.ldx != 64 || p.dil < 1 || p.dil > Pair64MxGeom::MAXDIL || p.M <= 0 || !p.w1_mx || !p.w2_mx || !e.mx_x4[0] || !e.mx_x4[1] || !e.mx_xs[0] || !e.mx_xs[1] ||
        !(e.out32 || e.mxo_h) || e.out16 || e.add16_a || e.post_lrelu || e.seq_bias || e.out32_before_post || !(e.res_inv_slope >= 1.0f) ||
        (e.out32 && e.ldo != 64) || (e.acc32 && e.ldacc != 64) ||
        (e.mxo_h && !(e.mxo_logC == 6 && (e.mxo_partial || (e.mxo_q4[0] && e.mxo_qs[0])) && e.mxo_q4[1] && e.mxo_qs[1] && e.mxo_slope >= 0.f && e.mxo_slope <= 1.f)) ||
        (e.mxo_partial && !(e.mxo_h && e.mxo_slope == 1.0f && !e.out32)) || e.acc_h)
        return -1;
    const int n_cu = device_cus();
    const int ntiles = (p.M + Pair64MxGeom::BMO - 1) / Pair64MxGeom::BMO;
    const int grid = ntiles < n_cu ? ntiles : n_cu;
    if (e.acc32) hipLaunchKernelGGL((resblock_pair_c64_mx_kernel<true>), dim3(grid), dim3(512), Pair64MxGeom::TOTAL, s, p);
    else hipLaunchKernelGGL((resblock_pair_c64_mx_kernel<false>), dim3(grid), dim3(512), Pair64MxGeom::TOTAL, s, p);
    return 0;
}
