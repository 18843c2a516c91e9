// One-tap "MX" GEMM (nn.Linear: the mel decoder's QKV and attention-output projections, reference modules/encoder.py:72-109) -- the arithmetic of
// ev_gemm_mx.h (one fp16 MFMA on the hi parts + two block-scaled fp4 MFMAs for the cross terms), but a pipeline of its own: conv_gemm_mx_kernel stages
// the activation slab once per K-chunk and re-uses it for every tap, so its ring assumes >= 3 steps per slab; with one tap every step needs its own
// 256-row activation piece as well as its weight tile.  Round 3 ran these GEMMs as three fp16 MFMAs per product (187 / 118 TF/s algorithmic).
//
// 8 waves, 256 x 128 tile, each wave 64 x 64 outputs (the MX kernel's wave tile and epilogues), two blocks per CU.  A step = one 64-byte column block of
// the operands: 32 channels in the fp16 pass (K / 32 steps), 128 channels of fp4 codes in each cross-term pass (K / 128 steps each).  Three LDS stages of
// [A 256 rows x 64 B | W 128 rows x 64 B | activation scales 1 KB | weight scales 1 KB] (26 KB each), filled by LDS-DMA two steps ahead: every wave issues
// exactly four requests per step (two activation pieces, one weight piece, one scale piece -- a dummy into a trash line outside the fp4 passes and for the
// waves that carry no scales), so "step s has landed" is the static vmcnt(4); ONE barrier per step (all eight waves run the same phase: with one tap a
// step has 8 KB + 16 KB of requests for 16 MFMAs per wave, i.e. it is request-bound, and splitting the waves into a matrix and a load group buys nothing).
// The activation planes come from mx_planes_kernel (the producing LayerNorm / attention write fp32).
#pragma once

static constexpr int MX1_STAGE = 16384 + 8192 + 2048, MX1_NS = 3, MX1_TRASH = MX1_NS * MX1_STAGE;
static constexpr size_t MX1_LDS = (size_t)MX1_TRASH + 1024;
static_assert(MX1_LDS <= 80 * 1024 && 8 * 32 * (64 * 4 + 16) <= (int)MX1_LDS, "two blocks per CU; the epilogue scratch aliases the stages");

template <int EPI>
__global__ __launch_bounds__(512, 4) void gemm_mx1_kernel(const ConvGemmParams p) {
    constexpr int BM = 256, TC = 64, MT = 4, NT = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave & 3, wc = wave >> 2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    int bid = blockIdx.x;
    const int nblk = gridDim.x, nN = p.N >> 7;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    const int m0 = (bid / nN) * BM, nt = bid % nN, n0 = nt * 128;
    const int n16 = p.K >> 5, n4 = p.K >> 7, nstep = n16 + 2 * n4;
    const unsigned K2 = (unsigned)p.K >> 1;                // row pitch of a fp4 plane in bytes; the fp16 planes' is 4 x that

    const int prow = lane >> 2;
    const unsigned pp16 = (unsigned)(((lane & 3) ^ ((lane >> 3) & 3)) << 4);          // source-side swizzle of the 16-byte parts
    const unsigned arK0 = __umul24((unsigned)(wave * 16 + prow), K2), arK1 = __umul24((unsigned)((wave + 8) * 16 + prow), K2);
    const unsigned wrK = arK0;                                                          // weight piece: rows 16 w .. 16 w + 15 of the 128-row tile
    const char* const wmx = reinterpret_cast<const char*>(p.W_mx);
    const size_t nw4 = (size_t)p.N * K2, nws = (size_t)nN * n4 * 128;
    const char* const xb16 = uniform_ptr(reinterpret_cast<const char*>(p.A) + (long)m0 * (long)(4 * K2));
    const char* const xb4h = uniform_ptr(reinterpret_cast<const char*>(p.mx_x4[0]) + (long)m0 * (long)K2);
    const char* const xb4l = uniform_ptr(reinterpret_cast<const char*>(p.mx_x4[1]) + (long)m0 * (long)K2);
    const char* const wb16 = uniform_ptr(reinterpret_cast<const char*>(p.W) + (long)n0 * (long)(4 * K2));
    const char* const wb4l = uniform_ptr(wmx + (long)n0 * (long)K2);
    const char* const wb4h = uniform_ptr(wmx + nw4 + (long)n0 * (long)K2);
    const char* const sxh = uniform_ptr(reinterpret_cast<const char*>(p.mx_xs[0]) + (long)m0 * 4);
    const char* const sxl = uniform_ptr(reinterpret_cast<const char*>(p.mx_xs[1]) + (long)m0 * 4);
    const char* const swl = uniform_ptr(wmx + 2 * nw4 + (size_t)nt * n4 * 128);
    const char* const swh = uniform_ptr(wmx + 2 * nw4 + nws + (size_t)nt * n4 * 128);
    const unsigned xs_stride = p.mx_xs_stride;
    // scale piece: wave 0 carries the 256 x 4 activation-scale bytes of the chunk, wave 1 the 128 weight-scale bytes (lanes beyond them re-read the last
    // 16 bytes), everybody else and every fp16 step a dummy that lands in the trash line
    const unsigned scv = wave == 0 ? (unsigned)lane * 16u : min((unsigned)lane * 16u, 112u);

    const unsigned sdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);           // A piece 0 of the wave inside a stage; piece 1: + 8 KB; W: + 16 KB
    // all four requests of step S into stage ST (S, ST wave-uniform).  No `if` around a request: the operands are selected.
#define EV_MX1_ISSUE(S, ST)                                                                                              \
    {                                                                                                                    \
        const int s_ = (S);                                                                                              \
        const bool f16_ = s_ < n16, sec_ = s_ >= n16 + n4;                                                               \
        const unsigned j_ = (unsigned)(f16_ ? s_ : (sec_ ? s_ - n16 - n4 : s_ - n16));                                   \
        const char* const ab_ = uniform_ptr((f16_ ? xb16 : (sec_ ? xb4l : xb4h)) + j_ * 64u);                            \
        const char* const wb_ = uniform_ptr((f16_ ? wb16 : (sec_ ? wb4h : wb4l)) + j_ * 64u);                            \
        const unsigned sh_ = f16_ ? 2u : 0u;                                                                             \
        const unsigned st_ = sdst + (unsigned)(ST) * MX1_STAGE;                                                          \
        glds16(ab_, (arK0 << sh_) + pp16, st_);                                                                          \
        glds16(ab_, (arK1 << sh_) + pp16, st_ + 8192);                                                                   \
        glds16(wb_, (wrK << sh_) + pp16, st_ + 16384);                                                                   \
        const bool carry_ = !f16_ && wave < 2;                                                                           \
        const char* const sb_ = uniform_ptr(wave == 0 ? (sec_ ? sxl : sxh) + j_ * xs_stride : (sec_ ? swh : swl) + j_ * 128u); \
        glds16(carry_ ? sb_ : xb16, carry_ ? scv : 0u,                                                                   \
               carry_ ? lds0 + (unsigned)(ST) * MX1_STAGE + 24576 + (unsigned)wave * 1024 : lds0 + MX1_TRASH);           \
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int arow = wt * 64 + fr;
    const int aoff = arow * 64 + ((fq ^ ((arow >> 1) & 3)) << 4);         // + b * 1024
    const int woff = 16384 + swz(wc * TC + fr, fq);                       // + a * 1024
    const int xs_off = 24576 + arow * 4 + fq;                             // + b * 64
    const int ws_off = 24576 + 1024 + wc * TC + fr;                       // + a * 16

    bool tile_live = true;          // (a tile without a valid row: straight to the epilogue's masked zeros, see conv_gemm_mx_kernel)
    if (p.row_valid) {
        const uint8_t* vp = p.row_valid;
        const int r4 = m0 + lane * 4, vs = p.valid_shift;
        const unsigned any = vp[r4 >> vs] | vp[(r4 + 1) >> vs] | vp[(r4 + 2) >> vs] | vp[(r4 + 3) >> vs];
        tile_live = __builtin_amdgcn_ballot_w64(any != 0) != 0ull;
    }
    if (tile_live) {
        EV_MX1_ISSUE(0, 0)
        EV_MX1_ISSUE(1, 1)          // (nstep >= 6: K % 128 == 0)
        int st = 0;                 // stage of the current step
#define EV_MX1_STEP(MX)                                                                                                  \
        {                                                                                                                \
            if (s + 1 < nstep) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            __builtin_amdgcn_s_barrier();          /* step s has landed for every wave; everybody is done with the stage of step s - 1 */ \
            __builtin_amdgcn_sched_barrier(0);                                                                           \
            const int st2 = st == 0 ? 2 : st - 1;          /* (s + 2) % 3 */                                             \
            if (s + 2 < nstep) EV_MX1_ISSUE(s + 2, st2)                                                                  \
            const char* const sb_ = smem + st * MX1_STAGE;                                                               \
            uint4 xf[MT], wf[NT];                                                                                        \
            int xsc[MT], wsc[NT];                                                                                        \
            _Pragma("unroll") for (int a = 0; a < NT; ++a) wf[a] = *reinterpret_cast<const uint4*>(sb_ + woff + a * 1024); \
            _Pragma("unroll") for (int b = 0; b < MT; ++b) xf[b] = *reinterpret_cast<const uint4*>(sb_ + aoff + b * 1024); \
            if constexpr (MX) {                                                                                          \
                _Pragma("unroll") for (int a = 0; a < NT; ++a) wsc[a] = *reinterpret_cast<const uint8_t*>(sb_ + ws_off + a * 16); \
                _Pragma("unroll") for (int b = 0; b < MT; ++b) xsc[b] = *reinterpret_cast<const uint8_t*>(sb_ + xs_off + b * 64); \
            }                                                                                                            \
            _Pragma("unroll") for (int a = 0; a < NT; ++a)                                                               \
                _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                         \
                    if constexpr (MX) mfma_mx_inplace(acc[a][b], wf[a], xf[b], wsc[a], xsc[b]);                          \
                    else mfma_inplace(acc[a][b], *reinterpret_cast<half8*>(&wf[a]), *reinterpret_cast<half8*>(&xf[b])); \
                }                                                                                                        \
            st = st == 2 ? 0 : st + 1;                                                                                   \
        }
        int s = 0;
        for (; s < n16; ++s) EV_MX1_STEP(false)
        for (; s < nstep; ++s) EV_MX1_STEP(true)
#undef EV_MX1_STEP
        __builtin_amdgcn_s_barrier();          // every wave is out of the last stage: the epilogue's scratch may overwrite it
        mfma_asm_fence(acc);
    }
#undef EV_MX1_ISSUE
    EV_TRACE_EPI_DUMMY
    gemm_epilogue_fast<MT, NT, EPI>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * 64, n0 + wc * TC EV_TRACE_EPI_ARGS);
}

// epilogue variants of the one-tap kernel: plain fp32 output (QKV), fp32 residual + fp32 output (the attention's output projection, in place)
static int mx1_epi_variant(const ConvGemmParams& p) {
    if (p.seq_bias || p.add16_a || p.out16 || p.out32_before_post || p.post_lrelu || p.act != ACT_NONE || p.acc32 || p.acc_h || p.mxo_partial || p.mxo_h || !p.out32) return -1;
    if (!p.res) return EPI_O32;
    return p.res_dtype == DT_F32 ? (EPI_RES32 | EPI_O32 | EPI_LEAN) : -1;
}
static bool mx1_shape_ok(const ConvGemmParams& p) {
    return p.W_mx && p.taps == 1 && p.N % 128 == 0 && p.K % 128 == 0 && p.M % 256 == 0 && p.lda == p.K && mx1_epi_variant(p) >= 0;
}
static hipError_t mx1_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = hipFuncSetAttribute((const void*)gemm_mx1_kernel<EPI_O32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MX1_LDS); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)gemm_mx1_kernel<EPI_RES32 | EPI_O32 | EPI_LEAN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MX1_LDS); if (r != hipSuccess) e = r;
    return e;
}
static void launch_mx1_kernel(const ConvGemmParams& p, hipStream_t s) {
    const int grid = (p.M / 256) * (p.N / 128);
    if (mx1_epi_variant(p) == EPI_O32) hipLaunchKernelGGL((gemm_mx1_kernel<EPI_O32>), dim3(grid), dim3(512), MX1_LDS, s, p);
    else hipLaunchKernelGGL((gemm_mx1_kernel<EPI_RES32 | EPI_O32 | EPI_LEAN>), dim3(grid), dim3(512), MX1_LDS, s, p);
}
