// "MX" conv-GEMM: fp32-class activations, every product evaluated as
//
//        x.w  =  xh.wh                         one fp16 MFMA            (v_mfma_f32_16x16x32_f16)
//              + Q(wl).Q(xh) + Q(wh).Q(xl)     two block-scaled fp4 MFMAs (v_mfma_scale_f32_16x16x128_f8f6f4: K = 128 per instruction,
//                                              4x the fp16 rate, one E8M0 scale per lane = per 32 consecutive K elements)
//
// with xh = fp16(x), xl = x - xh (and the same for w).  The cross terms are 2^-11 of the result, so fp4's ~2 significant bits turn
// fp16's 3e-4 per-product rounding into ~3e-5: the generator's waveform error drops from 2.3e-3 (fp16 operands) to 3.5e-4
// (tools/precision_study_mx.py), inside the 1e-3 contract, for 1.5 MFMA-units instead of the split-precision mode's 3
// (tools/mfma_ubench.hip: 64 f16 + 32 MX MFMAs per K = 128 sustain 1.16-1.29 PF/s algorithmic on random operands, register-resident,
// against 1.96 PF/s for the fp16 MFMAs alone and 0.65 for three of them).
//
// Included by ev_gemm.hip (inside namespace ev): re-uses its LDS-DMA helpers, swizzle and epilogues.
//
// Operand planes.  A 128-channel chunk of a row is 64 bytes of fp4 codes -- byte-for-byte the geometry of a 32-channel fp16 chunk -- so
// the phased kernel's whole staging machinery (384-row slabs of 64-byte rows, source-side XOR swizzle, weight ring of four 8-KB
// tiles, counted vmcnt) carries over unchanged, and the kernel is the SAME pipeline run over three operand pairs back to back:
//        pass 0:  A = wh (fp16)   B = xh (fp16)    K / 32 chunks, f16 MFMA
//        pass 1:  A = Q(wl) fp4   B = Q(xh) fp4    K / 128 chunks, MX MFMA
//        pass 2:  A = Q(wh) fp4   B = Q(xl) fp4    K / 128 chunks, MX MFMA
// as one continuous step sequence (no drain between passes), all three into the same fp32 accumulators.
// Scales: activations one byte per (row, 32 channels), stored chunk-major [K/128][rows][4]; weights one byte per (output channel,
// tap, 128 channels), stored [N/128][K/128][taps][128].  Per chunk they are 1.3 KB + taps x 128 B and travel in the third slab piece
// of waves 4-7, which in the fp16 pass only re-reads the slab's last row (rows 320..383 of a slab buffer are never read by a
// fragment): same piece count per wave in every pass, so every vmcnt immediate stays what it was.
//
// Activation planes come from the PRODUCER's epilogue (EPI_MXP / mx_emit_planes in ev_gemm.hip: the consumer's leaky-relu, fp16 hi plane,
// fp4 codes of hi and of the fp32 remainder, scales -- 3.06 bytes per element instead of fp32's 4, and no tensor pass of its own), or,
// for an fp32 input that some other kernel wrote (conv_pre's output, the op tests), from mx_planes_kernel.
#pragma once

static constexpr int MX_SLACK = 64;          // readable rows in front of / behind every plane (the conv halo of the first / last tile)

__host__ __device__ inline size_t mx_align256(size_t v) { return (v + 255) & ~(size_t)255; }
struct MxPlaneLayout { size_t h, x4h, x4l, sh, sl, total; size_t rows; };
static MxPlaneLayout mx_layout(int M, int K) {
    MxPlaneLayout L;
    L.rows = (size_t)M + 2 * MX_SLACK;
    size_t o = 0;
    L.h = o; o = mx_align256(o + L.rows * K * 2);
    L.x4h = o; o = mx_align256(o + L.rows * (K / 2));
    L.x4l = o; o = mx_align256(o + L.rows * (K / 2));
    L.sh = o; o = mx_align256(o + (size_t)(K / 128) * L.rows * 4);
    L.sl = o; o = mx_align256(o + (size_t)(K / 128) * L.rows * 4);
    L.total = o;
    return L;
}
size_t mx_scratch_bytes(int M, int K) { return mx_layout(M, K).total; }

// ---- fp32 [M][K] -> planes.  One thread = 8 consecutive channels of a row (one 16-byte store of the fp16 plane, one dword of each
// code plane); the four threads of a 32-channel block agree on the block maxima through two xor-shuffles.
__global__ __launch_bounds__(256) void mx_planes_kernel(const float* __restrict__ A, int lda, int M, int K, int pro, float slope,
                                                        __half* __restrict__ H, uint8_t* __restrict__ x4h, uint8_t* __restrict__ x4l,
                                                        uint8_t* __restrict__ sh, uint8_t* __restrict__ sl, unsigned s_stride) {
    const int tpr = K >> 3;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = gid / tpr;
    const int g = (int)(gid - row * tpr);
    if (row >= M) return;
    const float4 a0 = *reinterpret_cast<const float4*>(A + row * lda + g * 8), a1 = *reinterpret_cast<const float4*>(A + row * lda + g * 8 + 4);
    float v[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    if (pro) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], v[e] * slope);
    }
    half8 hv;
    float hf[8], lf[8], mh = 0.f, ml = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 h = (_Float16)v[e];
        hv[e] = h; hf[e] = (float)h; lf[e] = v[e] - hf[e];
        mh = fmaxf(mh, fabsf(hf[e])); ml = fmaxf(ml, fabsf(lf[e]));
    }
    mh = fmaxf(mh, __shfl_xor(mh, 1)); mh = fmaxf(mh, __shfl_xor(mh, 2));
    ml = fmaxf(ml, __shfl_xor(ml, 1)); ml = fmaxf(ml, __shfl_xor(ml, 2));
    const unsigned bh = mx_scale_byte(mh), bl = mx_scale_byte(ml);
    const float ih = __uint_as_float((254u - bh) << 23), il = __uint_as_float((254u - bl) << 23);
    unsigned ch = 0, cl = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) { ch |= mx_fp4_code(hf[e] * ih) << (4 * e); cl |= mx_fp4_code(lf[e] * il) << (4 * e); }
    *reinterpret_cast<half8*>(reinterpret_cast<_Float16*>(H) + row * K + g * 8) = hv;
    *reinterpret_cast<unsigned*>(x4h + row * (K >> 1) + g * 4) = ch;
    *reinterpret_cast<unsigned*>(x4l + row * (K >> 1) + g * 4) = cl;
    if ((g & 3) == 0) {
        const long so = (long)(g >> 4) * s_stride + row * 4 + ((g >> 2) & 3);
        sh[so] = (uint8_t)bh; sl[so] = (uint8_t)bl;
    }
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// A = 16 output channels x 128 K (fp4: 16 bytes per lane, lane = (row l & 15, 32-element K block l >> 4)), B likewise for 16 time steps;
// byte 0 of sa / sb is the lane's E8M0 block scale.  In place, as mfma_inplace.
__device__ __forceinline__ void mfma_mx_inplace(f32x4& c, const uint4& a, const uint4& b, int sa, int sb) {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                 : "+v"(c) : "v"(*reinterpret_cast<const u32x4*>(&a)), "v"(*reinterpret_cast<const u32x4*>(&b)), "v"(sa), "v"(sb));
}


// LDS of a slab buffer: rows 0..319 of 64 B (fragments), [20 KB, 22 KB): activation scales (4 B per slab row, from row (m0 - c d) & ~3),
// [22 KB, 24 KB): weight scales [tap][128]
static constexpr int MX_XS_OFF = 20480, MX_WS_OFF = 22528;

// One 256-row x 128-channel tile of a launch.  bid_in / nblk: the block's index among the launch's (or, in a grouped launch, the problem's) blocks and their
// number; a grouped launch pads every problem to a multiple of eight blocks so that bid_in & 7 is the XCD the hardware's round-robin gave the block.
// ZT (TAPS == 3 only): the tap whose weights are all zero for this WAVE's 64 output channels -- a transposed conv in polyphase form, conv_gemm_mx_up_kernel -- or -1.
// Its steps keep their requests, waits and barriers and issue no fragment reads and no matrix instructions.  A compile-time parameter: with a run-time test around
// the MFMA block hipcc gave the accumulators different registers on the two paths (430-490 spilled registers in every k = 3 instantiation).
template <int TAPS, int EPI, int ZT = -1>
__device__ __forceinline__ void conv_gemm_mx_tile(const ConvGemmParams& p, const int bid_in, const int nblk) {
    constexpr int BM = PH_BM, XBUF = PH_XBUF, WBUF = PH_WBUF, U = TAPS, TC = 64, MT = 4, NT = 4;
    static_assert(U >= 3 && TAPS * 128 <= 2048 && PH_SLABR >= BM + MAX_SPAN + 64, "pipeline depth / scale pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Xs = smem;                 // [2][SLABR][64]
    char* const Ws = smem + 2 * XBUF;      // [4][8 KB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave & 3, wc = wave >> 2;          // wc is also the phase group: waves w and w + 4 share a SIMD
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    // EV_MXT (tuning builds: build.py --variant mxt EV_MXT; tools/bench_mxgemm.py --timeline): wave 0 of every block records the chip-wide 100-MHz clock
    // (s_memrealtime) at its start, at the start of its epilogue and at its end, + its hardware id, to ((unsigned long long*)p.row_seq)[blockIdx.x * 4 + 0..3]
#ifdef EV_MXT
    const unsigned long long mxt0 = __builtin_amdgcn_s_memrealtime();
#endif

    // (Round 4 tried a persistent tile loop here, twice.  Second form: the next tile's first six requests and its row_valid look-up issued between a tile's last
    // barrier and its epilogue, 16-row swizzled epilogue scratch beside them: bit-identical, 3-5 % SLOWER in every shape, with or without the early requests,
    // with or without a staggered start of a CU's two blocks (profiles/r4_e_mx_persistent_prefetch_ab.txt).  A wave's vmcnt is one in-order FIFO of its loads AND
    // stores: a wave that has just issued a tile's stores cannot wait for any newer request without draining them, while a fresh block starts with an empty FIFO
    // behind the retired block's draining stores.  The dispatcher's back-filling IS the overlap.  First form:)
    // (Round 4 tried a persistent tile loop here -- min(tiles, 2 x CUs) blocks walking the tile list with stride gridDim.x, bit-identical results --
    // to save the ~5 us per residency round that a tile costs outside its K loop and epilogue (profiles/r3_k_mx_gemm_ablation.txt).  Measured
    // in one process against one block per tile (profiles/r4_b_mx_persistent_ab.txt): 0-5 % SLOWER on every stage-0 / stage-1 shape.  The hardware's
    // own dispatch back-fills a CU the moment a block retires; a static stride cannot, and the loop needs a block barrier per tile because the
    // epilogue's transposing scratch aliases the staging buffers.  The fixed cost is pipeline fill latency, not block launch.)
    int bid = bid_in;
    const int nN = p.N >> 7;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    if (bid >= (p.M / BM) * nN) return;          // (padding blocks of a grouped launch; never true for a launch of its own)
    const int m0 = (bid / nN) * BM, nt = bid % nN, n0 = nt * 128;

    const int nkc16_ = p.K >> 5, nkc4 = p.K >> 7;
    const unsigned K2 = (unsigned)p.K >> 1;                // row / tap pitch of a fp4 plane in bytes; the fp16 planes' is 4 x that
    const unsigned wrp4 = K2 * TAPS;
#ifdef EV_MX_ABL
    // reserved0 bit 9 (abl 32): every block stages the activation rows of one of 16 tiles -- the slab requests hit L2 instead of HBM (results are garbage by design):
    // what a launch costs when the slab round trip is short
    const int row0 = (((p.reserved0 >> 4) & 32) ? (bid & 15) * BM : m0) - p.center * p.dil;
#else
    const int row0 = m0 - p.center * p.dil;                // first slab row (negative for the first tile: slack rows)
#endif
    const int row0a = row0 & ~3, soff = row0 - row0a;      // the scale run starts on a 16-byte boundary

    const int prow = lane >> 2;
    const unsigned pp16 = (unsigned)(((lane & 3) ^ ((lane >> 3) & 3)) << 4);
    const unsigned wrK = __umul24((unsigned)(wave * 16 + prow), wrp4);
    unsigned rsK[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = (wave + 8 * i) * 16 + prow, rs = min(r, BM + (TAPS - 1) * p.dil - 1);
        rsK[i] = __umul24((unsigned)rs, K2);
    }
    // lane offset of this wave's scale piece (waves 4, 5: activation scales; 6, 7: weight scales), clamped inside the run
    // (activation run: 4 bytes per slab row from row0a, rounded up to 16; nothing beyond it is requested: the planes have 64 slack rows)
    const int xs_run = ((BM + (TAPS - 1) * p.dil + soff) * 4 + 15) & ~15;
    unsigned scv;
    if (wave == 4) scv = min(lane * 16, xs_run - 16);
    else if (wave == 5) scv = min(1024 + lane * 16, xs_run - 16);
    else if (wave == 6) scv = min(lane * 16, TAPS * 128 - 16);
    else scv = min(1024 + lane * 16, TAPS * 128 - 16);

    const char* const wmx = reinterpret_cast<const char*>(p.W_mx);
    const size_t nw4 = (size_t)p.N * wrp4, nws = (size_t)nN * nkc4 * TAPS * 128;
    const char* const xb16 = uniform_ptr(reinterpret_cast<const char*>(p.A) + (long)row0 * (long)(4 * K2));
    const char* const xb4h = uniform_ptr(reinterpret_cast<const char*>(p.mx_x4[0]) + (long)row0 * (long)K2);
    const char* const xb4l = uniform_ptr(reinterpret_cast<const char*>(p.mx_x4[1]) + (long)row0 * (long)K2);
    const char* const wb16 = uniform_ptr(reinterpret_cast<const char*>(p.W) + (long)n0 * (long)(4 * wrp4));
    const char* const wb4l = uniform_ptr(wmx + (long)n0 * (long)wrp4);
    const char* const sxh = uniform_ptr(reinterpret_cast<const char*>(p.mx_xs[0]) + (long)row0a * 4);
    const char* const sxl = uniform_ptr(reinterpret_cast<const char*>(p.mx_xs[1]) + (long)row0a * 4);
    const char* const swl = uniform_ptr(wmx + 2 * nw4 + (size_t)nt * nkc4 * TAPS * 128);
    const unsigned xs_stride = p.mx_xs_stride;

    // Operand sources of a chunk are plain scalar arithmetic on these bases (two-way uniform selects at most: a lambda returning a struct
    // of pointers picked three ways became a table in SCRATCH, indexed inside the loop).  The second fp4 pass is the first one shifted by
    // constant byte distances.
    const long dx4 = (long)(xb4l - xb4h), dw4 = (long)nw4;
    const bool scw = wave >= 4;                                                            // this wave's third slab piece carries scales
    const char* const sc0 = uniform_ptr(wave < 6 ? sxh : swl);
    const long dsc = wave < 6 ? (long)(sxl - sxh) : (long)nws;
    const unsigned scstep = wave < 6 ? xs_stride : (unsigned)(TAPS * 128);

    const unsigned xdst = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024), wdst = xdst + 2 * XBUF;
    // weight tile of tap US (base WB, pitch shift SH) -> ring slot SLOT; slab piece I (base XB / scale run SC, next chunk) -> slab buffer BUF.
    // No `if` around an issue: an asm statement under a condition becomes a real branch inside the MFMA sequence; the operands are selected.
#define EV_MX_ISSUE_W(WB, SH, US, SLOT) glds16((WB) + (unsigned)(US) * (K2 << (SH)), (wrK << (SH)) + pp16, wdst + (unsigned)(SLOT) * WBUF);
#define EV_MX_ISSUE_X(XB, SC, SH, ISMX, BUF, I)                                                                      \
    {                                                                                                                \
        if constexpr ((I) == 2) {                                                                                    \
            const bool sc_ = (ISMX) && scw;                                                                          \
            glds16(sc_ ? (SC) : (XB), sc_ ? scv : (rsK[2] << (SH)) + pp16, xdst + (unsigned)(BUF) * XBUF + 2 * 8192); \
        } else glds16((XB), (rsK[I] << (SH)) + pp16, xdst + (unsigned)(BUF) * XBUF + (I) * 8192);                     \
    }

    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fq = lane >> 4;
    const int woff = swz(wc * TC + fr, fq);
    const int ws_off = MX_WS_OFF + wc * TC + fr;                       // + tap * 128 + a * 16
    const int xs_off = MX_XS_OFF + (wt * 64 + fr + soff) * 4 + fq;     // + (tap * dil + b * 16) * 4

    // A tile without a single valid row (the gap rows between utterances and the padding behind the last one are whole 256-row tiles at the generator's
    // upsampled rates: 64 of the 8256 tiles of a stage-1 launch at B = 32 x 1024 frames, exactly the ones that spill 16 full rounds of 512 resident
    // blocks into a 17th) needs no products: its outputs are the epilogue's masked zeros whatever the accumulators hold, so it goes straight there.
    // Every wave reads the tile's 256 validity bytes itself (4 rows per lane) -- wave-uniform and identical in all 8 waves, no barrier involved.
    bool tile_live = true;
    if (p.row_valid) {
        const uint8_t* vp = p.row_valid;
        const int r4 = m0 + lane * 4, vs = p.valid_shift;
        const unsigned any = vp[r4 >> vs] | vp[(r4 + 1) >> vs] | vp[(r4 + 2) >> vs] | vp[(r4 + 3) >> vs];
        tile_live = __builtin_amdgcn_ballot_w64(any != 0) != 0ull;
    }
    if (tile_live) {
    EV_MX_ISSUE_X(xb16, sc0, 2u, false, 0, 0)
    EV_MX_ISSUE_X(xb16, sc0, 2u, false, 0, 1)
    EV_MX_ISSUE_X(xb16, sc0, 2u, false, 0, 2)
    EV_MX_ISSUE_W(wb16, 2u, 0, 0)
    EV_MX_ISSUE_W(wb16, 2u, 1, 1)
    EV_MX_ISSUE_W(wb16, 2u, 2, 2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wc == 1) __builtin_amdgcn_s_barrier();

    // One (chunk, tap) step; see conv_gemm_phased_body for the schedule (requests inside the matrix phase, fragment reads retired by the
    // compiler's own lgkmcnt in front of the first MFMA, two barriers per step).  MX: true in the fp4 passes.  In scope: cwb (this chunk's
    // weight base), CSH (its pitch shift, static), nwb / nsh / nxb / nsc / nmx (the next chunk's), more, Xb, dil_, sbase, q.
#define EV_MX_STEP(MX, CSH, u)                                                                                          \
    {                                                                                                                   \
        const int s = sbase + (u);                                                                                      \
        const bool in_cur_static = ((u) + 3 < U);                                                                       \
        const bool in_cur = in_cur_static || !more;                                                                     \
        const int u3 = in_cur_static ? (u) + 3 : (more ? (u) + 3 - U : U - 1);                                          \
        const char* const wsel = in_cur_static ? cwb : (more ? nwb : cwb);                                              \
        const unsigned wsh = in_cur_static ? (unsigned)(CSH) : (more ? nsh : (unsigned)(CSH));                          \
        (void)in_cur;                                                                                                   \
        uint4 xf[MT], wf[NT];                                                                                           \
        int xsc[MT], wsc[NT];                                                                                           \
        if (ZT != (u)) {                                                                                      \
            const int row0_ = wt * 64 + fr + (u) * dil_;                                                                \
            const char* xp = Xb + row0_ * 64 + ((fq ^ ((row0_ >> 1) & 3)) << 4);                                        \
            const char* wp = Ws + (s & 3) * WBUF + woff;                                                                \
            _Pragma("unroll") for (int a = 0; a < NT; ++a) wf[a] = *reinterpret_cast<const uint4*>(wp + a * 1024);      \
            _Pragma("unroll") for (int b = 0; b < MT; ++b) xf[b] = *reinterpret_cast<const uint4*>(xp + b * 1024);      \
            if constexpr (MX) {                                                                                         \
                _Pragma("unroll") for (int a = 0; a < NT; ++a) wsc[a] = *reinterpret_cast<const uint8_t*>(Xb + ws_off + (u) * 128 + a * 16); \
                _Pragma("unroll") for (int b = 0; b < MT; ++b) xsc[b] = *reinterpret_cast<const uint8_t*>(Xb + xs_off + ((u) * dil_ + b * 16) * 4); \
            }                                                                                                           \
        }                                                                                                               \
        if ((u) == 1 && more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_barrier();                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_setprio(1);                                                                                  \
        if (ZT == (u)) {          /* polyphase transposed conv: this tap's weights are zeros -- requests only */ \
            EV_MX_ISSUE_W(wsel, wsh, u3, (s + 3) & 3)                                                                   \
            if ((u) == 0 && more) {                                                                                     \
                EV_MX_ISSUE_X(nxb, nsc, nsh, nmx, (q + 1) & 1, 0)                                                       \
                EV_MX_ISSUE_X(nxb, nsc, nsh, nmx, (q + 1) & 1, 1)                                                       \
                EV_MX_ISSUE_X(nxb, nsc, nsh, nmx, (q + 1) & 1, 2)                                                       \
            }                                                                                                           \
        } else {                                                                                                        \
        _Pragma("unroll") for (int a = 0; a < NT; ++a) {                                                                \
            _Pragma("unroll") for (int b = 0; b < MT; ++b) {                                                            \
                if constexpr (MX) mfma_mx_inplace(acc[a][b], wf[a], xf[b], wsc[a], xsc[b]);                             \
                else mfma_inplace(acc[a][b], *reinterpret_cast<half8*>(&wf[a]), *reinterpret_cast<half8*>(&xf[b]));     \
                const int idx = a * MT + b;                                                                             \
                if (idx == 2) { EV_MX_ISSUE_W(wsel, wsh, u3, (s + 3) & 3) }                                             \
                if ((u) == 0 && more) {                                                                                 \
                    if (idx == 4) { EV_MX_ISSUE_X(nxb, nsc, nsh, nmx, (q + 1) & 1, 0) }                                 \
                    if (idx == 5) { EV_MX_ISSUE_X(nxb, nsc, nsh, nmx, (q + 1) & 1, 1) }                                 \
                    if (idx == 6) { EV_MX_ISSUE_X(nxb, nsc, nsh, nmx, (q + 1) & 1, 2) }                                 \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
        }                                                                                                               \
        __builtin_amdgcn_s_setprio(0);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        __builtin_amdgcn_s_barrier();                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }

    // two loops, not one loop with an fp16 / fp4 branch: with both bodies behind a branch hipcc gave the accumulators different registers on the
    // two paths (64 moves per chunk and 80-150 spilled registers); back to back they simply flow from the first loop into the second
    int sbase = 0, q = 0;
    // EV_MX_ABL (tuning builds: build.py --variant mxabl EV_MX_ABL; tools/bench_mxgemm.py): reserved0 bit 4 = one chunk per pass instead of K / 32 and
    // K / 128 (prologue + epilogue of a tile with a minimal main loop), bit 5 = no epilogue (results are garbage by design)
#ifdef EV_MX_ABL
    const int abl = p.reserved0 >> 4;
    const int nkc16 = (abl & 1) ? 1 : nkc16_, n4half = (abl & 1) ? 1 : nkc4;
#else
    const int nkc16 = nkc16_, n4half = nkc4;
#endif
    for (int kc = 0; kc < nkc16; ++kc, ++q) {            // fp16 pass; a next chunk always exists (K % 128 == 0: nkc4 >= 1)
        const bool more = true, lastc = kc + 1 == nkc16;
        const char* const cwb = wb16 + (unsigned)kc * 64u;
        const char* const nxb = uniform_ptr(lastc ? xb4h : xb16 + (unsigned)(kc + 1) * 64u);
        const char* const nwb = uniform_ptr(lastc ? wb4l : wb16 + (unsigned)(kc + 1) * 64u);
        const char* const nsc = sc0;
        const unsigned nsh = lastc ? 0u : 2u;
        const bool nmx = lastc;
        const char* const Xb = Xs + (q & 1) * XBUF;
        int dil_ = p.dil;
        asm volatile("" : "+s"(dil_));            // (opaque per chunk: keeps hipcc from hoisting the fragment addresses of all taps)
#pragma unroll
        for (int u = 0; u < U; ++u) EV_MX_STEP(false, 2, u)
        sbase += U;
    }
    const int n4 = 2 * n4half;
    for (int q4 = 0; q4 < n4; ++q4, ++q) {               // Q(wl).Q(xh) over the K chunks, then Q(wh).Q(xl)
        const bool more = q4 + 1 < n4;
        const int qn = more ? q4 + 1 : q4;
        const bool sec = q4 >= n4half, secn = qn >= n4half;
        const unsigned kc = (unsigned)(sec ? q4 - n4half : q4), kn = (unsigned)(secn ? qn - n4half : qn);
        const char* const cwb = uniform_ptr(wb4l + (sec ? dw4 : 0L) + kc * 64u);
        const char* const nxb = uniform_ptr(xb4h + (secn ? dx4 : 0L) + kn * 64u);
        const char* const nwb = uniform_ptr(wb4l + (secn ? dw4 : 0L) + kn * 64u);
        const char* const nsc = uniform_ptr(sc0 + (secn ? dsc : 0L) + kn * scstep);
        constexpr unsigned nsh = 0u;
        constexpr bool nmx = true;
        const char* const Xb = Xs + (q & 1) * XBUF;
        int dil_ = p.dil;
        asm volatile("" : "+s"(dil_));
#pragma unroll
        for (int u = 0; u < U; ++u) EV_MX_STEP(true, 0, u)
        sbase += U;
    }
#undef EV_MX_STEP
#undef EV_MX_ISSUE_W
#undef EV_MX_ISSUE_X
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (wc == 0) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
    mfma_asm_fence(acc);
    }       // tile_live
#ifdef EV_MX_ABL
    if ((p.reserved0 >> 4) & 2) {
        if (p.M < 0) {          // never: keeps the accumulators alive
            float sum = 0.f;
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
            p.out32[tid] = sum;
        }
        return;
    }
#endif
#ifdef EV_MXT
    const unsigned long long mxt1 = __builtin_amdgcn_s_memrealtime();
    if (p.reserved0 & 1) __builtin_amdgcn_s_setprio(3);          // A/B (tools/bench_mxgemm.py --dbg 1): the epilogue above the partner block's matrix phases
    if (p.reserved0 & 2) __builtin_amdgcn_s_setprio(1);          // ... or level with them
#endif
    if constexpr (EPI == EPI_GENERIC) gemm_epilogue_lds<MT, NT>(p, acc, smem + wave * epi_wave_bytes<TC>(), m0 + wt * 64, n0 + wc * TC);
    else {
        EV_TRACE_EPI_DUMMY
        // 16-row transposing passes through 4 KB of XOR-swizzled scratch per wave: 0-3 % faster than the 32-row padded form and 6-21 fewer live registers
        // in the residual-from-planes variants -- every instantiation compiles without spills (profiles/r4_e_mx_persistent_prefetch_ab.txt, dbg 72)
        gemm_epilogue_fast<MT, NT, EPI, 16>(p, acc, smem + wave * 4096, m0 + wt * 64, n0 + wc * TC EV_TRACE_EPI_ARGS);
    }
#ifdef EV_MXT
    if (p.row_seq && wave == 0) {
        const unsigned long long mxt2b = __builtin_amdgcn_s_memrealtime();          // every store of the wave is issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (... and acknowledged: what s_endpgm waits for anyway)
        const unsigned long long mxt2 = __builtin_amdgcn_s_memrealtime();
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (lane == 0) {
            unsigned long long* o = reinterpret_cast<unsigned long long*>(const_cast<int32_t*>(p.row_seq)) + (size_t)blockIdx.x * 4;
            o[0] = mxt0; o[1] = mxt1; o[2] = mxt2; o[3] = ((unsigned long long)xcc << 48) | ((unsigned long long)(hwid & 0xffffu) << 32) | (unsigned)(mxt2 - mxt2b);
        }
    }
#endif
}

template <int TAPS, int EPI>
__global__ __launch_bounds__(512, 4) void conv_gemm_mx_kernel(const ConvGemmParams p) {
    conv_gemm_mx_tile<TAPS, EPI>(p, (int)blockIdx.x, (int)gridDim.x);
}

// ConvTranspose1d(k = 2 s, stride s, pad s / 2) as a 3-tap conv with N = s * C_out (packer._convT_to_gemm; ConvGemmParams::polyphase_cout = C_out): output phases below
// s / 2 have no weights in tap 2, the others none in tap 0.  A wave's 64 output channels lie inside one phase, so each WAVE runs the instantiation that skips its zero
// tap (same requests, waits and barriers in both: the block's waves stay in step) -- a third of the launch's matrix instructions and fragment reads are not issued.
// The remaining products and their order are those of the plain launch: the same bits.
template <int EPI>
__global__ __launch_bounds__(512, 4) void conv_gemm_mx_up_kernel(const ConvGemmParams p) {
    const int nblk = (int)gridDim.x, nN = p.N >> 7;
    int bid = (int)blockIdx.x;
    {
        const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;          // (conv_gemm_mx_tile's own remap: the tile this block will take)
    }
    const int wc = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int phase = ((bid % nN) * 128 + wc * 64) / p.polyphase_cout, half_s = (p.N / p.polyphase_cout) >> 1;
    if (phase < half_s) conv_gemm_mx_tile<3, EPI, 2>(p, (int)blockIdx.x, nblk);
    else conv_gemm_mx_tile<3, EPI, 0>(p, (int)blockIdx.x, nblk);
}

// Grouped launch (round 6): the same-level convs of a stage's three ResBlocks (k = 11, 7, 3; same M, N, K, same epilogue form) are independent, and a launch of its
// own costs each of them 30-50 us of ramp + tail (tools/bench_mxgemm.py: 2 x T(M / 2) - T(M)).  One grid carries the three problems' tiles back to back, longest
// tiles first: a CU that runs out of k = 11 tiles continues with k = 7 tiles, and the launch ends on the short k = 3 tiles.  Every tile runs the code of its own
// instantiation on its own problem: the results are those of the three launches, bit for bit.
struct ConvGemmGroup3 {
    ConvGemmParams p[3];          // taps 11, 7, 3
    int n0, n01;                  // blocks of problem 0, of problems 0 + 1 (each problem padded to a multiple of 8 blocks)
};
template <int EPI>
__global__ __launch_bounds__(512, 4) void conv_gemm_mx_group3_kernel(const ConvGemmGroup3 g) {
    const int b = (int)blockIdx.x;
    if (b < g.n0) conv_gemm_mx_tile<11, EPI>(g.p[0], b, g.n0);
    else if (b < g.n01) conv_gemm_mx_tile<7, EPI>(g.p[1], b - g.n0, g.n01 - g.n0);
    else conv_gemm_mx_tile<3, EPI>(g.p[2], b - g.n01, (int)gridDim.x - g.n01);
}

#include "ev_gemm_mx1.h"

template <int TAPS, int EPI>
static void launch_mx_epi(const ConvGemmParams& p, hipStream_t s) {
    const int grid = (p.M / PH_BM) * (p.N / 128);
    hipLaunchKernelGGL((conv_gemm_mx_kernel<TAPS, EPI>), dim3(grid), dim3(512), PH_LDS, s, p);
}
// epilogue variant of a DT_MX launch, or -1: fp32 output with the split-precision path's operand sets, each with or without the plane
// set of the result (EPI_MXP); planes-only for conv1 of a ResBlock pair (its fp32 value has no other reader)
static int mx_epi_variant(const ConvGemmParams& p) {
    if (p.taps == 1) return mx1_epi_variant(p);
    const bool rare_act = p.act != ACT_NONE && p.act != ACT_LRELU;
    const bool odd_slope = p.act == ACT_LRELU && !(p.act_slope >= 0.f && p.act_slope <= 1.f);
    const bool mxp = p.mxo_h != nullptr;
    const bool part = p.mxo_partial != 0, accpl = p.acc_h != nullptr;
    if (mxp && !((part || (p.mxo_q4[0] && p.mxo_qs[0])) && p.mxo_q4[1] && p.mxo_qs[1] && p.ldo == p.N && p.mxo_slope >= 0.f && p.mxo_slope <= 1.f &&
                 (p.mxo_logC == 0 || (p.mxo_logC >= 6 && p.mxo_logC <= 12 && p.N % (1 << p.mxo_logC) == 0)))) return -1;
    if (mxp && rare_act && !p.out32 && !p.res && !p.acc32 && !p.seq_bias && !p.add16_a && !p.out16 && !p.post_lrelu)
        return EPI_RARE_ACT | EPI_MXP;                                       // conv-FFN's first conv: erf-GELU, planes only
    const bool respl = p.res && p.res_dtype == DT_MX;          // residual from a plane set (fp16 hi plane + fp4 remainder codes + scales)
    if (part || accpl) {
        // the MRF sum of a stage as partial plane sets: the last conv of a ResBlock adds the running sum (a partial plane set, or nothing for the first
        // ResBlock) and writes the new one (partial) or -- the last ResBlock -- the next up-conv's full plane set.  Planes only, residual from planes.
        if (!(respl && mxp && !p.out32 && !p.acc32 && !p.seq_bias && !p.add16_a && !p.out16 && !p.post_lrelu && !odd_slope && !rare_act)) return -1;
        if (!(p.res_x4 && p.res_xs && p.ldres == p.N && p.res_inv_slope >= 1.0f)) return -1;
        if (accpl && !(p.acc_x4 && p.acc_xs && p.ldacc == p.N)) return -1;
        if (part && (p.mxo_slope != 1.0f || !(p.mxo_logC == 0 || (1 << p.mxo_logC) == p.N))) return -1;          // raw values, the launch's own [M][N] geometry
        return EPI_RESPL | EPI_LEAN | EPI_MXP | (accpl ? EPI_ACCPL : 0) | (part ? EPI_PART : 0);
    }
    if (p.seq_bias || p.add16_a || p.out16 || p.out32_before_post || p.post_lrelu || odd_slope || rare_act) return (mxp || respl) ? -1 : EPI_GENERIC;
    const bool res32 = p.res && p.res_dtype == DT_F32;
    const int m = mxp ? EPI_MXP : 0;
    if (respl) {
        if (!(p.res_x4 && p.res_xs && p.ldres == p.N && p.res_inv_slope >= 1.0f)) return -1;
        // planes only: conv2 of a pair inside a ResBlock; with the running MRF sum added: the last conv of a stage whose fp32 value only the next up-conv's planes carry
        if (!p.out32) return mxp ? (p.acc32 ? (EPI_RESPL | EPI_ACC32 | EPI_LEAN | EPI_MXP) : (EPI_RESPL | EPI_LEAN | EPI_MXP)) : -1;
        return p.acc32 ? (EPI_RESPL | EPI_ACC32 | EPI_O32 | EPI_LEAN | m) : (EPI_RESPL | EPI_O32 | EPI_LEAN | m);
    }
    if (p.res && !res32) return mxp ? -1 : EPI_GENERIC;
    if (!p.out32) return (mxp && !p.res && !p.acc32) ? EPI_MXP : -1;
    if (!p.acc32 && !p.res) return EPI_O32 | m;
    if (!p.acc32 && res32) return EPI_RES32 | EPI_O32 | EPI_LEAN | m;
    if (p.acc32 && res32) return EPI_RES32 | EPI_ACC32 | EPI_O32 | EPI_LEAN | m;
    return mxp ? -1 : EPI_GENERIC;
}
#define EV_MX_VARIANTS(X) X(EPI_O32) X(EPI_RES32 | EPI_O32 | EPI_LEAN) X(EPI_RES32 | EPI_ACC32 | EPI_O32 | EPI_LEAN) X(EPI_MXP) X(EPI_O32 | EPI_MXP) \
    X(EPI_RES32 | EPI_O32 | EPI_LEAN | EPI_MXP) X(EPI_RES32 | EPI_ACC32 | EPI_O32 | EPI_LEAN | EPI_MXP) X(EPI_RARE_ACT | EPI_MXP) X(EPI_GENERIC)   \
    X(EPI_RESPL | EPI_LEAN | EPI_MXP) X(EPI_RESPL | EPI_O32 | EPI_LEAN) X(EPI_RESPL | EPI_O32 | EPI_LEAN | EPI_MXP)                                    \
    X(EPI_RESPL | EPI_ACC32 | EPI_O32 | EPI_LEAN) X(EPI_RESPL | EPI_ACC32 | EPI_O32 | EPI_LEAN | EPI_MXP) X(EPI_RESPL | EPI_ACC32 | EPI_LEAN | EPI_MXP)           \
    X(EPI_RESPL | EPI_LEAN | EPI_MXP | EPI_PART) X(EPI_RESPL | EPI_ACCPL | EPI_LEAN | EPI_MXP | EPI_PART) X(EPI_RESPL | EPI_ACCPL | EPI_LEAN | EPI_MXP)           \
    EV_MX_VARIANTS_NL(X)
// (round 6 A/B, tuning builds only -- EV_MX_NOLEAN: the plane-set-residual epilogues with the NEXT pass's operands requested one pass ahead (two register sets) instead of at
//  the top of their own pass; chosen per launch by reserved0 bit 1)
#ifdef EV_MX_NOLEAN
#define EV_MX_VARIANTS_NL(X) X(EPI_RESPL | EPI_MXP) X(EPI_RESPL | EPI_MXP | EPI_PART) X(EPI_RESPL | EPI_ACCPL | EPI_MXP | EPI_PART) X(EPI_RESPL | EPI_ACCPL | EPI_MXP)
#else
#define EV_MX_VARIANTS_NL(X)
#endif
template <int TAPS>
static void launch_mx_taps(const ConvGemmParams& p, int e, hipStream_t s) {
    if constexpr (TAPS == 3) {
        if (p.polyphase_cout > 0 && (e == EPI_MXP || e == (EPI_O32 | EPI_MXP))) {
            const int grid = (p.M / PH_BM) * (p.N / 128);
            if (e == EPI_MXP) hipLaunchKernelGGL((conv_gemm_mx_up_kernel<EPI_MXP>), dim3(grid), dim3(512), PH_LDS, s, p);
            else hipLaunchKernelGGL((conv_gemm_mx_up_kernel<EPI_O32 | EPI_MXP>), dim3(grid), dim3(512), PH_LDS, s, p);
            return;
        }
    }
#ifdef EV_MX_NOLEAN
    if ((p.reserved0 & 2) && (e & EPI_RESPL) && (e & EPI_MXP) && !(e & (EPI_O32 | EPI_ACC32))) e &= ~EPI_LEAN;
#endif
    switch (e) {
#define EV_MX_CASE(E) case (E): launch_mx_epi<TAPS, (E)>(p, s); break;
        EV_MX_VARIANTS(EV_MX_CASE)
#undef EV_MX_CASE
        default: break;
    }
}
template <int TAPS>
static hipError_t mx_attr_taps() {
    hipError_t e = hipSuccess, r;
#define EV_MX_ATTR(E) r = hipFuncSetAttribute((const void*)conv_gemm_mx_kernel<TAPS, (E)>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    EV_MX_VARIANTS(EV_MX_ATTR)
#undef EV_MX_ATTR
    return e;
}
static hipError_t mx_set_attributes() {
    hipError_t e = hipSuccess, r;
    r = mx_attr_taps<3>(); if (r != hipSuccess) e = r;
    r = mx_attr_taps<7>(); if (r != hipSuccess) e = r;
    r = mx_attr_taps<11>(); if (r != hipSuccess) e = r;
    r = mx1_set_attributes(); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_gemm_mx_up_kernel<EPI_MXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_gemm_mx_up_kernel<EPI_O32 | EPI_MXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_gemm_mx_group3_kernel<EPI_MXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    r = hipFuncSetAttribute((const void*)conv_gemm_mx_group3_kernel<EPI_RESPL | EPI_LEAN | EPI_MXP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PH_LDS); if (r != hipSuccess) e = r;
    return e;
}

// Which calls take the MX kernel is a function of the layer's shape only (never of M): an utterance gets the same arithmetic alone
// and inside a batch.  Everything else of a DT_MX call runs as the split-precision (three fp16 MFMAs) kernel.
static bool mx_shape_ok(const ConvGemmParams& p) {
    if (p.taps == 1) return mx1_shape_ok(p);          // nn.Linear: the one-tap pipeline (ev_gemm_mx1.h)
    // (polyphase hint: a wave's 64 output channels must lie inside one phase, and the phases split evenly into the two tap pairs)
    if (p.polyphase_cout && (p.taps != 3 || p.polyphase_cout < 0 || p.polyphase_cout % 64 || p.N % p.polyphase_cout || ((p.N / p.polyphase_cout) & 1))) return false;
    return p.W_mx && p.N % 128 == 0 && p.K % 128 == 0 && (p.taps == 3 || p.taps == 7 || p.taps == 11) && p.M % PH_BM == 0 &&
           p.lda == p.K && (p.taps - 1) * p.dil <= MAX_SPAN;
}
static bool mx_planes_in(const ConvGemmParams& p) { return p.mx_x4[0] != nullptr; }
static bool mx_eligible(const ConvGemmParams& p) {
    if (!mx_shape_ok(p) || mx_epi_variant(p) < 0) return false;
    if (mx_planes_in(p)) return p.mx_x4[1] && p.mx_xs[0] && p.mx_xs[1] && !p.pro_lrelu;
    return p.mx_scratch && p.mx_scratch_size >= mx_scratch_bytes(p.M, p.K);
}
// 0 = this DT_MX call can run (as the MX kernel, or -- fp32 input, no plane output -- as the split-precision fallback)
static bool conv64_mx_eligible(const ConvGemmParams& p);
static bool mx64_eligible(const ConvGemmParams& p);
int mx_check(const ConvGemmParams& p) {
    if (p.dtype != DT_MX) return p.mxo_h || mx_planes_in(p) || (p.res && p.res_dtype == DT_MX) || p.acc_h || p.mxo_partial ? -1 : 0;       // plane sets exist only between DT_MX launches
    if (mx_eligible(p) || mx64_eligible(p) || conv64_mx_eligible(p)) return 0;
    return (p.mxo_h || mx_planes_in(p) || !p.W_lo || (p.res && p.res_dtype == DT_MX) || p.acc_h || p.mxo_partial) ? -1 : 0;
}
MxScratchPlanes mx_scratch_planes(void* scratch, int M, int K) {
    const MxPlaneLayout L = mx_layout(M, K);
    char* const base = reinterpret_cast<char*>(scratch);
    const size_t r0 = MX_SLACK;
    MxScratchPlanes v;
    v.h = base + L.h + r0 * K * 2;
    v.q4[0] = base + L.x4h + r0 * (K / 2); v.q4[1] = base + L.x4l + r0 * (K / 2);
    v.qs[0] = base + L.sh + r0 * 4; v.qs[1] = base + L.sl + r0 * 4;
    v.qs_stride = (unsigned)(L.rows * 4);
    return v;
}
static void launch_mx(const ConvGemmParams& p_in, hipStream_t s) {
    ConvGemmParams p = p_in;
    if (!mx_planes_in(p)) {
        const MxScratchPlanes sp = mx_scratch_planes(p.mx_scratch, p.M, p.K);
        __half* H = reinterpret_cast<__half*>(sp.h);
        uint8_t* x4h = reinterpret_cast<uint8_t*>(sp.q4[0]);
        uint8_t* x4l = reinterpret_cast<uint8_t*>(sp.q4[1]);
        uint8_t* sh = reinterpret_cast<uint8_t*>(sp.qs[0]);
        uint8_t* sl = reinterpret_cast<uint8_t*>(sp.qs[1]);
        const unsigned s_stride = sp.qs_stride;
        const long threads = (long)p.M * (p.K / 8);
        hipLaunchKernelGGL(mx_planes_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float*>(p.A), p.lda, p.M, p.K,
                           p.pro_lrelu, p.pro_slope, H, x4h, x4l, sh, sl, s_stride);
        p.A = H; p.lda = p.K; p.pro_lrelu = 0;
        p.mx_x4[0] = x4h; p.mx_x4[1] = x4l; p.mx_xs[0] = sh; p.mx_xs[1] = sl; p.mx_xs_stride = s_stride;
    }
    if (p.taps == 1) { launch_mx1_kernel(p, s); return; }
    const int e = mx_epi_variant(p);
    switch (p.taps) {
        case 3: launch_mx_taps<3>(p, e, s); break;
        case 7: launch_mx_taps<7>(p, e, s); break;
        default: launch_mx_taps<11>(p, e, s); break;
    }
}

// Grouped launch of three independent DT_MX convs (see conv_gemm_mx_group3_kernel): plane sets in, the same M / N / K and epilogue form (conv1 of a ResBlock pair:
// planes only; conv2 inside a ResBlock: residual from planes, planes only), taps {3, 7, 11} in any order.  0 = launched as one grid; -1 = not such a triple
// (the caller launches them one by one).  Same bits either way.
int launch_conv_gemm_group3(const ConvGemmParams* ps, hipStream_t s, bool check_only) {
    int order[3] = {-1, -1, -1};          // problem with 11, 7, 3 taps
    for (int i = 0; i < 3; ++i) {
        const ConvGemmParams& p = ps[i];
        if (p.dtype != DT_MX || p.taps == 1 || !mx_planes_in(p) || !mx_eligible(p) || p.reserved0) return -1;
        if (p.M != ps[0].M || p.N != ps[0].N || p.K != ps[0].K) return -1;
        const int slot = p.taps == 11 ? 0 : (p.taps == 7 ? 1 : (p.taps == 3 ? 2 : -1));
        if (slot < 0 || order[slot] >= 0) return -1;
        order[slot] = i;
    }
    const int e = mx_epi_variant(ps[0]);
    if (e != mx_epi_variant(ps[1]) || e != mx_epi_variant(ps[2])) return -1;
    if (e != EPI_MXP && e != (EPI_RESPL | EPI_LEAN | EPI_MXP)) return -1;
    if (check_only) return 0;
    ConvGemmGroup3 g;
    for (int i = 0; i < 3; ++i) g.p[i] = ps[order[i]];
    const int tiles = (ps[0].M / PH_BM) * (ps[0].N / 128), padded = (tiles + 7) & ~7;
    g.n0 = padded; g.n01 = 2 * padded;
    const int grid = 2 * padded + tiles;          // (the last problem needs no padding blocks)
    if (e == EPI_MXP) hipLaunchKernelGGL((conv_gemm_mx_group3_kernel<EPI_MXP>), dim3(grid), dim3(512), PH_LDS, s, g);
    else hipLaunchKernelGGL((conv_gemm_mx_group3_kernel<EPI_RESPL | EPI_LEAN | EPI_MXP>), dim3(grid), dim3(512), PH_LDS, s, g);
    return 0;
}
