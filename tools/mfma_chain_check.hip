// does a dependent chain  v_mfma_f32_16x16x32_f16 -> legacy v_mfma_f32_16x16x16_f16  on ONE accumulator lose products on the MI355X
// (DESIGN section 4, round-4 item 6 claimed so after the attention kernel measured 5e-5 ... 4e-4, not repeatable), or was it a missing software wait state?
//
// Exact test: operands are small integers (|v| <= 3, sums < 2^11: every fp16 product and fp32 sum is exact), the expected accumulator is computed on
// the host in integers.  Every variant runs the chain  acc = 0; acc = K32(a, b, acc); acc = K16(a2, b2, acc); [hi/lo style second accumulator likewise]
// many times per wave with operands re-read from memory, 4 waves per SIMD resident, and counts accumulator words that differ from the expectation:
//   builtin      both MFMAs as compiler builtins (hipcc pads by its own hazard table)
//   asm0         both as inline asm, in place ("+v"), NOTHING between them (what an asm kernel without the fence would do), 20 wait states before the read
//   asmN         the same with s_nop N between the two MFMAs (N = 3, 7, 15)
//   asm_noread   asm chain, and the accumulator read by a VALU op after only 2 wait states (the round-3 bug of ev_pair_mx.h, as a positive control: this
//                one is EXPECTED to fail sometimes -- it shows the test can see a lost product)
//   k32k32       control: two K = 32 MFMAs (second with a zero half), the form the attention kernel ships
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/mfma_chain_check tools/mfma_chain_check.hip && tools/build/mfma_chain_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

enum { V_BUILTIN = 0, V_ASM0, V_ASM3, V_ASM7, V_ASM15, V_ASM_NOREAD, V_K32K32, V_ASM4, V_ASM5, V_ASM6, V_K16K32_BUILTIN, V_K16K32_ASM0, V_F16_MX_ASM0, V_MX_F16_ASM0, V_F16_MX_ASM7, V_F16_MXB8_ASM0, V_MXB8_F16_ASM0, V_MX4_MXB8_ASM0, V_MXB8_MX4_ASM0, NVAR };
static const char* VNAME[NVAR] = {"builtin K32->K16", "asm K32->K16, 0 states between", "asm, s_nop 3 between", "asm, s_nop 7 between", "asm, s_nop 15 between",
                                  "asm, VALU read after 2 states (positive control)", "builtin K32->K32 (shipped form)",
                                  "asm, s_nop 4 between (5 states)", "asm, s_nop 5 between (6 states)", "asm, s_nop 6 between (7 states)",
                                  "builtin K16->K32 (reverse order)", "asm K16->K32, 0 states between",
                                  "asm f16 K32 -> fp4 scale MFMA, 0 states (the MX kernels' pass 0 -> pass 1)", "asm fp4 scale MFMA -> f16 K32, 0 states",
                                  "asm f16 K32 -> fp4 scale MFMA, s_nop 7 between",
                                  "asm f16 K32 -> fp4 x bf8 scale MFMA (cbsz:4 blgp:1), 0 states (round 6: E5M2 activations)", "asm fp4 x bf8 scale MFMA -> f16 K32, 0 states",
                                  "asm fp4 x fp4 scale MFMA -> fp4 x bf8 scale MFMA, 0 states", "asm fp4 x bf8 scale MFMA -> fp4 x fp4 scale MFMA, 0 states"};

template <int VAR>
__global__ __launch_bounds__(256, 4) void chain_kernel(const h8* __restrict__ A, const h8* __restrict__ B, const h4* __restrict__ A2, const h4* __restrict__ B2,
                                                       f4* __restrict__ out, int iters, int nsets, const u32x4* __restrict__ QA, const u32x4* __restrict__ QB,
                                                       const u32x8* __restrict__ QB8) {
    const int lane = threadIdx.x & 63, gw = blockIdx.x * 4 + (threadIdx.x >> 6);
    f4 sum = f4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const int set = (gw + it) % nsets;
        const h8 a = A[set * 64 + lane], b = B[set * 64 + lane];
        const h4 a2 = A2[set * 64 + lane], b2 = B2[set * 64 + lane];
        f4 acc = f4{0.f, 0.f, 0.f, 0.f};
        if constexpr (VAR == V_BUILTIN) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a2, b2, acc, 0, 0, 0);
        } else if constexpr (VAR == V_K32K32) {
            const h4 z = h4{(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            const h8 a2w = __builtin_shufflevector(a2, z, 0, 1, 2, 3, 4, 5, 6, 7), b2w = __builtin_shufflevector(b2, z, 0, 1, 2, 3, 4, 5, 6, 7);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2w, b2w, acc, 0, 0, 0);
        } else if constexpr (VAR == V_K16K32_BUILTIN) {
            acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a2, b2, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
        } else if constexpr (VAR == V_F16_MX_ASM0 || VAR == V_MX_F16_ASM0 || VAR == V_F16_MX_ASM7) {
            // fp4 operands: 16 bytes per lane = 32 e2m1 codes (K block lane >> 4 of 128), unit E8M0 scales (127).  Built from a2 / b2's integer values
            // in {-3..3}: all exactly representable in e2m1 (0, 0.5, 1, 1.5, 2, 3, 4, 6).  The host expectation adds sum_k qa[r][k] qb[k][c] over K = 128.
            const u32x4 qa = QA[set * 64 + lane], qb = QB[set * 64 + lane];
            const int one = 127;
            asm volatile("s_nop 1" ::: "memory");
            if constexpr (VAR == V_F16_MX_ASM0)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %3, %4, %0, %5, %5 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                             : "+v"(acc) : "v"(a), "v"(b), "v"(qa), "v"(qb), "v"(one));
            else if constexpr (VAR == V_F16_MX_ASM7)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %3, %4, %0, %5, %5 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                             : "+v"(acc) : "v"(a), "v"(b), "v"(qa), "v"(qb), "v"(one));
            else
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %3, %4, %0, %5, %5 op_sel_hi:[0,0,0] cbsz:4 blgp:4\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                             : "+v"(acc) : "v"(a), "v"(b), "v"(qa), "v"(qb), "v"(one));
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc));
        } else if constexpr (VAR == V_F16_MXB8_ASM0 || VAR == V_MXB8_F16_ASM0 || VAR == V_MX4_MXB8_ASM0 || VAR == V_MXB8_MX4_ASM0) {
            // (round 6) the B operand of the block-scaled MFMA as E5M2: 32 bytes per lane (byte k = element k of the lane's K block), the same integers as QB
            const u32x4 qa = QA[set * 64 + lane], qb = QB[set * 64 + lane];
            const u32x8 qb8 = QB8[set * 64 + lane];          // (8-bit layout: lane group j = elements 16 j .. + 15 in registers 0-3, 64 + 16 j .. in 4-7)
            const int one = 127;
            asm volatile("s_nop 1" ::: "memory");
            if constexpr (VAR == V_F16_MXB8_ASM0)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_scale_f32_16x16x128_f8f6f4 %0, %3, %4, %0, %5, %5 op_sel_hi:[0,0,0] cbsz:4 blgp:1"
                             : "+v"(acc) : "v"(a), "v"(b), "v"(qa), "v"(qb8), "v"(one));
            else if constexpr (VAR == V_MXB8_F16_ASM0)
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %3, %4, %0, %5, %5 op_sel_hi:[0,0,0] cbsz:4 blgp:1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                             : "+v"(acc) : "v"(a), "v"(b), "v"(qa), "v"(qb8), "v"(one));
            else if constexpr (VAR == V_MX4_MXB8_ASM0)
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %4, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4\n\t"
                             "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %3, %0, %4, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:1"
                             : "+v"(acc) : "v"(qa), "v"(qb), "v"(qb8), "v"(one));
            else
                asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %3, %0, %4, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:1\n\t"
                             "v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %4, %4 op_sel_hi:[0,0,0] cbsz:4 blgp:4"
                             : "+v"(acc) : "v"(qa), "v"(qb), "v"(qb8), "v"(one));
            asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc));
        } else {
            // operands are settled (loaded, waited for by the compiler, then two states of distance) before the string starts
            asm volatile("s_nop 1" ::: "memory");
            if constexpr (VAR == V_ASM0 || VAR == V_ASM_NOREAD)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(a2), "v"(b2));
            else if constexpr (VAR == V_K16K32_ASM0)
                asm volatile("v_mfma_f32_16x16x16_f16 %0, %3, %4, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(a2), "v"(b2));
            else if constexpr (VAR == V_ASM4)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 4\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(a2), "v"(b2));
            else if constexpr (VAR == V_ASM5)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 5\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(a2), "v"(b2));
            else if constexpr (VAR == V_ASM6)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 6\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(a2), "v"(b2));
            else if constexpr (VAR == V_ASM3)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 3\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(a2), "v"(b2));
            else if constexpr (VAR == V_ASM7)
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(a2), "v"(b2));
            else
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0\n\ts_nop 7\n\ts_nop 7\n\tv_mfma_f32_16x16x16_f16 %0, %3, %4, %0" : "+v"(acc) : "v"(a), "v"(b), "v"(a2), "v"(b2));
            if constexpr (VAR == V_ASM_NOREAD) asm volatile("s_nop 0" : "+v"(acc));                       // 1 + hipcc's boundary state: too early on purpose
            else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(acc));                               // 20 states: the library's mfma_asm_fence
        }
        sum += acc * (float)(1 + (it & 3));
    }
    out[(size_t)gw * 64 + lane] = sum;
}

// 16x16x32: lane l holds A[row = l & 15][k = 8 (l >> 4) .. + 8], B[k = 8 (l >> 4) .. + 8][col = l & 15]; 16x16x16: k = 4 (l >> 4) .. + 4.
// D: lane l holds D[row = 4 (l >> 4) + i][col = l & 15], i = 0..3
int main() {
    const int nsets = 64, blocks = 1024 * 4, iters = 400;
    std::vector<_Float16> A(nsets * 64 * 8), B(nsets * 64 * 8), A2(nsets * 64 * 4), B2(nsets * 64 * 4);
    std::vector<float> exp_set((size_t)nsets * 64 * 4), exp_mx((size_t)nsets * 64 * 4);
    std::vector<unsigned> QA((size_t)nsets * 64 * 4, 0u), QB((size_t)nsets * 64 * 4, 0u), QB8((size_t)nsets * 64 * 8, 0u);
    std::vector<float> exp_mx2((size_t)nsets * 64 * 4);
    auto bf8 = [](int v) { const unsigned mag[4] = {0x00u, 0x3Cu, 0x40u, 0x42u}; return (v < 0 ? 0x80u : 0u) | mag[v < 0 ? -v : v]; };      // E5M2 codes of 0, 1, 2, 3
    auto fp4 = [](int v) { const unsigned mag[4] = {0u, 2u, 4u, 5u}; return (v < 0 ? 8u : 0u) | mag[v < 0 ? -v : v]; };
    unsigned rng = 12345u;
    auto rnd = [&]() { rng = rng * 1664525u + 1013904223u; return (int)((rng >> 24) % 7) - 3; };
    for (int s = 0; s < nsets; ++s) {
        int a[16][32], b[32][16], a2[16][16], b2[16][16];
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { a[i][k] = rnd(); b[k][i] = rnd(); }
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 16; ++k) { a2[i][k] = rnd(); b2[k][i] = rnd(); }
        static int qa[16][128], qb[128][16];
        for (int i = 0; i < 16; ++i) for (int k = 0; k < 128; ++k) { qa[i][k] = rnd(); qb[k][i] = rnd(); }
        for (int l = 0; l < 64; ++l)          // lane l: row / column l & 15, K block l >> 4 (32 elements = 16 bytes, element 2j in the low nibble of byte j)
            for (int e = 0; e < 32; ++e) {
                QA[((size_t)s * 64 + l) * 4 + e / 8] |= fp4(qa[l & 15][32 * (l >> 4) + e]) << (4 * (e % 8));
                QB[((size_t)s * 64 + l) * 4 + e / 8] |= fp4(qb[32 * (l >> 4) + e][l & 15]) << (4 * (e % 8));
                QB8[((size_t)s * 64 + l) * 8 + e / 4] |= bf8(qb[(e < 16 ? 16 * (l >> 4) + e : 64 + 16 * (l >> 4) + e - 16)][l & 15]) << (8 * (e % 4));
            }
        for (int l = 0; l < 64; ++l) {
            for (int e = 0; e < 8; ++e) { A[(s * 64 + l) * 8 + e] = (_Float16)a[l & 15][8 * (l >> 4) + e]; B[(s * 64 + l) * 8 + e] = (_Float16)b[8 * (l >> 4) + e][l & 15]; }
            for (int e = 0; e < 4; ++e) { A2[(s * 64 + l) * 4 + e] = (_Float16)a2[l & 15][4 * (l >> 4) + e]; B2[(s * 64 + l) * 4 + e] = (_Float16)b2[4 * (l >> 4) + e][l & 15]; }
            for (int i = 0; i < 4; ++i) {
                const int r = 4 * (l >> 4) + i, c = l & 15;
                int v = 0;
                for (int k = 0; k < 32; ++k) v += a[r][k] * b[k][c];
                for (int k = 0; k < 16; ++k) v += a2[r][k] * b2[k][c];
                exp_set[((size_t)s * 64 + l) * 4 + i] = (float)v;
                int w = 0;
                for (int k = 0; k < 32; ++k) w += a[r][k] * b[k][c];
                for (int k = 0; k < 128; ++k) w += qa[r][k] * qb[k][c];
                exp_mx[((size_t)s * 64 + l) * 4 + i] = (float)w;
                int w2 = 0;
                for (int k = 0; k < 128; ++k) w2 += 2 * qa[r][k] * qb[k][c];
                exp_mx2[((size_t)s * 64 + l) * 4 + i] = (float)w2;
            }
        }
    }
    h8 *dA, *dB; h4 *dA2, *dB2; f4* dout;
    CHK(hipMalloc(&dA, A.size() * 2)); CHK(hipMalloc(&dB, B.size() * 2)); CHK(hipMalloc(&dA2, A2.size() * 2)); CHK(hipMalloc(&dB2, B2.size() * 2));
    u32x4 *dQA, *dQB; u32x8* dQB8;
    CHK(hipMalloc(&dQA, QA.size() * 4)); CHK(hipMalloc(&dQB, QB.size() * 4)); CHK(hipMalloc(&dQB8, QB8.size() * 4));
    CHK(hipMemcpy(dQB8, QB8.data(), QB8.size() * 4, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dQA, QA.data(), QA.size() * 4, hipMemcpyHostToDevice)); CHK(hipMemcpy(dQB, QB.data(), QB.size() * 4, hipMemcpyHostToDevice));
    const size_t nout = (size_t)blocks * 4 * 64;
    CHK(hipMalloc(&dout, nout * sizeof(f4)));
    CHK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dA2, A2.data(), A2.size() * 2, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB2, B2.data(), B2.size() * 2, hipMemcpyHostToDevice));
    // expectation per (wave, lane, i): sum over it of exp_set[(gw + it) % nsets] * (1 + (it & 3)): small integers, exact in fp32 (|.| < 2^24)
    std::vector<float> expect(nout * 4), expect_mx(nout * 4), expect_mx2(nout * 4), got(nout * 4);
    for (int tab = 0; tab < 3; ++tab) {
        std::vector<float>& ex = tab == 2 ? expect_mx2 : tab ? expect_mx : expect;
        const std::vector<float>& es = tab == 2 ? exp_mx2 : tab ? exp_mx : exp_set;
        for (size_t gw = 0; gw < (size_t)blocks * 4; ++gw) {
            if (gw >= (size_t)nsets) {          // the operand sequence of a wave depends on gw % nsets only
                memcpy(&ex[gw * 256], &ex[(gw % nsets) * 256], 256 * sizeof(float));
                continue;
            }
            for (int l = 0; l < 64; ++l)
                for (int i = 0; i < 4; ++i) {
                    double v = 0;
                    for (int it = 0; it < iters; ++it) v += (double)es[(((gw + it) % nsets) * 64 + l) * 4 + i] * (1 + (it & 3));
                    ex[(gw * 64 + l) * 4 + i] = (float)v;
                }
        }
    }
    int bad_total = 0;
    for (int var = 0; var < NVAR; ++var) {
        long bad = 0, runs_bad = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CHK(hipMemset(dout, 0, nout * sizeof(f4)));
            switch (var) {
#define LAUNCH(V) case V: hipLaunchKernelGGL(chain_kernel<V>, dim3(blocks), dim3(256), 0, 0, dA, dB, dA2, dB2, dout, iters, nsets, dQA, dQB, dQB8); break;
                LAUNCH(V_BUILTIN) LAUNCH(V_ASM0) LAUNCH(V_ASM3) LAUNCH(V_ASM7) LAUNCH(V_ASM15) LAUNCH(V_ASM_NOREAD) LAUNCH(V_K32K32) LAUNCH(V_ASM4) LAUNCH(V_ASM5) LAUNCH(V_ASM6) LAUNCH(V_K16K32_BUILTIN) LAUNCH(V_K16K32_ASM0)
                LAUNCH(V_F16_MX_ASM0) LAUNCH(V_MX_F16_ASM0) LAUNCH(V_F16_MX_ASM7)
                LAUNCH(V_F16_MXB8_ASM0) LAUNCH(V_MXB8_F16_ASM0) LAUNCH(V_MX4_MXB8_ASM0) LAUNCH(V_MXB8_MX4_ASM0)
#undef LAUNCH
            }
            CHK(hipDeviceSynchronize());
            CHK(hipMemcpy(got.data(), dout, nout * sizeof(f4), hipMemcpyDeviceToHost));
            long b = 0;
            const std::vector<float>& ex = (var == V_MX4_MXB8_ASM0 || var == V_MXB8_MX4_ASM0) ? expect_mx2 :
                                           (var == V_F16_MX_ASM0 || var == V_MX_F16_ASM0 || var == V_F16_MX_ASM7 || var == V_F16_MXB8_ASM0 || var == V_MXB8_F16_ASM0) ? expect_mx : expect;
            for (size_t i = 0; i < got.size(); ++i) b += got[i] != ex[i];
            bad += b; runs_bad += b != 0;
        }
        printf("%-52s wrong accumulator words: %ld of %zu x 5 runs (%ld runs affected)%s\n", VNAME[var], bad, got.size(), runs_bad,
               var == V_ASM_NOREAD ? "   [expected to fail: shows the check can see a lost product]" : "");
        if (var != V_ASM_NOREAD) bad_total += bad != 0;
    }
    printf("%s\n", bad_total ? "RESULT: a chain variant other than the positive control lost products" : "RESULT: every K32->K16 chain variant is exact; only the early VALU read (if any) fails");
    return 0;
}
