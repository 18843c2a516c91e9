// MFMA issue-rate / power-limit micro-benchmark for gfx950 (tuning tool, not part of the product).
//
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_ubench tools/mfma_ubench.hip && gpurun_out/mfma_ubench
//
// Answers (VERDICT r2 "next" #5): what does the matrix pipe sustain on RANDOM operands, register-resident (no LDS, no
// global traffic), with the phased conv-GEMM kernel's instruction mix (16 independent 16x16x32 f16 MFMAs per step on a
// 64 x 64 wave tile)?  The same for 32x32x16 f16, for the block-scaled MX instruction (fp8 / fp6 / fp4 operands, K = 128)
// and for the "hi.hi in fp16 + cross terms in MX" mixes the split-precision mode could issue instead of three fp16 MFMAs.
// Also checks the MX operand layout (fp4 nibble order, per-lane E8M0 scale byte) against a host emulation.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

__device__ __forceinline__ h8 rand_h8(uint32_t& s, bool zero) {
    h8 r;
    for (int i = 0; i < 8; ++i) r[i] = zero ? (_Float16)0.f : (_Float16)(((int)(lcg(s) >> 8) % 2001 - 1000) * 1e-3f);
    return r;
}
__device__ __forceinline__ i8v rand_i8(uint32_t& s, bool zero, uint32_t mask) {
    i8v r;
    for (int i = 0; i < 8; ++i) r[i] = zero ? 0 : (int)(lcg(s) & mask);
    return r;
}

// MODE 0: 16 x (16x16x32 f16) per step            (1 unit)
// MODE 1:  4 x (32x32x16 f16) per step            (same FLOPs as mode 0: 64x64x32 per wave and step)
// MODE 2: per 4 steps (K = 128): 64 f16 16x16x32 + 32 MX (2 cross terms x 16 tiles), MX format = FMT
// MODE 3: MX only: 16 x (16x16x128) per step
// (round 6) FA / FB: the formats of the A and of the B operand (0 = fp8 e4m3, 1 = bf8 e5m2, 2 = fp6, 4 = fp4); FA = 4, FB = 1 is the "fp4 weights x E5M2
// activations" cross term of the maxima-free quantiser: the instruction then runs at the wider operand's rate
template <int MODE, int FA, int FB>
__global__ __launch_bounds__(256, 2) void ubench(float* out, long long* ticks, int iters, int zero) {
    uint32_t s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
    const uint32_t mask = (FA == 0 || FB == 0) ? 0x7e7e7e7eu | 0x80808080u : (FA == 1 || FB == 1) ? 0x7b7b7b7bu | 0x80808080u : 0xffffffffu;   // fp8 / bf8: no NaN / inf codes
    h8 a[4], b[4];
    i8v ma[4], mb[4], na[4], nb[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = rand_h8(s, zero); b[i] = rand_h8(s, zero);
        ma[i] = rand_i8(s, zero, mask); mb[i] = rand_i8(s, zero, mask);
        na[i] = rand_i8(s, zero, mask); nb[i] = rand_i8(s, zero, mask);
    }
    f4 acc[16];
    f16v acc32[4];
    for (int i = 0; i < 16; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
    const int sc = 0x7f7f7f7f - 0x0b0b0b0b * (MODE == 2);   // cross terms carry 2^-11
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 2) {
#pragma unroll
            for (int rep = 0; rep < (MODE == 2 ? 4 : 1); ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
        }
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc32[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc32[i * 2 + j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc32[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 2], b[j + 2], acc32[i * 2 + j], 0, 0, 0);
        }
        if (MODE == 2 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ma[i], mb[j], acc[i * 4 + j], FA, FB, 0, sc, 0, sc);
                    if (MODE == 2)
                        acc[i * 4 + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(na[i], nb[j], acc[i * 4 + j], FA, FB, 0, sc, 0, sc);
                }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc32[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}

struct Result { double ms, tflops_f16, tflops_mx, ghz; };

template <int MODE, int FA, int FB = FA>
static Result run(const char* name, int iters, int zero, float* d_out, long long* d_ticks) {
    const int blocks = 256 * 2 * 4;     // 2 blocks of 4 waves per CU resident, 4 rounds
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Result best{1e30, 0, 0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((ubench<MODE, FA, FB>), dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, iters, zero);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long ticks; CK(hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost));
        double waves = blocks * 4.0;
        double f16_flops = 0, mx_flops = 0;
        if (MODE == 0 || MODE == 1) f16_flops = waves * iters * 2.0 * 64 * 64 * 32;
        if (MODE == 2) { f16_flops = waves * iters * 2.0 * 64 * 64 * 128; mx_flops = 2 * f16_flops; }
        if (MODE == 3) mx_flops = waves * iters * 2.0 * 64 * 64 * 128;
        Result r{ms, f16_flops / ms * 1e-9, mx_flops / ms * 1e-9, 0};
        if (rep > 0 && ms < best.ms) best = r;   // first launch warms up
        printf("  %-44s zero=%d rep %d: %8.3f ms  f16 %8.1f TF/s  mx %8.1f TF/s  block0 wave cycles/iter %.1f\n", name, zero, rep, ms,
               r.tflops_f16, r.tflops_mx, (double)ticks / iters);
    }
    return best;
}

// ---------------------------------------------------------------- MX layout check (fp4 and fp8, per-lane scales)
static float fp4_val(int c) {
    static const float t[8] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    return (c & 8) ? -t[c & 7] : t[c & 7];
}
static float fp8_val(int c) {   // OCP e4m3fn
    int s = c >> 7, e = (c >> 3) & 15, m = c & 7;
    float v = e == 0 ? ldexpf((float)m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -v : v;
}

static float bf8_val(int c) {   // OCP e5m2 (the top byte of an IEEE half)
    int s = c >> 7, e = (c >> 2) & 31, m = c & 3;
    float v = e == 0 ? ldexpf((float)m / 4.f, -14) : ldexpf(1.f + m / 4.f, e - 15);
    return s ? -v : v;
}

template <int FA, int FB>
__global__ void mx_one(const int* a, const int* b, const int* sa, const int* sb, float* d) {
    int l = threadIdx.x;
    i8v va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = a[l * 8 + i]; vb[i] = b[l * 8 + i]; }
    f4 c{0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, FA, FB, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = c[r];
}

template <int FA, int FB = FA>
static void layout_check() {
    std::vector<int> a(64 * 8, 0), b(64 * 8, 0), sa(64), sb(64);
    srand(7 + FA + 16 * FB);
    auto fix = [](int c, int f) {           // no NaN / inf codes; bf8 additionally inside a range whose products stay well inside fp32 (exponent field 8..22)
        if (f == 0 && (c & 0x7f) == 0x7f) c ^= 1;
        if (f == 1) c = (c & 0x83) | ((8 + ((c >> 2) & 31) % 15) << 2);
        return c;
    };
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < (FA == 4 ? 16 : 32); ++i) ((unsigned char*)&a[l * 8])[i] = fix(rand() & 255, FA);
        for (int i = 0; i < (FB == 4 ? 16 : 32); ++i) ((unsigned char*)&b[l * 8])[i] = fix(rand() & 255, FB);
        sa[l] = 125 + rand() % 5; sb[l] = 125 + rand() % 5;    // byte 0 of the scale register (opsel 0)
    }
    // layout (tools/mx_layout_probe.hip, round 6): lane l holds row (A) / column (B) l % 16.  fp4: K block l / 16 = 32 consecutive elements, byte j = elements 2j (low
    // nibble), 2j + 1, in the FIRST four registers of the operand.  8-bit formats: lane group j = l / 16 holds elements 16 j .. 16 j + 15 in registers 0-3 and
    // 64 + 16 j .. 64 + 16 j + 15 in registers 4-7 -- also when the other operand is fp4.  The E8M0 scale of K block kb (elements 32 kb .. 32 kb + 31) of row m is
    // byte 0 of lane m + 16 kb's scale register in EVERY format.
    auto elem = [&](const std::vector<int>& v, int m, int k, int f) {      // element k (0..127) of row / column m
        if (f == 4) {
            const unsigned char* p = (const unsigned char*)&v[(m + 16 * (k >> 5)) * 8];
            int byte = p[(k & 31) >> 1];
            return fp4_val((k & 1) ? byte >> 4 : byte & 15);
        }
        const unsigned char* p = (const unsigned char*)&v[(m + 16 * ((k & 63) >> 4)) * 8];
        const int c = p[(k & 15) + 16 * (k >> 6)];
        return f == 1 ? bf8_val(c) : fp8_val(c);
    };
    double ref[16][16];
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
        double acc = 0;
        for (int k = 0; k < 128; ++k) {
            const int kb = k >> 5;
            acc += ldexp(1.0, sa[kb * 16 + m] - 127) * ldexp(1.0, sb[kb * 16 + n] - 127) * elem(a, m, k, FA) * elem(b, n, k, FB);
        }
        ref[m][n] = acc;
    }
    int *da, *db, *dsa, *dsb; float* dd;
    CK(hipMalloc(&da, 64 * 32)); CK(hipMalloc(&db, 64 * 32)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 1024));
    CK(hipMemcpy(da, a.data(), 64 * 32, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 64 * 32, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((mx_one<FA, FB>), dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
    std::vector<float> d(256);
    CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        maxerr = fmax(maxerr, fabs(d[l * 4 + r] - ref[row][col]));
        maxref = fmax(maxref, fabs(ref[row][col]));
    }
    printf("MX layout check A fmt %d x B fmt %d: max |gpu - host| = %.3e (max |ref| %.3e) -> %s\n", FA, FB, maxerr, maxref,
           maxerr <= 1e-5 * maxref ? "OK" : maxerr <= 2e-4 * maxref ? "OK (layout; the 8-bit formats' adder tree keeps ~14 bits below the largest product of a 128-element dot: a wrong layout is O(1))" : "MISMATCH");
}

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* d_out; long long* d_ticks;
    CK(hipMalloc(&d_out, 256 * 2 * 4 * 256 * 4)); CK(hipMalloc(&d_ticks, 8));
    layout_check<4>();
    layout_check<0>();
    layout_check<1>();
    layout_check<4, 1>();          // fp4 weights x E5M2 activations (round 6)
    layout_check<1, 4>();
    for (int zero = 0; zero < 2; ++zero) {
        printf("== operands %s, %d iterations, 2 x 4 waves per CU\n", zero ? "ZERO" : "random", iters);
        run<0, 0>("16 x f16 16x16x32 per step (phased kernel mix)", iters, zero, d_out, d_ticks);
        run<1, 0>("8 x f16 32x32x16 per step", iters, zero, d_out, d_ticks);
        run<3, 0>("16 x MX fp8 16x16x128", iters / 4, zero, d_out, d_ticks);
        run<3, 2>("16 x MX fp6 16x16x128", iters / 4, zero, d_out, d_ticks);
        run<3, 4>("16 x MX fp4 16x16x128", iters / 4, zero, d_out, d_ticks);
        run<3, 1>("16 x MX bf8 16x16x128", iters / 4, zero, d_out, d_ticks);
        run<3, 4, 1>("16 x MX fp4 (A) x bf8 (B) 16x16x128", iters / 4, zero, d_out, d_ticks);
        run<2, 4, 1>("K=128: 64 f16 + 32 MX fp4 x bf8 (2.0 units)", iters / 4, zero, d_out, d_ticks);
        run<2, 0>("K=128: 64 f16 + 32 MX fp8 (2.0 units)", iters / 4, zero, d_out, d_ticks);
        run<2, 2>("K=128: 64 f16 + 32 MX fp6 (1.5 units)", iters / 4, zero, d_out, d_ticks);
        run<2, 4>("K=128: 64 f16 + 32 MX fp4 (1.5 units)", iters / 4, zero, d_out, d_ticks);
    }
    return 0;
}
