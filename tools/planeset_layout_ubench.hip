// does ONE interleaved row record per plane-set tensor move bytes faster than the five arrays of mx_layout()?
//
// An elementwise read-modify-write in the conv-GEMM's tile geometry -- one 512-thread block per 256-row tile, 80 KB of LDS per block so that exactly
// two blocks are resident per CU (512 tiles in flight), all of a tile's loads issued before its first store -- over
//   (i)  the five arrays of a C-channel plane set: h [rows][2C], q4[0] / q4[1] [rows][C/2], qs[0] / qs[1] chunk-major [C/128][rows][4]   (ev_gemm_mx.h, mx_layout)
//   (ii) one record per row: [ h 2C | q4[0] C/2 | q4[1] C/2 | qs[0] C/32 | qs[1] C/32 | pad to 16 B ]
// and, as the yardstick, (iii) the same bytes as one flat array streamed by a grid-stride kernel that fills the chip (what "an elementwise kernel
// reaches" means in DESIGN.md: 6.2 TB/s).  Decision rule of the verdict: convert the generator only if (ii) >= 1.25 x (i).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/planeset_layout_ubench tools/planeset_layout_ubench.hip && tools/build/planeset_layout_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
static constexpr int BM = 256;

struct Planes { char *h, *q0, *q1, *s0, *s1; unsigned s_stride; };

__device__ __forceinline__ u32x4 mix(u32x4 v) { return v ^ u32x4{0x01010101u, 0x02020202u, 0x04040404u, 0x08080808u}; }

// (i): five arrays.  Piece counts per tile at C channels: h 256 * 2C / 16, q 256 * (C/2) / 16 each, scales 256 * 4 / 16 per 128-channel chunk and plane.
template <int C>
__global__ __launch_bounds__(512, 2) void rmw_planes(Planes in, Planes out, int ntiles) {
    extern __shared__ char lds_[];
    (void)lds_;
    const int tile = blockIdx.x, tid = threadIdx.x;
    const long r0 = (long)tile * BM;
    constexpr int NH = BM * 2 * C / 16 / 512, NQ = BM * (C / 2) / 16 / 512 > 0 ? BM * (C / 2) / 16 / 512 : 1, NS = C / 128;
    constexpr bool QPART = BM * (C / 2) / 16 < 512;       // C = 64: a code plane of a tile is 8 KB = one 16-byte piece for every thread; C = 32: half the threads
    u32x4 vh[NH], vq0[NQ], vq1[NQ], vs0[NS], vs1[NS];
    const char* hp = in.h + r0 * 2 * C;
#pragma unroll
    for (int i = 0; i < NH; ++i) vh[i] = *reinterpret_cast<const u32x4*>(hp + (long)(i * 512 + tid) * 16);
    const bool qact = !QPART || tid < BM * (C / 2) / 16;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (qact) {
            vq0[i] = *reinterpret_cast<const u32x4*>(in.q0 + r0 * (C / 2) + (long)(i * 512 + tid) * 16);
            vq1[i] = *reinterpret_cast<const u32x4*>(in.q1 + r0 * (C / 2) + (long)(i * 512 + tid) * 16);
        }
    }
    const bool sact = tid < BM * 4 / 16;
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        if (sact) {
            vs0[c] = *reinterpret_cast<const u32x4*>(in.s0 + (long)c * in.s_stride + r0 * 4 + tid * 16);
            vs1[c] = *reinterpret_cast<const u32x4*>(in.s1 + (long)c * in.s_stride + r0 * 4 + tid * 16);
        }
    }
    char* ho = out.h + r0 * 2 * C;
#pragma unroll
    for (int i = 0; i < NH; ++i) *reinterpret_cast<u32x4*>(ho + (long)(i * 512 + tid) * 16) = mix(vh[i]);
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (qact) {
            *reinterpret_cast<u32x4*>(out.q0 + r0 * (C / 2) + (long)(i * 512 + tid) * 16) = mix(vq0[i]);
            *reinterpret_cast<u32x4*>(out.q1 + r0 * (C / 2) + (long)(i * 512 + tid) * 16) = mix(vq1[i]);
        }
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        if (sact) {
            *reinterpret_cast<u32x4*>(out.s0 + (long)c * out.s_stride + r0 * 4 + tid * 16) = mix(vs0[c]);
            *reinterpret_cast<u32x4*>(out.s1 + (long)c * out.s_stride + r0 * 4 + tid * 16) = mix(vs1[c]);
        }
    }
}

// (ii): one record of REC bytes per row; a tile is BM * REC contiguous bytes
template <int REC>
__global__ __launch_bounds__(512, 2) void rmw_records(const char* in, char* out, int ntiles) {
    extern __shared__ char lds_[];
    (void)lds_;
    const int tile = blockIdx.x, tid = threadIdx.x;
    constexpr int PIECES = BM * REC / 16, NP = (PIECES + 511) / 512;
    const char* ip = in + (long)tile * BM * REC;
    char* op = out + (long)tile * BM * REC;
    u32x4 v[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i)
        if (i * 512 + tid < PIECES) v[i] = *reinterpret_cast<const u32x4*>(ip + (long)(i * 512 + tid) * 16);
#pragma unroll
    for (int i = 0; i < NP; ++i)
        if (i * 512 + tid < PIECES) *reinterpret_cast<u32x4*>(op + (long)(i * 512 + tid) * 16) = mix(v[i]);
}

// (iii): yardstick -- the same byte count as one flat array, grid-stride, 8 pieces in flight per thread
__global__ __launch_bounds__(256) void rmw_flat(const char* in, char* out, long pieces) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 7 * stride < pieces; i += 8 * stride) {
        u32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const u32x4*>(in + (i + k * stride) * 16);
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<u32x4*>(out + (i + k * stride) * 16) = mix(v[k]);
    }
    for (; i < pieces; i += stride) *reinterpret_cast<u32x4*>(out + i * 16) = mix(*reinterpret_cast<const u32x4*>(in + i * 16));
}

static size_t al(size_t v) { return (v + 255) & ~(size_t)255; }

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; ++i) launch();
    CHK(hipEventRecord(b, 0));
    CHK(hipEventSynchronize(b));
    float ms;
    CHK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

template <int C, int REC>
static void run(long rows, const char* what) {
    const int ntiles = (int)(rows / BM);
    const size_t bh = al((size_t)rows * 2 * C), bq = al((size_t)rows * C / 2), bs = al((size_t)(C / 128 > 0 ? C / 128 : 1) * rows * 4);
    size_t total = bh + 2 * bq + 2 * bs;
    if (total < (size_t)rows * REC + 4096) total = (size_t)rows * REC + 4096;          // the padded records are the larger footprint
    char *din, *dout;
    CHK(hipMalloc(&din, total)); CHK(hipMalloc(&dout, total));
    CHK(hipMemset(din, 1, total)); CHK(hipMemset(dout, 0, total));
    auto mk = [&](char* base) { Planes p; p.h = base; p.q0 = base + bh; p.q1 = p.q0 + bq; p.s0 = p.q1 + bq; p.s1 = p.s0 + bs; p.s_stride = (unsigned)(rows * 4); return p; };
    const Planes pi = mk(din), po = mk(dout);
    const int lds = 80 * 1024;
    CHK(hipFuncSetAttribute((const void*)rmw_planes<(C >= 128 ? C : 128)>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHK(hipFuncSetAttribute((const void*)rmw_records<REC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const double bytes_planes = 2.0 * rows * (2.0 * C + C + (C / 32) * 2.0);       // read + write, algorithmic (scales: one byte per 32 channels and plane)
    const double bytes_rec = 2.0 * rows * REC;
    double t1 = 0, t2 = 0, t3 = 0;
    if constexpr (C >= 128) t1 = time_ms([&] { hipLaunchKernelGGL(rmw_planes<(C >= 128 ? C : 128)>, dim3(ntiles), dim3(512), lds, 0, pi, po, ntiles); }, 20);
    t2 = time_ms([&] { hipLaunchKernelGGL(rmw_records<REC>, dim3(ntiles), dim3(512), lds, 0, (const char*)din, dout, ntiles); }, 20);
    const long pieces = (long)((size_t)rows * REC / 16);
    t3 = time_ms([&] { hipLaunchKernelGGL(rmw_flat, dim3(256 * 16), dim3(256), 0, 0, (const char*)din, dout, pieces); }, 20);
    printf("%-28s rows %8ld  C %3d  planes (i): %7.3f ms = %5.2f TB/s   records of %3d B (ii): %7.3f ms = %5.2f TB/s (%.2f TB/s of plane bytes)   flat (iii): %7.3f ms = %5.2f TB/s   (ii)/(i) on plane bytes: %.3f\n",
           what, rows, C, t1, t1 > 0 ? bytes_planes / t1 * 1e-9 : 0.0, REC, t2, bytes_rec / t2 * 1e-9, bytes_planes / t2 * 1e-9, t3, bytes_rec / t3 * 1e-9,
           t1 > 0 ? t1 / t2 : 0.0);
    CHK(hipFree(din)); CHK(hipFree(dout));
}

int main() {
    // stage 1 (C = 128, 32 x 1024 frames x 64 = 2.1 M rows) and stage 0 (C = 256, 264 k rows): the k = 3 launches the verdict names
    run<128, 400>(2113536, "stage 1, record 392 -> 400");
    run<128, 448>(2113536, "stage 1, record 392 -> 448");
    run<256, 784>(264192, "stage 0, record 784");
    run<256, 784>(2113536, "C = 256 at stage-1 rows");
    return 0;
}
