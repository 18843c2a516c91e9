// Which K element does byte p of lane group j of an 8-BIT operand of v_mfma_scale_f32_16x16x128_f8f6f4 hold when the other operand is fp4?
// (round 6: fp4 weights x E5M2 activations.)  The fp4 layout is known and checked (tools/mfma_ubench.hip: lane l = row l & 15, K block l >> 4, nibble e = element
// 32 (l >> 4) + e).  Probe: A = fp4 one-hot at K element ka (every row), B = bf8 one-hot (1.0) at (lane group jb, byte pb) of every column; the product is
// non-zero iff the two positions are the same K element.  Prints the map  (jb, pb) -> k  and whether it matches the closed form this library then uses.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/build/mx_layout_probe tools/mx_layout_probe.hip && tools/build/mx_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int i8v __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int FB>
__global__ void probe(int* kmap, int* nhits) {
    const int l = threadIdx.x, pos = blockIdx.x, jb = pos >> 5, pb = pos & 31;      // B one-hot position
    i8v vb = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((l >> 4) == jb) vb[pb >> 2] = (FB == 1 ? 0x3C : 0x38) << (8 * (pb & 3));    // 1.0 in e5m2 / e4m3
    int found = -1, hits = 0;
    for (int ka = 0; ka < 128; ++ka) {
        i8v va = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((l >> 4) == (ka >> 5)) va[(ka & 31) >> 3] = 0x2 << (4 * (ka & 7));     // fp4 1.0 at element ka of the row
        f4 c = {0.f, 0.f, 0.f, 0.f};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(va, vb, c, 4, FB, 0, 127, 0, 127);
        if (c[0] != 0.f) { found = ka; ++hits; }
    }
    if (l == 0) { kmap[pos] = found; nhits[pos] = hits; }
}

template <int FB>
static void run(const char* name) {
    int *dk, *dn;
    CK(hipMalloc(&dk, 512)); CK(hipMalloc(&dn, 512));
    hipLaunchKernelGGL(probe<FB>, dim3(128), dim3(64), 0, 0, dk, dn);
    std::vector<int> k(128), n(128);
    CK(hipMemcpy(k.data(), dk, 512, hipMemcpyDeviceToHost)); CK(hipMemcpy(n.data(), dn, 512, hipMemcpyDeviceToHost));
    bool natural = true, split16 = true, one = true;
    for (int p = 0; p < 128; ++p) {
        const int j = p >> 5, b = p & 31;
        one &= n[p] == 1;
        natural &= k[p] == 32 * j + b;
        split16 &= k[p] == (b < 16 ? 16 * j + b : 64 + 16 * j + (b - 16));
    }
    printf("%s B operand against an fp4 A operand: one K element per byte: %s; natural (k = 32 j + byte): %s; split (bytes 0-15: k = 16 j + byte, bytes 16-31: k = 64 + 16 j + byte - 16): %s\n",
           name, one ? "yes" : "NO", natural ? "YES" : "no", split16 ? "YES" : "no");
    for (int j = 0; j < 4; ++j) {
        printf("  lane group %d:", j);
        for (int b = 0; b < 32; ++b) printf(" %d", k[j * 32 + b]);
        printf("\n");
    }
}

int main() {
    run<1>("bf8 (e5m2)");
    run<0>("fp8 (e4m3)");
    return 0;
}
