// Scalar pieces of the MX plane-set quantiser shared by ev_gemm.hip (conv-GEMM epilogues, mx_planes_kernel) and ev_misc.hip (the LayerNorm that writes
// its consumer's plane set).  Format: ev_gemm_mx.h / emotivoice_amd/mxfp4.py.
#pragma once
#include <hip/hip_runtime.h>

namespace ev {

// E8M0 byte of the block scale 2^(floor(log2 amax) - 2), clamped to [1, 254] (an all-zero block gets 1)
__device__ __forceinline__ unsigned mx_scale_byte(float amax) {
    const int e = (int)((__float_as_uint(amax) >> 23) & 0xffu) - 2;
    return (unsigned)min(max(e, 1), 254);
}
// fp4 (e2m1) code of an already scaled value: |y| <= 6 after scaling (larger saturates), round to nearest even
__device__ __forceinline__ unsigned mx_fp4_code(float y) {
    const float a = fabsf(y);
    unsigned c = (a > 0.25f) + (a >= 0.75f) + (a > 1.25f) + (a >= 1.75f) + (a > 2.5f) + (a >= 3.5f) + (a > 5.0f);
    return c | ((__float_as_uint(y) >> 28) & 8u);
}

}  // namespace ev
