// Geometry of the direct 3-channel 3x3 convolutions (pw_mfma.hip: k_pw_fwd<.., C3>, k_pw_wgrad<.., C3>; c3_bn.hip): x is the 4-channel NHWC image
// [B,H,W,4]; the "patch row" of an output pixel is gathered on the fly in the order k = 16*ky + 4*slot + ch with slot 0, 1, 2 = kx 0, 1, 2 and
// slot 3 / channel 3 zero (K = 48).
#pragma once
#include <stdint.h>
struct C3Geom { int H, W, Ho, Wo, stride; uint32_t bytes; int tpr; uint32_t m_tpr, m_ho, m_howo, m_wo; };   // tpr = 32-pixel tiles per output row; m_* = floor(2^32 / d)
// n / d for n < 2^32 with m = floor(2^32 / d): the estimate is q or q - 1
__device__ __forceinline__ uint32_t udiv_m(uint32_t n, uint32_t d, uint32_t m) {
    uint32_t q = __umulhi(n, m);
    if (n - q * d >= d) ++q;
    return q;
}
