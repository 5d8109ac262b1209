// The last decoder block's tail as ONE GEMM (round 4).  Reference nets/tcct.py:908-914 (MPUpBlock.forward), :1031 (`x_0 + y_0`) and :1035-1040 (`t324`):
//     u  = up(y) + skip                    (bilinear x2, align_corners=True; skip = the CNN's level-0 output)
//     d0 = W1 u + b1                       (MPUpBlock.post, 1x1, 32 -> 32)
//     s0 = d0 + skip
//     g0 = W2 s0 + b2                      (FTC.t324, 1x1, 32 -> 32)
// d0 is read by nothing else at level 0 and s0 only by t324, and there is no nonlinearity between the three steps:
//     g0 = A v + (A + B) skip + c          with v = up(y), A = W2 W1, B = W2, c = W2 b1 + b2
// i.e. a 64 -> 32 pointwise convolution over the (never materialised) concatenation [v | skip] with the composed weight Wc = [A | A + B]
// (tcct_pw_fwd_cat2 / tcct_pw_bwd_cat2_bias).  u, d0 and s0 -- three 452 MB tensors at the bench shape -- are never written; the backward pass is one
// kernel (dv, dskip, dWc, dc from one pass over dg0) instead of two GEMM backward kernels, and this file turns dWc, dc into dW1, db1, dW2, db2:
//     dA = dWc[:, :32] + dWc[:, 32:],  dB = dWc[:, 32:]
//     dW2 = dA W1^T + dB + dc b1^T,    dW1 = W2^T dA,    db1 = W2^T dc,    db2 = dc
// All 32 x 32: one block each way.
#include "common.h"

__global__ void k_tail_compose(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                               float* __restrict__ wc /*[32][64]*/, float* __restrict__ c /*[32]*/) {
    __shared__ float s1[32][33], s2[32][33];
    const int t = threadIdx.x, i = t >> 5, j = t & 31;
    s1[i][j] = w1[t]; s2[i][j] = w2[t];
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) a += s2[i][k] * s1[k][j];          // A[i][j] = sum_k W2[i][k] W1[k][j]
    wc[i * 64 + j] = a;
    wc[i * 64 + 32 + j] = a + s2[i][j];
    if (t < 32) {
        float v = b2[t];
#pragma unroll
        for (int k = 0; k < 32; ++k) v += s2[t][k] * b1[k];
        c[t] = v;
    }
}
/* Wc fp32 [32][64] = [W2 W1 | W2 W1 + W2], c fp32 [32] = W2 b1 + b2 from the two 1x1 convolutions (weights [32][32] as stored by nn.Conv2d, biases [32]) */
extern "C" int tcct_tail_compose(const float* w1, const float* b1, const float* w2, const float* b2, float* wc, float* c, tcct_stream_t stream) {
    TCCT_CHECK(w1 && b1 && w2 && b2 && wc && c, "tail_compose: NULL argument");
    hipLaunchKernelGGL(k_tail_compose, dim3(1), dim3(1024), 0, (hipStream_t)stream, w1, b1, w2, b2, wc, c);
    TCCT_LAUNCH_OK();
}

__global__ void k_tail_compose_bwd(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                   const float* __restrict__ dwc /*[32][64]*/, const float* __restrict__ dc /*[32]*/, float* __restrict__ dw1,
                                   float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2) {
    __shared__ float s1[32][33], s2[32][33], sa[32][33], sc[32], sb1[32];
    const int t = threadIdx.x, i = t >> 5, j = t & 31;
    s1[i][j] = w1[t]; s2[i][j] = w2[t];
    const float g1 = dwc[i * 64 + j], g2 = dwc[i * 64 + 32 + j];
    sa[i][j] = g1 + g2;                                              // dA
    if (t < 32) { sc[t] = dc[t]; sb1[t] = b1[t]; }
    __syncthreads();
    float a = g2 + sc[i] * sb1[j], b = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        a += sa[i][k] * s1[j][k];                                    // (dA W1^T)[i][j] = sum_k dA[i][k] W1[j][k]
        b += s2[k][i] * sa[k][j];                                    // (W2^T dA)[i][j] = sum_k W2[k][i] dA[k][j]
    }
    dw2[t] = a;
    dw1[t] = b;
    if (t < 32) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) v += s2[k][t] * sc[k];          // (W2^T dc)[t]
        db1[t] = v;
        db2[t] = sc[t];
    }
}
/* gradients of the four tensors from those of the composed pair: dwc fp32 [32][64], dc fp32 [32] -> dw1, dw2 [32][32], db1, db2 [32] (all overwritten) */
extern "C" int tcct_tail_compose_bwd(const float* w1, const float* b1, const float* w2, const float* dwc, const float* dc, float* dw1, float* db1,
                                     float* dw2, float* db2, tcct_stream_t stream) {
    TCCT_CHECK(w1 && b1 && w2 && dwc && dc && dw1 && db1 && dw2 && db2, "tail_compose_bwd: NULL argument");
    hipLaunchKernelGGL(k_tail_compose_bwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, w1, b1, w2, dwc, dc, dw1, db1, dw2, db2);
    TCCT_LAUNCH_OK();
}
