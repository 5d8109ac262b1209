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

// ---- the same tail THROUGH the level-0 aux head (FTC.aux0, 1x1 32 -> n_class, reference nets/tcct.py:994, :1041): when nothing else reads g0 (the
// feature-polarization loss is off: `feats` is not evaluated), logits0 = W3 g0 + b3 = (W3 Wc) [up(y) | skip] + (W3 c + b3): g0 and dg0 -- two more 452 MB
// tensors at the bench shape -- never exist, and the fused 64 -> 32 backward kernel is replaced by the small-N kernels of the aux heads.
#define TAIL_MAXC 8
__global__ void k_tail_compose3(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                const float* __restrict__ w3 /*[C][32]*/, const float* __restrict__ b3, int C, float* __restrict__ wcc /*[C][64]*/,
                                float* __restrict__ wa /*[C][32]*/, float* __restrict__ wb /*[C][32]*/, float* __restrict__ ccc /*[C]*/) {
    __shared__ float s1[32][33], s2[32][33], swc[32][65], sc[32], s3[TAIL_MAXC][33];
    const int t = threadIdx.x, i = t >> 5, j = t & 31;
    s1[i][j] = w1[t]; s2[i][j] = w2[t];
    if (t < C * 32) s3[t >> 5][t & 31] = w3[t];
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) a += s2[i][k] * s1[k][j];
    swc[i][j] = a; swc[i][32 + j] = a + s2[i][j];
    if (t < 32) {
        float v = b2[t];
#pragma unroll
        for (int k = 0; k < 32; ++k) v += s2[t][k] * b1[k];
        sc[t] = v;
    }
    __syncthreads();
    if (t < C * 64) {
        const int c = t >> 6, q = t & 63;
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) v += s3[c][k] * swc[k][q];
        wcc[c * 64 + q] = v;
        if (q < 32) wa[c * 32 + q] = v; else wb[c * 32 + q - 32] = v;
    }
    if (t < C) {
        float v = b3[t];
#pragma unroll
        for (int k = 0; k < 32; ++k) v += s3[t][k] * sc[k];
        ccc[t] = v;
    }
}
/* wcc fp32 [C][64] = W3 [W2 W1 | W2 W1 + W2] (and its halves wa, wb [C][32]), ccc [C] = W3 (W2 b1 + b2) + b3; C <= 8 classes */
extern "C" int tcct_tail_compose3(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int C,
                                  float* wcc, float* wa, float* wb, float* ccc, tcct_stream_t stream) {
    TCCT_CHECK(w1 && b1 && w2 && b2 && w3 && b3 && wcc && wa && wb && ccc, "tail_compose3: NULL argument");
    TCCT_CHECK(C >= 1 && C <= TAIL_MAXC, "tail_compose3: C=%d (1..%d)", C, TAIL_MAXC);
    hipLaunchKernelGGL(k_tail_compose3, dim3(1), dim3(1024), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, b3, C, wcc, wa, wb, ccc);
    TCCT_LAUNCH_OK();
}

__global__ void k_tail_compose3_bwd(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                    const float* __restrict__ w3, int C, const float* __restrict__ dwa /*[C][32]*/, const float* __restrict__ dwb /*[C][32]*/,
                                    const float* __restrict__ dccc /*[C]*/, float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                    float* __restrict__ db2, float* __restrict__ dw3, float* __restrict__ db3) {
    __shared__ float s1[32][33], s2[32][33], swc[32][65], sc[32], s3[TAIL_MAXC][33], sg[TAIL_MAXC][65], sgc[TAIL_MAXC], sdwc[32][65], sdc[32], sa[32][33], sb1[32];
    const int t = threadIdx.x, i = t >> 5, j = t & 31;
    s1[i][j] = w1[t]; s2[i][j] = w2[t];
    if (t < C * 32) { s3[t >> 5][t & 31] = w3[t]; sg[t >> 5][t & 31] = dwa[t]; sg[t >> 5][32 + (t & 31)] = dwb[t]; }
    if (t < C) sgc[t] = dccc[t];
    if (t < 32) sb1[t] = b1[t];
    __syncthreads();
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) a += s2[i][k] * s1[k][j];
    swc[i][j] = a; swc[i][32 + j] = a + s2[i][j];
    if (t < 32) {
        float v = b2[t];
#pragma unroll
        for (int k = 0; k < 32; ++k) v += s2[t][k] * sb1[k];
        sc[t] = v;
    }
    __syncthreads();
    // dW3 = dWcc Wc^T + dccc c^T  [C][32];  dWc = W3^T dWcc  [32][64];  dc = W3^T dccc
    if (t < C * 32) {
        const int c = t >> 5, k2 = t & 31;
        float v = sgc[c] * sc[k2];
#pragma unroll
        for (int q = 0; q < 64; ++q) v += sg[c][q] * swc[k2][q];
        dw3[t] = v;
    }
    if (t < C) db3[t] = sgc[t];
    {
        float v0 = 0.f, v1 = 0.f;
        for (int c = 0; c < C; ++c) { v0 += s3[c][i] * sg[c][j]; v1 += s3[c][i] * sg[c][32 + j]; }
        sdwc[i][j] = v0; sdwc[i][32 + j] = v1;
    }
    if (t < 32) {
        float v = 0.f;
        for (int c = 0; c < C; ++c) v += s3[c][t] * sgc[c];
        sdc[t] = v;
    }
    __syncthreads();
    // as k_tail_compose_bwd
    const float g2 = sdwc[i][32 + j];
    sa[i][j] = sdwc[i][j] + g2;
    __syncthreads();
    float x = g2 + sdc[i] * sb1[j], y = 0.f;
#pragma unroll
    for (int k = 0; k < 32; ++k) { x += sa[i][k] * s1[j][k]; y += s2[k][i] * sa[k][j]; }
    dw2[t] = x;
    dw1[t] = y;
    if (t < 32) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) v += s2[k][t] * sdc[k];
        db1[t] = v;
        db2[t] = sdc[t];
    }
}
/* gradients of the six tensors from those of the composed pair: dwa / dwb fp32 [C][32] (the two halves of d wcc), dccc [C]; all outputs overwritten */
extern "C" int tcct_tail_compose3_bwd(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, int C, const float* dwa,
                                      const float* dwb, const float* dccc, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3,
                                      tcct_stream_t stream) {
    TCCT_CHECK(w1 && b1 && w2 && b2 && w3 && dwa && dwb && dccc && dw1 && db1 && dw2 && db2 && dw3 && db3, "tail_compose3_bwd: NULL argument");
    TCCT_CHECK(C >= 1 && C <= TAIL_MAXC, "tail_compose3_bwd: C=%d (1..%d)", C, TAIL_MAXC);
    hipLaunchKernelGGL(k_tail_compose3_bwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, w1, b1, w2, b2, w3, C, dwa, dwb, dccc, dw1, db1, dw2, db2, dw3, db3);
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ levels 1-3: aux head composed through t32x
// Reference nets/tcct.py:1036-1044: g_i = t32x(x_i + y_i), logits_i = aux_i(g_i).  When the feature-polarization loss is off nothing but aux_i reads
// g_i, and both are 1x1 convolutions: logits_i = (Wa Wt) s_i + (Wa bt + ba) -- one 32 -> C GEMM on s_i with the composed weight; g_i and its gradient
// (113 MB each at level 1 of the bench shape) are never written.  Gradients: dWt = Wa^T dWh, dWa = dWh Wt^T + dch bt^T, dbt = Wa^T dch, dba = dch.
__global__ void k_head_compose(const float* __restrict__ wt, const float* __restrict__ bt, const float* __restrict__ wa, const float* __restrict__ ba,
                               int C, float* __restrict__ wh /*[C][32]*/, float* __restrict__ ch /*[C]*/) {
    __shared__ float st[32][33], sa[16][33];
    const int t = threadIdx.x, i = t >> 5, j = t & 31;
    st[i][j] = wt[t];
    if (i < C) sa[i][j] = wa[i * 32 + j];
    __syncthreads();
    if (i < C) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) a += sa[i][k] * st[k][j];
        wh[i * 32 + j] = a;
    }
    if (t < C) {
        float v = ba[t];
#pragma unroll
        for (int k = 0; k < 32; ++k) v += sa[t][k] * bt[k];
        ch[t] = v;
    }
}
/* wh fp32 [C][32] = Wa Wt, ch [C] = Wa bt + ba (wt [32][32], bt [32] = the t32x convolution; wa [C][32], ba [C] = the aux head; C <= 16) */
extern "C" int tcct_head_compose(const float* wt, const float* bt, const float* wa, const float* ba, int C, float* wh, float* ch, tcct_stream_t stream) {
    TCCT_CHECK(wt && bt && wa && ba && wh && ch, "head_compose: NULL argument");
    TCCT_CHECK(C >= 1 && C <= 16, "head_compose: C=%d unsupported (1..16)", C);
    hipLaunchKernelGGL(k_head_compose, dim3(1), dim3(1024), 0, (hipStream_t)stream, wt, bt, wa, ba, C, wh, ch);
    TCCT_LAUNCH_OK();
}

__global__ void k_head_compose_bwd(const float* __restrict__ wt, const float* __restrict__ bt, const float* __restrict__ wa, int C,
                                   const float* __restrict__ dwh /*[C][32]*/, const float* __restrict__ dch /*[C]*/, float* __restrict__ dwt,
                                   float* __restrict__ dbt, float* __restrict__ dwa, float* __restrict__ dba) {
    __shared__ float st[32][33], sa[16][33], sg[16][33], sc[16], sb[32];
    const int t = threadIdx.x, i = t >> 5, j = t & 31;
    st[i][j] = wt[t];
    if (i < C) { sa[i][j] = wa[i * 32 + j]; sg[i][j] = dwh[i * 32 + j]; }
    if (t < C) sc[t] = dch[t];
    if (t < 32) sb[t] = bt[t];
    __syncthreads();
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += sa[c][i] * sg[c][j];            // (Wa^T dWh)[i][j]
    dwt[t] = a;
    if (i < C) {
        float b = sc[i] * sb[j];
#pragma unroll
        for (int k = 0; k < 32; ++k) b += sg[i][k] * st[j][k];       // (dWh Wt^T)[i][j] = sum_k dWh[i][k] Wt[j][k]
        dwa[i * 32 + j] = b;
    }
    if (t < 32) {
        float v = 0.f;
        for (int c = 0; c < C; ++c) v += sa[c][t] * sc[c];
        dbt[t] = v;
    }
    if (t < C) dba[t] = sc[t];
}
/* the four gradients from those of the composed pair (all outputs overwritten) */
extern "C" int tcct_head_compose_bwd(const float* wt, const float* bt, const float* wa, int C, const float* dwh, const float* dch, float* dwt, float* dbt,
                                     float* dwa, float* dba, tcct_stream_t stream) {
    TCCT_CHECK(wt && bt && wa && dwh && dch && dwt && dbt && dwa && dba, "head_compose_bwd: NULL argument");
    TCCT_CHECK(C >= 1 && C <= 16, "head_compose_bwd: C=%d unsupported (1..16)", C);
    hipLaunchKernelGGL(k_head_compose_bwd, dim3(1), dim3(1024), 0, (hipStream_t)stream, wt, bt, wa, C, dwh, dch, dwt, dbt, dwa, dba);
    TCCT_LAUNCH_OK();
}
