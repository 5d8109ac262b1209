// RegNet.lap_reg as ONE kernel each way (round 5).  Reference nets/reg.py:66-74,115-116: pseu = |dw3x3(dw3x3(x; w1, b1); w2, b2)| on the [B, C-1, H, W] class planes (the
// logits without class 0, and the one-hot label planes), both depthwise 3 x 3 with zero padding and bias.  The op-by-op form was dw -> dw -> |.| (three fp32 passes of a
// 113 MB tensor at the bench shape, twice per step) and in the backward pass |.|' -> dw^T -> dw^T plus two weight-gradient passes.  Here a block stages a tile of x with its
// halo in LDS once and walks the stages in LDS:
//   forward  (halo 2): x -> y1 = b1 + w1 * x (ZERO outside the image: the second convolution pads ITS input with zeros) -> out = |b2 + w2 * y1|
//   backward (halo 4): x -> y1 -> z -> dz = dout sign(z) -> dy1 = w2^T * dz (zero outside the image) -> dx = w1^T * dy1, and from the tile's own pixels
//                      dW2 = sum dz (x) y1, db2 = sum dz, dW1 = sum dy1 (x) x, db1 = sum dy1: y1 and z are recomputed, nothing but x and dout is read.
// fp32 throughout (the loss side is fp32 in every mode), NHWC with C = 4 or 8 channels (5 or 9 classes), a thread owns one pixel and a float4 of channels.  Same
// sums as the three-kernel form up to the order of the fp32 additions inside a 3 x 3 window (tests/test_kernels_gpu.py compares against it).
#include "common.h"

#define LR_TW 32
#define LR_TH 8
#define LR_T (LR_TW * LR_TH)

struct LrW { float w1[9], b1, w2[9], b2; };          // per channel

__device__ __forceinline__ float4 lr_ld(const float* __restrict__ p, int N, int H, int W, int C, int n, int h, int w, int c4) {
    if (h < 0 || h >= H || w < 0 || w >= W) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(p + (((int64_t)n * H + h) * W + w) * C + c4 * 4);
}
__device__ __forceinline__ float lr_get(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ __forceinline__ void lr_set(float4& v, int k, float x) { if (k == 0) v.x = x; else if (k == 1) v.y = x; else if (k == 2) v.z = x; else v.w = x; }

// stage: dst (tile + halo hd) = bias + correlation of src (tile + halo hd + 1) with w[9] (FLIP: with the flipped taps, no bias: the transposed convolution);
// dst pixels outside the image are ZERO.  src / dst: LDS arrays of float4 with row pitches (LR_TW + 2 hs) / (LR_TW + 2 hd).
template <bool FLIP>
__device__ __forceinline__ void lr_stage(const float4* __restrict__ src, float4* __restrict__ dst, int hs, int hd, const float (&w)[4][9], const float (&b)[4],
                                         int h0, int w0, int H, int W, int tid) {
    const int pw = LR_TW + 2 * hd, ph = LR_TH + 2 * hd, ps = LR_TW + 2 * hs;
    for (int i = tid; i < pw * ph; i += LR_T) {
        const int r = i / pw, c = i - r * pw;
        const int ih = h0 - hd + r, iw = w0 - hd + c;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
            float a[4] = {FLIP ? 0.f : b[0], FLIP ? 0.f : b[1], FLIP ? 0.f : b[2], FLIP ? 0.f : b[3]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 v = src[(r + (hs - hd) + ky - 1) * ps + (c + (hs - hd) + kx - 1)];
                    const int t = FLIP ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx;
                    a[0] += w[0][t] * v.x; a[1] += w[1][t] * v.y; a[2] += w[2][t] * v.z; a[3] += w[3][t] * v.w;
                }
            o = make_float4(a[0], a[1], a[2], a[3]);
        }
        dst[i] = o;
    }
}

__device__ __forceinline__ void lr_weights(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                           int c4, float (&W1)[4][9], float (&B1)[4], float (&W2)[4][9], float (&B2)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c4 * 4 + k;
#pragma unroll
        for (int t = 0; t < 9; ++t) { W1[k][t] = w1[c * 9 + t]; W2[k][t] = w2[c * 9 + t]; }
        B1[k] = b1 ? b1[c] : 0.f; B2[k] = b2 ? b2[c] : 0.f;
    }
}

__global__ void __launch_bounds__(LR_T) k_lapreg_fwd(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ out,
                                                     int N, int H, int W, int C, int tilesH, int tilesW, int64_t ntiles) {
    __shared__ float4 sX[(LR_TH + 4) * (LR_TW + 4)];
    __shared__ float4 sY[(LR_TH + 2) * (LR_TW + 2)];
    const int tid = threadIdx.x, CV = C >> 2;
    for (int64_t tile = blockIdx.x; tile < ntiles * CV; tile += gridDim.x) {
        const int c4 = (int)(tile % CV);
        int64_t t = tile / CV;
        const int tw = (int)(t % tilesW); t /= tilesW;
        const int th = (int)(t % tilesH);
        const int n = (int)(t / tilesH);
        const int h0 = th * LR_TH, w0 = tw * LR_TW;
        float W1[4][9], B1[4], W2[4][9], B2[4];
        lr_weights(w1, b1, w2, b2, c4, W1, B1, W2, B2);
        __syncthreads();
        for (int i = tid; i < (LR_TH + 4) * (LR_TW + 4); i += LR_T) {
            const int r = i / (LR_TW + 4), c = i - r * (LR_TW + 4);
            sX[i] = lr_ld(x, N, H, W, C, n, h0 - 2 + r, w0 - 2 + c, c4);
        }
        __syncthreads();
        lr_stage<false>(sX, sY, 2, 1, W1, B1, h0, w0, H, W, tid);
        __syncthreads();
        const int r = tid / LR_TW, c = tid - r * LR_TW;
        const int oh = h0 + r, ow = w0 + c;
        if (oh < H && ow < W) {
            float a[4] = {B2[0], B2[1], B2[2], B2[3]};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 v = sY[(r + ky) * (LR_TW + 2) + (c + kx)];
                    const int tp = ky * 3 + kx;
                    a[0] += W2[0][tp] * v.x; a[1] += W2[1][tp] * v.y; a[2] += W2[2][tp] * v.z; a[3] += W2[3][tp] * v.w;
                }
            *reinterpret_cast<float4*>(out + (((int64_t)n * H + oh) * W + ow) * C + c4 * 4) = make_float4(fabsf(a[0]), fabsf(a[1]), fabsf(a[2]), fabsf(a[3]));
        }
    }
}

// sums[C][20]: per channel dW1[9], db1, dW2[9], db2 (zero on entry; fp32 atomics, one per value and block)
__global__ void __launch_bounds__(LR_T) k_lapreg_bwd(const float* __restrict__ x, const float* __restrict__ dout, const float* __restrict__ w1,
                                                     const float* __restrict__ b1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                     float* __restrict__ dx, float* __restrict__ sums, int N, int H, int W, int C, int tilesH, int tilesW,
                                                     int64_t ntiles) {
    __shared__ float4 sX[(LR_TH + 8) * (LR_TW + 8)];          // x, halo 4
    __shared__ float4 sY[(LR_TH + 6) * (LR_TW + 6)];          // y1, halo 3
    __shared__ float4 sZ[(LR_TH + 4) * (LR_TW + 4)];          // dz = dout sign(z), halo 2
    __shared__ float4 sD[(LR_TH + 2) * (LR_TW + 2)];          // dy1, halo 1
    __shared__ float sR[LR_T / 64][80];
    const int tid = threadIdx.x, CV = C >> 2, lane = tid & 63, wave = tid >> 6;
    int cur_c4 = -1;
    float acc[4][20];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 20; ++j) acc[k][j] = 0.f;
    auto flush = [&](int c4) {          // the block's partial sums of channel vector c4 -> wave sums -> one atomic per value
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                float v = acc[k][j];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) sR[wave][k * 20 + j] = v;
                acc[k][j] = 0.f;
            }
        __syncthreads();
        if (tid < 80) {
            float v = 0.f;
#pragma unroll
            for (int wv = 0; wv < LR_T / 64; ++wv) v += sR[wv][tid];
            if (v != 0.f) atomicAdd(&sums[(c4 * 4 + tid / 20) * 20 + tid % 20], v);
        }
        __syncthreads();
    };
    // a block walks tiles of ONE channel vector at a time (tile index = c4-major) so that its register partial sums are flushed once per channel vector
    const int64_t per = ntiles;
    for (int64_t tile = blockIdx.x; tile < per * CV; tile += gridDim.x) {
        const int c4 = (int)(tile / per);
        if (c4 != cur_c4) { if (cur_c4 >= 0) flush(cur_c4); cur_c4 = c4; }
        int64_t t = tile - (int64_t)c4 * per;
        const int tw = (int)(t % tilesW); t /= tilesW;
        const int th = (int)(t % tilesH);
        const int n = (int)(t / tilesH);
        const int h0 = th * LR_TH, w0 = tw * LR_TW;
        float W1[4][9], B1[4], W2[4][9], B2[4];
        lr_weights(w1, b1, w2, b2, c4, W1, B1, W2, B2);
        __syncthreads();
        for (int i = tid; i < (LR_TH + 8) * (LR_TW + 8); i += LR_T) {
            const int r = i / (LR_TW + 8), c = i - r * (LR_TW + 8);
            sX[i] = lr_ld(x, N, H, W, C, n, h0 - 4 + r, w0 - 4 + c, c4);
        }
        __syncthreads();
        lr_stage<false>(sX, sY, 4, 3, W1, B1, h0, w0, H, W, tid);           // y1 on halo 3
        __syncthreads();
        for (int i = tid; i < (LR_TH + 4) * (LR_TW + 4); i += LR_T) {       // z on halo 2 -> dz = dout sign(z)  (sign(0) = 0, as torch.abs' backward)
            const int r = i / (LR_TW + 4), c = i - r * (LR_TW + 4);
            const int ih = h0 - 2 + r, iw = w0 - 2 + c;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ih >= 0 && ih < H && iw >= 0 && iw < W) {
                float a[4] = {B2[0], B2[1], B2[2], B2[3]};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 v = sY[(r + ky) * (LR_TW + 6) + (c + kx)];
                        const int tp = ky * 3 + kx;
                        a[0] += W2[0][tp] * v.x; a[1] += W2[1][tp] * v.y; a[2] += W2[2][tp] * v.z; a[3] += W2[3][tp] * v.w;
                    }
                const float4 g = *reinterpret_cast<const float4*>(dout + (((int64_t)n * H + ih) * W + iw) * C + c4 * 4);
                o.x = a[0] > 0.f ? g.x : (a[0] < 0.f ? -g.x : 0.f); o.y = a[1] > 0.f ? g.y : (a[1] < 0.f ? -g.y : 0.f);
                o.z = a[2] > 0.f ? g.z : (a[2] < 0.f ? -g.z : 0.f); o.w = a[3] > 0.f ? g.w : (a[3] < 0.f ? -g.w : 0.f);
            }
            sZ[i] = o;
        }
        __syncthreads();
        lr_stage<true>(sZ, sD, 2, 1, W2, B2, h0, w0, H, W, tid);            // dy1 = w2^T * dz on halo 1 (zero outside the image)
        __syncthreads();
        const int r = tid / LR_TW, c = tid - r * LR_TW;
        const int oh = h0 + r, ow = w0 + c;
        if (oh < H && ow < W) {
            if (dx) {
                float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float4 v = sD[(r + ky) * (LR_TW + 2) + (c + kx)];
                        const int tp = (2 - ky) * 3 + (2 - kx);
                        a[0] += W1[0][tp] * v.x; a[1] += W1[1][tp] * v.y; a[2] += W1[2][tp] * v.z; a[3] += W1[3][tp] * v.w;
                    }
                *reinterpret_cast<float4*>(dx + (((int64_t)n * H + oh) * W + ow) * C + c4 * 4) = make_float4(a[0], a[1], a[2], a[3]);
            }
            // this pixel's share of the weight gradients: dz (x) y1 window, dy1 (x) x window
            const float4 dz = sZ[(r + 2) * (LR_TW + 4) + (c + 2)], dy = sD[(r + 1) * (LR_TW + 2) + (c + 1)];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float4 yv = sY[(r + 3 + ky - 1) * (LR_TW + 6) + (c + 3 + kx - 1)];
                    const float4 xv = sX[(r + 4 + ky - 1) * (LR_TW + 8) + (c + 4 + kx - 1)];
                    const int tp = ky * 3 + kx;
                    acc[0][tp] += dy.x * xv.x; acc[1][tp] += dy.y * xv.y; acc[2][tp] += dy.z * xv.z; acc[3][tp] += dy.w * xv.w;
                    acc[0][10 + tp] += dz.x * yv.x; acc[1][10 + tp] += dz.y * yv.y; acc[2][10 + tp] += dz.z * yv.z; acc[3][10 + tp] += dz.w * yv.w;
                }
            acc[0][9] += dy.x; acc[1][9] += dy.y; acc[2][9] += dy.z; acc[3][9] += dy.w;
            acc[0][19] += dz.x; acc[1][19] += dz.y; acc[2][19] += dz.z; acc[3][19] += dz.w;
        }
    }
    if (cur_c4 >= 0) flush(cur_c4);
}

/* out = |dw3x3(dw3x3(x; w1, b1); w2, b2)|, fp32 NHWC [N,H,W,C], C = 4 or 8 (reference nets/reg.py:66-74,115-116: lap_reg on the class planes); w*: [C,1,3,3], b*: [C] */
extern "C" int tcct_lapreg_fwd(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* out, int N, int H, int W, int C,
                               tcct_stream_t stream) {
    TCCT_CHECK(C == 4 || C == 8, "lapreg_fwd: C=%d (4 or 8)", C);
    TCCT_CHECK(N > 0 && H > 0 && W > 0, "lapreg_fwd: empty tensor");
    const int tilesH = (H + LR_TH - 1) / LR_TH, tilesW = (W + LR_TW - 1) / LR_TW;
    const int64_t ntiles = (int64_t)N * tilesH * tilesW, work = ntiles * (C / 4);
    const int grid = (int)(work < 8192 ? work : 8192);
    hipLaunchKernelGGL(k_lapreg_fwd, dim3(grid), dim3(LR_T), 0, (hipStream_t)stream, x, w1, b1, w2, b2, out, N, H, W, C, tilesH, tilesW, ntiles);
    TCCT_LAUNCH_OK();
}

/* backward of tcct_lapreg_fwd from x and dout alone (y1 and z are recomputed in LDS): dx (NULL: not wanted -- the label planes) and
 * sums fp32 [C][20] = per channel {dW1[9], db1, dW2[9], db2} (cleared here unless the caller pre-zeroed its accumulation outputs) */
extern "C" int tcct_lapreg_bwd(const float* x, const float* dout, const float* w1, const float* b1, const float* w2, const float* b2, float* dx, float* sums,
                               int N, int H, int W, int C, tcct_stream_t stream) {
    TCCT_CHECK(C == 4 || C == 8, "lapreg_bwd: C=%d (4 or 8)", C);
    TCCT_CHECK(N > 0 && H > 0 && W > 0, "lapreg_bwd: empty tensor");
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(sums, 0, sizeof(float) * C * 20, st) != hipSuccess) { tcct_set_error("lapreg_bwd: memset failed"); return -2; }
    const int tilesH = (H + LR_TH - 1) / LR_TH, tilesW = (W + LR_TW - 1) / LR_TW;
    const int64_t ntiles = (int64_t)N * tilesH * tilesW, work = ntiles * (C / 4);
    const int grid = (int)(work < 2048 ? work : 2048);
    hipLaunchKernelGGL(k_lapreg_bwd, dim3(grid), dim3(LR_T), 0, st, x, dout, w1, b1, w2, b2, dx, sums, N, H, W, C, tilesH, tilesW, ntiles);
    TCCT_LAUNCH_OK();
}
