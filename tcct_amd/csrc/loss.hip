// Loss-side kernels: fused softmax + batch-global Dice sums (+ backward), the column-wise (over H) pieces of the
// boundary-regression loss, and small fp32 helpers.  Labels are class indices (uint8), never one-hot.
#include "common.h"

#define LB 256
#define MAXC 8

// ----------------------------------------------------------------------- softmax + Dice sums (kite/losses/loss.py:28-32,83-99)
// 1024-thread blocks on <= 512 blocks: the fp64 atomics at the end of every block serialise per address (see norm.hip)
#define DSB 1024
template <typename T>
__global__ void __launch_bounds__(DSB) k_dice_sums(const T* __restrict__ logits, const uint8_t* __restrict__ lab, int64_t M, int C,
                            double* __restrict__ sums /*[3][C]: I, P, G*/) {
    __shared__ float sm[3 * MAXC][DSB / 64];
    float I[MAXC], P[MAXC], G[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) I[c] = P[c] = G[c] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        float z[MAXC], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { z[c] = c < C ? ldf(logits + i * C + c) : -INFINITY; mx = fmaxf(mx, z[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { z[c] = c < C ? __expf(z[c] - mx) : 0.f; s += z[c]; }
        float inv = 1.f / s;
        int l = lab[i];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            float p = z[c] * inv;
            P[c] += p;
            if (c == l) { I[c] += p; G[c] += 1.f; }
        }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        float a = wave_sum(I[c]), b = wave_sum(P[c]), g = wave_sum(G[c]);
        if (lane == 0) { sm[c][w] = a; sm[MAXC + c][w] = b; sm[2 * MAXC + c][w] = g; }
    }
    __syncthreads();
    if (threadIdx.x < 3 * MAXC) {
        int q = threadIdx.x / MAXC, c = threadIdx.x % MAXC;
        if (c < C) {
            double a = 0.0;
            for (int k = 0; k < DSB / 64; ++k) a += (double)sm[threadIdx.x][k];
            atomicAdd(&sums[q * C + c], a);
        }
    }
}
__global__ void k_dice_finalize(const double* __restrict__ sums, int C, float* __restrict__ loss) {
    if (threadIdx.x == 0) {
        double l = 0.0;
        for (int c = 0; c < C; ++c) l += 1.0 - (1.0 + 2.0 * sums[c]) / (1.0 + sums[C + c] + sums[2 * C + c]);
        *loss = (float)l;
    }
}
extern "C" int tcct_softmax_dice_fwd(const void* logits, const uint8_t* labels, int64_t M, int C, double* sums, float* loss,
                                     int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C >= 2 && C <= MAXC, "softmax_dice_fwd: C=%d unsupported (2..%d)", C, MAXC);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, sizeof(double) * 3 * C, st) != hipSuccess) { tcct_set_error("softmax_dice_fwd: memset failed"); return -2; }
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_dice_sums<T>, dim3(tcct_grid(M, DSB, 512)), dim3(DSB), 0, st, (const T*)logits, labels, M, C, sums));
    hipLaunchKernelGGL(k_dice_finalize, dim3(1), dim3(64), 0, st, sums, C, loss);
    TCCT_LAUNCH_OK();
}

template <typename T>
__global__ void k_dice_bwd(const T* __restrict__ logits, const uint8_t* __restrict__ lab, int64_t M, int C,
                           const double* __restrict__ sums, const float* __restrict__ gout, float gscale,
                           T* __restrict__ dlogits) {
    float a[MAXC], b[MAXC];
    const float gs = gscale * (gout ? *gout : 1.f);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < C) {
            double U = 1.0 + sums[C + c] + sums[2 * C + c];
            a[c] = (float)(-2.0 / U);
            b[c] = (float)((1.0 + 2.0 * sums[c]) / (U * U));
        } else a[c] = b[c] = 0.f;
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        float z[MAXC], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { z[c] = c < C ? ldf(logits + i * C + c) : -INFINITY; mx = fmaxf(mx, z[c]); }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { z[c] = c < C ? __expf(z[c] - mx) : 0.f; s += z[c]; }
        float inv = 1.f / s;
        int l = lab[i];
        float dp[MAXC], dot = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { z[c] *= inv; dp[c] = b[c] + (c == l ? a[c] : 0.f); dot += z[c] * dp[c]; }
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) stf(dlogits + i * C + c, gs * z[c] * (dp[c] - dot));
    }
}
extern "C" int tcct_softmax_dice_bwd(const void* logits, const uint8_t* labels, int64_t M, int C, const double* sums,
                                     const float* grad_out, float grad_scale, void* dlogits, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C >= 2 && C <= MAXC, "softmax_dice_bwd: C=%d unsupported", C);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_dice_bwd<T>, dim3(tcct_grid(M, LB, 1 << 16)), dim3(LB), 0, (hipStream_t)stream, (const T*)logits, labels, M, C, sums, grad_out, grad_scale, (T*)dlogits));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------- deep-supervision heads: resize + softmax + Dice without the full-size logits
// The aux heads (FTC.forward, nets/tcct.py:1042-1044) are 5-channel maps at 1/2, 1/4, 1/8 resolution, resized to the input size
// (F.interpolate bilinear, align_corners=False) and fed to the Dice criterion.  Materialised, each costs a 141 MB fp32 tensor written
// and read in the forward and again (gradient) in the backward.  Here the interpolated logits are recomputed per pixel from the
// low-resolution map (L2-resident), bit-identical to k_bilinear_fwd's arithmetic:
//   forward : thread per full-resolution pixel -> softmax -> the three Dice sums;
//   backward: dL/dlow = R_h^T R_w^T G with G the per-pixel softmax-Dice gradient: pass 1 evaluates G for the 2S pixels of a row that
//             touch low-res column j and applies the column weights (T [B,H,w,C], <= 70 MB), pass 2 applies the row weights.
// Integer scale S = H/h = W/w only (2, 4, 8 on the path).
__device__ __forceinline__ void updice_logits(const float* __restrict__ r0, const float* __restrict__ r1, const Lerp& a, const Lerp& b,
                                              int C, float (&z)[MAXC]) {
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        z[c] = c < C ? a.l0 * (b.l0 * r0[b.i0 * C + c] + b.l1 * r0[b.i1 * C + c]) + a.l1 * (b.l0 * r1[b.i0 * C + c] + b.l1 * r1[b.i1 * C + c])
                     : -INFINITY;
}
__device__ __forceinline__ void softmax_inplace(float (&z)[MAXC], int C) {
    float mx = -INFINITY, s = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) mx = fmaxf(mx, z[c]);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) { z[c] = c < C ? __expf(z[c] - mx) : 0.f; s += z[c]; }
    const float inv = 1.f / s;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) z[c] *= inv;
}
#define UDB 1024     // <= 512 blocks (the fp64 atomics of the tail serialise per address), so large blocks for occupancy
// A thread owns low-res column j of one full-resolution row: the S pixels p = S j + k (forward) or the 2S pixels S j - S/2 + k that touch
// column j (backward) only need the low-res columns j-1, j, j+1, interpolated once along H (R[3][C]); the column weights depend on k
// alone: k < S/2 -> columns (j-1, j) with l1 = (k + .5)/S + .5, else (j, j+1) with l1 = (k + .5)/S - .5.  Clamped borders fall out of
// loading clamped columns (both taps equal the border column and the weights sum to 1).
template <int S>
__device__ __forceinline__ void updice_rows(const float* __restrict__ low, int n, int h, int w, int C, const Lerp& a, int j, float (&R)[3][MAXC]) {
    const float* r0 = low + ((int64_t)n * h + a.i0) * w * C;
    const float* r1 = low + ((int64_t)n * h + a.i1) * w * C;
    const int col[3] = {max(j - 1, 0), j, min(j + 1, w - 1)};
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) R[q][c] = c < C ? a.l0 * r0[col[q] * C + c] + a.l1 * r1[col[q] * C + c] : -INFINITY;
}
template <int S>
__device__ __forceinline__ void updice_pixel(const float (&R)[3][MAXC], int kk /* 0..2S-1: pixel S j - S/2 + kk */, int C, float (&z)[MAXC]) {
    // kk < S: taps (j-1, j); else (j, j+1).  position inside its low-res cell: t = ((kk + S/2) mod S + .5)/S
    const int k = (kk + S / 2) % S;
    const float f = ((float)k + 0.5f) / (float)S;
    const float l1 = k < S / 2 ? f + 0.5f : f - 0.5f, l0 = 1.f - l1;
    const int q = kk < S ? 0 : 1;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) z[c] = c < C ? l0 * R[q][c] + l1 * R[q + 1][c] : -INFINITY;
}
template <int S>
__global__ void __launch_bounds__(UDB) k_updice_sums(const float* __restrict__ low, const uint8_t* __restrict__ lab, int B, int h, int w,
                                                     int H, int W, int C, float sh, double* __restrict__ sums) {
    __shared__ float sm[3 * MAXC][UDB / 64];
    float I[MAXC], P[MAXC], G[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) I[c] = P[c] = G[c] = 0.f;
    // items (row, j) are flattened over the grid: w is 138..552 on the path, a (j-block, row) grid would idle up to half the lanes
    const int items = B * H * w;
    for (int it = blockIdx.x * UDB + threadIdx.x; it < items; it += gridDim.x * UDB) {
        const int row = it / w, j = it - row * w;
        const int n = row / H, ho = row - n * H;
        const Lerp a = src_index(ho, sh, h, 0);
        float R[3][MAXC];
        updice_rows<S>(low, n, h, w, C, a, j, R);
        const uint8_t* lr = lab + (int64_t)row * W + S * j;
#pragma unroll
        for (int k = 0; k < S; ++k) {
            float z[MAXC];
            updice_pixel<S>(R, k + S / 2, C, z);
            softmax_inplace(z, C);
            const int l = lr[k];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                P[c] += z[c];
                if (c == l) { I[c] += z[c]; G[c] += 1.f; }
            }
        }
    }
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        float x = wave_sum(I[c]), y = wave_sum(P[c]), g = wave_sum(G[c]);
        if (lane == 0) { sm[c][wv] = x; sm[MAXC + c][wv] = y; sm[2 * MAXC + c][wv] = g; }
    }
    __syncthreads();
    if (threadIdx.x < 3 * MAXC) {
        const int q = threadIdx.x / MAXC, c = threadIdx.x % MAXC;
        if (c < C) {
            double t = 0.0;
            for (int k = 0; k < UDB / 64; ++k) t += (double)sm[threadIdx.x][k];
            atomicAdd(&sums[q * C + c], t);
        }
    }
}
#define UPDICE_SCALES(S_, STMT) \
    do { if (S_ == 2) { constexpr int S = 2; STMT; } else if (S_ == 4) { constexpr int S = 4; STMT; } else if (S_ == 8) { constexpr int S = 8; STMT; } \
         else { constexpr int S = 16; STMT; } } while (0)
extern "C" int tcct_updice_fwd(const float* low, const uint8_t* labels, int B, int h, int w, int H, int W, int C, double* sums,
                               float* loss, tcct_stream_t stream) {
    TCCT_CHECK(C >= 2 && C <= MAXC, "updice_fwd: C=%d unsupported (2..%d)", C, MAXC);
    const int Sc = h > 0 ? H / h : 0;
    TCCT_CHECK(B >= 1 && h >= 1 && w >= 1 && H == Sc * h && W == Sc * w && (Sc == 2 || Sc == 4 || Sc == 8 || Sc == 16),
               "updice_fwd: needs an integer scale 2/4/8/16 (got %dx%d -> %dx%d)", h, w, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(sums, 0, sizeof(double) * 3 * C, st) != hipSuccess) { tcct_set_error("updice_fwd: memset failed"); return -2; }
    TCCT_CHECK((int64_t)B * H * w < (1LL << 31), "updice_fwd: tensor too large");
    UPDICE_SCALES(Sc, hipLaunchKernelGGL(k_updice_sums<S>, dim3(tcct_grid((int64_t)B * H * w, UDB, 512)), dim3(UDB), 0, st, low, labels, B, h, w, H, W, C, (float)h / (float)H, sums));
    hipLaunchKernelGGL(k_dice_finalize, dim3(1), dim3(64), 0, st, sums, C, loss);
    TCCT_LAUNCH_OK();
}
// pass 1: T[n, ho, j, c] = sum_p ww(p, j) G[n, ho, p, c] over the 2S pixels p = S j - S/2 + kk of row ho whose interpolation touches column j
// (weight l1 for kk < S, l0 for kk >= S; 1 where the resize clamps: both taps are column j)
template <int S>
__global__ void __launch_bounds__(256) k_updice_bwd_w(const float* __restrict__ low, const uint8_t* __restrict__ lab, int B, int h, int w,
                                                      int H, int W, int C, float sh, const double* __restrict__ sums,
                                                      const float* __restrict__ gout, float gscale, float* __restrict__ T) {
    float ka[MAXC], kb[MAXC];
    const float gs = gscale * (gout ? *gout : 1.f);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < C) {
            const double U = 1.0 + sums[C + c] + sums[2 * C + c];
            ka[c] = (float)(-2.0 / U);
            kb[c] = (float)((1.0 + 2.0 * sums[c]) / (U * U));
        } else ka[c] = kb[c] = 0.f;
    }
    const int items = B * H * w;
    for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
        const int row = it / w, j = it - row * w;
        const int n = row / H, ho = row - n * H;
        const Lerp a = src_index(ho, sh, h, 0);
        float R[3][MAXC];
        updice_rows<S>(low, n, h, w, C, a, j, R);
        const uint8_t* lr = lab + (int64_t)row * W;
        float acc[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 2 * S; ++kk) {
            const int p = S * j - S / 2 + kk;
            if (p < 0 || p >= W) continue;
            const int k = (kk + S / 2) % S;
            const float f = ((float)k + 0.5f) / (float)S;
            float wt = kk < S ? f + 0.5f - (k < S / 2 ? 0.f : 1.f) : 1.f - (k < S / 2 ? f + 0.5f : f - 0.5f);
            // kk < S: this pixel's taps are (j-1, j), column j gets l1; kk >= S: taps (j, j+1), column j gets l0
            if ((j == 0 && kk < S) || (j == w - 1 && kk >= S)) wt = 1.f;      // clamped: both taps are column j
            float z[MAXC];
            updice_pixel<S>(R, kk, C, z);
            softmax_inplace(z, C);
            const int l = lr[p];
            float dp[MAXC], dot = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c) { dp[c] = kb[c] + (c == l ? ka[c] : 0.f); dot += z[c] * dp[c]; }
#pragma unroll
            for (int c = 0; c < MAXC; ++c) acc[c] += wt * (gs * z[c] * (dp[c] - dot));
        }
        float* t = T + ((int64_t)row * w + j) * C;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) t[c] = acc[c];
    }
}
// pass 2: dlow[n, i, j, c] = sum_ho wh(ho, i) T[n, ho, j, c]
__global__ void k_updice_bwd_h(const float* __restrict__ T, int B, int h, int wC, int H, int S, float sh, float* __restrict__ dlow) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= wC) return;
    for (int row = blockIdx.y; row < B * h; row += gridDim.y) {
        const int n = row / h, i = row - n * h;
        const int o0 = max(0, S * i - S / 2), o1 = min(H - 1, S * i + (3 * S) / 2 - 1);
        float acc = 0.f;
        for (int o = o0; o <= o1; ++o) {
            const Lerp a = src_index(o, sh, h, 0);
            const float wt = (a.i0 == i ? a.l0 : 0.f) + (a.i1 == i ? a.l1 : 0.f);
            acc += wt * T[((int64_t)n * H + o) * wC + e];
        }
        dlow[(int64_t)row * wC + e] = acc;
    }
}
extern "C" int tcct_updice_bwd(const float* low, const uint8_t* labels, int B, int h, int w, int H, int W, int C, const double* sums,
                               const float* grad_out, float grad_scale, float* ws, float* dlow, tcct_stream_t stream) {
    TCCT_CHECK(C >= 2 && C <= MAXC, "updice_bwd: C=%d unsupported", C);
    TCCT_CHECK(B >= 1 && h >= 1 && w >= 1 && H % h == 0 && W % w == 0 && H / h == W / w && H / h >= 2 && H / h <= 16,
               "updice_bwd: needs an integer scale 2..16 (got %dx%d -> %dx%d)", h, w, H, W);
    TCCT_CHECK(ws != nullptr, "updice_bwd: workspace [B,H,w,C] fp32 is NULL");
    hipStream_t st = (hipStream_t)stream;
    const int S = H / h;
    TCCT_CHECK(S == 2 || S == 4 || S == 8 || S == 16, "updice_bwd: scale %d unsupported (2/4/8/16)", S);
    TCCT_CHECK((int64_t)B * H * w < (1LL << 31), "updice_bwd: tensor too large");
    { const int Sc = S; UPDICE_SCALES(Sc, hipLaunchKernelGGL(k_updice_bwd_w<S>, dim3(tcct_grid((int64_t)B * H * w, 256, 1 << 14)), dim3(256), 0, st, low, labels, B, h, w, H, W, C, (float)h / (float)H,
                                                              sums, grad_out, grad_scale, ws)); }
    const int wC = w * C, gx2 = (wC + 255) / 256;
    int gy2 = B * h; if (gy2 > 65535) gy2 = 65535;
    hipLaunchKernelGGL(k_updice_bwd_h, dim3(gx2, gy2), dim3(256), 0, st, ws, B, h, wC, H, S, (float)h / (float)H, dlow);
    TCCT_LAUNCH_OK();
}

// softmax probability of the labelled class (FPL sort key, nets/reg.py:89) and argmax class (predict, loop_seg.py:32)
template <typename T>
__global__ void k_softmax_pick(const T* __restrict__ logits, const uint8_t* __restrict__ lab, int64_t M, int C,
                               float* __restrict__ prob_lab, uint8_t* __restrict__ argmax) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        float z[MAXC], mx = -INFINITY;
        int am = 0;
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            z[c] = c < C ? ldf(logits + i * C + c) : -INFINITY;
            if (z[c] > mx) { mx = z[c]; am = c; }
        }
        if (argmax) argmax[i] = (uint8_t)am;
        if (prob_lab) {
            float s = 0.f, pl = 0.f;
            int l = lab[i];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) { float e = c < C ? __expf(z[c] - mx) : 0.f; s += e; if (c == l) pl = e; }
            prob_lab[i] = pl / s;
        }
    }
}
extern "C" int tcct_softmax_pick(const void* logits, const uint8_t* labels, int64_t M, int C, float* prob_lab,
                                 uint8_t* argmax, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C >= 2 && C <= MAXC, "softmax_pick: C=%d unsupported", C);
    TCCT_CHECK(!(prob_lab && !labels), "softmax_pick: prob_lab needs labels");
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_softmax_pick<T>, dim3(tcct_grid(M, LB, 1 << 16)), dim3(LB), 0, (hipStream_t)stream, (const T*)logits, labels, M, C, prob_lab, argmax));
    TCCT_LAUNCH_OK();
}

// per-sample per-class {intersection, pred count, label count} for MDiceLoss/MIouLoss scores (kite/losses/miou.py:28-91)
__global__ void k_confusion(const uint8_t* __restrict__ pred, const uint8_t* __restrict__ lab, int64_t HW, int C,
                            float* __restrict__ out /*[N][C][3]*/) {
    __shared__ float sm[3 * MAXC];
    const int n = blockIdx.y;
    if (threadIdx.x < 3 * MAXC) sm[threadIdx.x] = 0.f;
    __syncthreads();
    float I[MAXC], P[MAXC], G[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) I[c] = P[c] = G[c] = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += (int64_t)gridDim.x * blockDim.x) {
        int p = pred[n * HW + i], l = lab[n * HW + i];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) { P[c] += (p == c); G[c] += (l == c); I[c] += (p == c && l == c); }
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        float a = wave_sum(I[c]), b = wave_sum(P[c]), g = wave_sum(G[c]);
        if ((threadIdx.x & 63) == 0) { atomicAdd(&sm[c * 3], a); atomicAdd(&sm[c * 3 + 1], b); atomicAdd(&sm[c * 3 + 2], g); }
    }
    __syncthreads();
    if (threadIdx.x < 3 * C) atomicAdd(&out[(int64_t)n * C * 3 + threadIdx.x], sm[threadIdx.x]);
}
extern "C" int tcct_confusion_counts(const uint8_t* pred, const uint8_t* labels, int N, int64_t HW, int C, float* out,
                                     tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= MAXC, "confusion_counts: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out, 0, sizeof(float) * N * C * 3, st) != hipSuccess) { tcct_set_error("confusion_counts: memset failed"); return -2; }
    dim3 grid(tcct_grid(HW, LB, 256), N);
    hipLaunchKernelGGL(k_confusion, grid, dim3(LB), 0, st, pred, labels, HW, C, out);
    TCCT_LAUNCH_OK();
}

// ----------------------------------------------------------------------- channel slice  T [M,C] -> fp32 [M,n] and back
template <typename T>
__global__ void k_slice_fwd(const T* __restrict__ x, float* __restrict__ y, int64_t M, int C, int start, int n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = ldf(x + (i / n) * C + start + (i % n));
}
template <typename T>
__global__ void k_slice_bwd(const float* __restrict__ dy, T* __restrict__ dx, int64_t M, int C, int start, int n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * C; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % C) - start;
        stf(dx + i, (c >= 0 && c < n) ? dy[(i / C) * n + c] : 0.f);
    }
}
extern "C" int tcct_slice_channels_fwd(const void* x, float* y, int64_t M, int C, int start, int n, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(start >= 0 && n > 0 && start + n <= C, "slice_channels_fwd: bad range");
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_slice_fwd<T>, dim3(tcct_grid(M * n, LB, 1 << 16)), dim3(LB), 0, (hipStream_t)stream, (const T*)x, y, M, C, start, n));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_slice_channels_bwd(const float* dy, void* dx, int64_t M, int C, int start, int n, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(start >= 0 && n > 0 && start + n <= C, "slice_channels_bwd: bad range");
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_slice_bwd<T>, dim3(tcct_grid(M * C, LB, 1 << 16)), dim3(LB), 0, (hipStream_t)stream, dy, (T*)dx, M, C, start, n));
    TCCT_LAUNCH_OK();
}

// labels -> fp32 one-hot slice [M,n] (classes start..start+n-1) and the vertical label-edge map prob_true (nets/reg.py:111-114)
__global__ void k_label_planes(const uint8_t* __restrict__ lab, float* __restrict__ onehot, float* __restrict__ edge, int N,
                               int H, int W, int start, int n) {
    const int64_t total = (int64_t)N * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int l = lab[i];
        if (onehot)
            for (int c = 0; c < n; ++c) onehot[i * n + c] = (l == start + c) ? 1.f : 0.f;
        if (edge) {
            int h = (int)((i / W) % H);
            float e = 0.f;
            if (h > 0) {
                int lp = lab[i - W];
                // sum_c |onehot_c[h] - onehot_c[h-1]| over classes start..start+n-1, clamped at 1
                int cnt = 0;
                if (l != lp) { cnt += (l >= start && l < start + n); cnt += (lp >= start && lp < start + n); }
                e = cnt > 0 ? 1.f : 0.f;
            }
            edge[i] = e;
        }
    }
}
extern "C" int tcct_label_planes(const uint8_t* labels, float* onehot, float* edge, int N, int H, int W, int start, int n,
                                 tcct_stream_t stream) {
    hipLaunchKernelGGL(k_label_planes, dim3(tcct_grid((int64_t)N * H * W, LB, 1 << 16)), dim3(LB), 0, (hipStream_t)stream, labels, onehot, edge, N, H, W, start, n);
    TCCT_LAUNCH_OK();
}

// ----------------------------------------------------------------------- Gumbel column softmax over H, summed over CH channels
// x, eps: fp32 [N,H,W,CH]; thread = one (n, w, ch) column; lanes run over (w,ch) => coalesced rows.
// out[n,h,w] = sum_ch g/(1e-6 + S) with g = softmax_H(x - log(-log(eps))/2), S = sum_H g   (nets/reg.py:118-128)
// stats[n,w,ch] = {m, Z, S}
// Block = 64 columns (lanes, so the 4 channels of a pixel sit in adjacent lanes) x GSEG row segments: each thread walks H/GSEG
// rows of its column (online max/sum), segment partials are merged through LDS.  z is recomputed per pass (2 x logf per element
// is cheaper than a third tensor round trip).
#define GSEG 8
__device__ __forceinline__ float gumbel_z(float x, float e) { return x - 0.5f * logf(-logf(e)); }

template <int CH>
__global__ void __launch_bounds__(64 * GSEG)
k_gumbel_fwd(const float* __restrict__ x, const float* __restrict__ eps, float* __restrict__ out,
             float* __restrict__ stats, int N, int H, int W) {
    __shared__ float sm[GSEG][64], sz[GSEG][64];
    const int WC = W * CH;
    const int64_t cols = (int64_t)N * WC;
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = col < cols;
    const int64_t cc = ok ? col : 0;
    const int64_t n = cc / WC;
    const int wc = (int)(cc % WC);
    const float* xp = x + n * (int64_t)H * WC + wc;
    const float* ep = eps + n * (int64_t)H * WC + wc;
    const int hs = (H + GSEG - 1) / GSEG;
    const int h0 = seg * hs, h1 = min(H, h0 + hs);
    float m = -INFINITY, Z = 0.f;
    for (int h = h0; h < h1; ++h) {
        // eps == 0 (probability 2^-24 per draw, i.e. a few per step at 8x800x1104x4) gives z = -inf: such an element contributes
        // exp(-inf) = 0, as in torch.softmax; while the running maximum is still -inf the rescale exp(m - mn) would be exp(NaN)
        float z = gumbel_z(xp[(int64_t)h * WC], ep[(int64_t)h * WC]);
        float mn = fmaxf(m, z);
        if (mn > -INFINITY) Z = Z * __expf(m - mn) + __expf(z - mn);
        m = mn;
    }
    sm[seg][lane] = m; sz[seg][lane] = Z;
    __syncthreads();
    m = -INFINITY;
#pragma unroll
    for (int g = 0; g < GSEG; ++g) m = fmaxf(m, sm[g][lane]);
    Z = 0.f;
#pragma unroll
    for (int g = 0; g < GSEG; ++g) if (sm[g][lane] > -INFINITY) Z += sz[g][lane] * __expf(sm[g][lane] - m);
    __syncthreads();
    float S = 0.f;
    for (int h = h0; h < h1; ++h) S += __expf(gumbel_z(xp[(int64_t)h * WC], ep[(int64_t)h * WC]) - m) / Z;
    sm[seg][lane] = S;
    __syncthreads();
    S = 0.f;
#pragma unroll
    for (int g = 0; g < GSEG; ++g) S += sm[g][lane];
    if (ok && seg == 0) { stats[cc * 3] = m; stats[cc * 3 + 1] = Z; stats[cc * 3 + 2] = S; }
    const float den = 1.f / (1e-6f + S);
    for (int h = h0; h < h1; ++h) {
        float v = __expf(gumbel_z(xp[(int64_t)h * WC], ep[(int64_t)h * WC]) - m) / Z * den;
#pragma unroll
        for (int o = CH >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (ok && (wc % CH) == 0) out[(n * H + h) * (int64_t)W + wc / CH] = v;
    }
}
// dx[n,h,w,ch] = g * ((dout[h] - D)/(eps+S) - D (1-S)/(eps+S)^2),  D = sum_h dout[h] g[h]
template <int CH>
__global__ void __launch_bounds__(64 * GSEG)
k_gumbel_bwd(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ stats,
             const float* __restrict__ dout, float* __restrict__ dx, int N, int H, int W) {
    __shared__ float sd[GSEG][64];
    const int WC = W * CH;
    const int64_t cols = (int64_t)N * WC;
    const int lane = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + lane;
    const bool ok = col < cols;
    const int64_t cc = ok ? col : 0;
    const int64_t n = cc / WC;
    const int wc = (int)(cc % WC);
    const int w = wc / CH;
    const float* xp = x + n * (int64_t)H * WC + wc;
    const float* ep = eps + n * (int64_t)H * WC + wc;
    const float* dp = dout + n * (int64_t)H * W + w;
    float* dxp = dx + n * (int64_t)H * WC + wc;
    const float m = stats[cc * 3], Z = stats[cc * 3 + 1], S = stats[cc * 3 + 2];
    const int hs = (H + GSEG - 1) / GSEG;
    const int h0 = seg * hs, h1 = min(H, h0 + hs);
    float D = 0.f;
    for (int h = h0; h < h1; ++h) D += dp[(int64_t)h * W] * (__expf(gumbel_z(xp[(int64_t)h * WC], ep[(int64_t)h * WC]) - m) / Z);
    sd[seg][lane] = D;
    __syncthreads();
    D = 0.f;
#pragma unroll
    for (int g = 0; g < GSEG; ++g) D += sd[g][lane];
    const float den = 1.f / (1e-6f + S);
    const float k2 = D * (1.f - S) * den * den;
    if (ok)
        for (int h = h0; h < h1; ++h) {
            float g = __expf(gumbel_z(xp[(int64_t)h * WC], ep[(int64_t)h * WC]) - m) / Z;
            dxp[(int64_t)h * WC] = g * ((dp[(int64_t)h * W] - D) * den - k2);
        }
}
extern "C" int tcct_gumbel_colsoftmax_fwd(const float* x, const float* eps, float* out, float* stats, int N, int H, int W,
                                          int CH, tcct_stream_t stream) {
    TCCT_CHECK(CH == 4, "gumbel_colsoftmax_fwd: CH=%d unsupported (4)", CH);
    int64_t cols = (int64_t)N * W * CH;
    hipLaunchKernelGGL(k_gumbel_fwd<4>, dim3((unsigned)((cols + 63) / 64)), dim3(64 * GSEG), 0, (hipStream_t)stream, x, eps, out, stats, N, H, W);
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_gumbel_colsoftmax_bwd(const float* x, const float* eps, const float* stats, const float* dout, float* dx,
                                          int N, int H, int W, int CH, tcct_stream_t stream) {
    TCCT_CHECK(CH == 4, "gumbel_colsoftmax_bwd: CH=%d unsupported (4)", CH);
    int64_t cols = (int64_t)N * W * CH;
    hipLaunchKernelGGL(k_gumbel_bwd<4>, dim3((unsigned)((cols + 63) / 64)), dim3(64 * GSEG), 0, (hipStream_t)stream, x, eps, stats, dout, dx, N, H, W);
    TCCT_LAUNCH_OK();
}

// ----------------------------------------------------------------------- column (over H) softmax and weighted column sum, fp32 [N,H,W]
// N*W columns only (8 832 at the bench shape): one thread per column leaves the chip empty and every thread walking 800 dependent
// rows.  Block = 64 adjacent columns x CSEG row segments (lanes run along W => coalesced rows), segment partials combined in LDS.
#define CSEG 16
struct ColPos { int64_t off; int h0, h1, lane, seg; bool ok; };
__device__ __forceinline__ ColPos col_pos(int N, int H, int W) {
    ColPos p;
    p.lane = threadIdx.x & 63; p.seg = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + p.lane;
    p.ok = col < (int64_t)N * W;
    const int64_t cc = p.ok ? col : 0;
    p.off = (cc / W) * (int64_t)H * W + (cc % W);
    const int hs = (H + CSEG - 1) / CSEG;
    p.h0 = p.seg * hs; p.h1 = min(H, p.h0 + hs);
    return p;
}
__global__ void __launch_bounds__(64 * CSEG) k_colsoftmax_fwd(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W) {
    __shared__ float sm[CSEG][64], sz[CSEG][64];
    const ColPos p = col_pos(N, H, W);
    const float* xp = x + p.off;
    float m = -INFINITY, Z = 0.f;
    for (int h = p.h0; h < p.h1; ++h) {
        float z = xp[(int64_t)h * W];
        float mn = fmaxf(m, z);
        Z = Z * __expf(m - mn) + __expf(z - mn);
        m = mn;
    }
    sm[p.seg][p.lane] = m; sz[p.seg][p.lane] = Z;
    __syncthreads();
    m = -INFINITY;
#pragma unroll
    for (int g = 0; g < CSEG; ++g) m = fmaxf(m, sm[g][p.lane]);
    Z = 0.f;
#pragma unroll
    for (int g = 0; g < CSEG; ++g) { float mg = sm[g][p.lane]; Z += mg == -INFINITY ? 0.f : sz[g][p.lane] * __expf(mg - m); }
    const float inv = 1.f / Z;
    if (p.ok)
        for (int h = p.h0; h < p.h1; ++h) y[p.off + (int64_t)h * W] = __expf(xp[(int64_t)h * W] - m) * inv;
}
__global__ void __launch_bounds__(64 * CSEG) k_colsoftmax_bwd(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W) {
    __shared__ float sd[CSEG][64];
    const ColPos p = col_pos(N, H, W);
    float dot = 0.f;
    for (int h = p.h0; h < p.h1; ++h) dot += y[p.off + (int64_t)h * W] * dy[p.off + (int64_t)h * W];
    sd[p.seg][p.lane] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int g = 0; g < CSEG; ++g) dot += sd[g][p.lane];
    if (p.ok)
        for (int h = p.h0; h < p.h1; ++h) dx[p.off + (int64_t)h * W] = y[p.off + (int64_t)h * W] * (dy[p.off + (int64_t)h * W] - dot);
}
extern "C" int tcct_colsoftmax_fwd(const float* x, float* y, int N, int H, int W, tcct_stream_t stream) {
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1, "colsoftmax_fwd: empty tensor");
    hipLaunchKernelGGL(k_colsoftmax_fwd, dim3((unsigned)(((int64_t)N * W + 63) / 64)), dim3(64 * CSEG), 0, (hipStream_t)stream, x, y, N, H, W);
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_colsoftmax_bwd(const float* y, const float* dy, float* dx, int N, int H, int W, tcct_stream_t stream) {
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1, "colsoftmax_bwd: empty tensor");
    hipLaunchKernelGGL(k_colsoftmax_bwd, dim3((unsigned)(((int64_t)N * W + 63) / 64)), dim3(64 * CSEG), 0, (hipStream_t)stream, y, dy, dx, N, H, W);
    TCCT_LAUNCH_OK();
}

// edge[n,w] = sum_h x[n,h,w] * wts[h]   (column soft-argmax, nets/reg.py:146-150; wts = (h + jitter - .5)/H)
__global__ void __launch_bounds__(64 * CSEG) k_colwsum_fwd(const float* __restrict__ x, const float* __restrict__ wts, float* __restrict__ out, int N, int H, int W) {
    __shared__ float sd[CSEG][64];
    const ColPos p = col_pos(N, H, W);
    float s = 0.f;
    for (int h = p.h0; h < p.h1; ++h) s += x[p.off + (int64_t)h * W] * wts[h];
    sd[p.seg][p.lane] = s;
    __syncthreads();
    if (p.seg == 0 && p.ok) {
        s = 0.f;
#pragma unroll
        for (int g = 0; g < CSEG; ++g) s += sd[g][p.lane];
        out[(int64_t)blockIdx.x * 64 + p.lane] = s;
    }
}
__global__ void k_colwsum_bwd(const float* __restrict__ dout, const float* __restrict__ wts, float* __restrict__ dx, int N, int H, int W) {
    // grid.y strides over rows (n, h); threads run along W: no per-element index division
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= W) return;
    for (int row = blockIdx.y; row < N * H; row += gridDim.y) {
        const int n = row / H, h = row - n * H;
        dx[(int64_t)row * W + w] = dout[(int64_t)n * W + w] * wts[h];
    }
}
extern "C" int tcct_colwsum_fwd(const float* x, const float* wts, float* out, int N, int H, int W, tcct_stream_t stream) {
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1, "colwsum_fwd: empty tensor");
    hipLaunchKernelGGL(k_colwsum_fwd, dim3((unsigned)(((int64_t)N * W + 63) / 64)), dim3(64 * CSEG), 0, (hipStream_t)stream, x, wts, out, N, H, W);
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_colwsum_bwd(const float* dout, const float* wts, float* dx, int N, int H, int W, tcct_stream_t stream) {
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1, "colwsum_bwd: empty tensor");
    const int64_t rows = (int64_t)N * H;
    dim3 g((unsigned)((W + LB - 1) / LB), (unsigned)(rows < 2048 ? rows : 2048));
    hipLaunchKernelGGL(k_colwsum_bwd, g, dim3(LB), 0, (hipStream_t)stream, dout, wts, dx, N, H, W);
    TCCT_LAUNCH_OK();
}

// ----------------------------------------------------------------------- layer-boundary coordinates from a class-index mask
// out[n][k-1][w] = #{h : mask[n,h,w] < k}, k = 1..C-1: for a layered (top-to-bottom monotone) segmentation this is the row at which
// layer k starts in column w; counting instead of searching the first switch keeps isolated mislabelled pixels from moving the
// boundary by more than one row each (the reference only has the unused soft_argmax, nets/reg.py:27-35).
__global__ void __launch_bounds__(64 * CSEG) k_mask_boundaries(const uint8_t* __restrict__ mask, int32_t* __restrict__ out, int N, int H, int W, int C) {
    __shared__ int sc[CSEG][MAXC][64];
    const ColPos p = col_pos(N, H, W);
    int cnt[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) cnt[k] = 0;
    for (int h = p.h0; h < p.h1; ++h) {
        const int l = mask[p.off + (int64_t)h * W];
#pragma unroll
        for (int k = 1; k < MAXC; ++k) cnt[k] += (l < k);
    }
#pragma unroll
    for (int k = 1; k < MAXC; ++k) sc[p.seg][k][p.lane] = cnt[k];
    __syncthreads();
    if (p.ok)
        for (int k = 1 + p.seg; k < C; k += CSEG) {
            int s = 0;
#pragma unroll
            for (int g = 0; g < CSEG; ++g) s += sc[g][k][p.lane];
            const int64_t col = (int64_t)blockIdx.x * 64 + p.lane;
            const int64_t n = col / W, w = col % W;
            out[(n * (C - 1) + (k - 1)) * W + w] = s;
        }
}
extern "C" int tcct_mask_boundaries(const uint8_t* mask, int32_t* out, int N, int H, int W, int C, tcct_stream_t stream) {
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1, "mask_boundaries: empty tensor");
    TCCT_CHECK(C >= 2 && C <= MAXC, "mask_boundaries: C=%d unsupported (2..%d)", C, MAXC);
    hipLaunchKernelGGL(k_mask_boundaries, dim3((unsigned)(((int64_t)N * W + 63) / 64)), dim3(64 * CSEG), 0, (hipStream_t)stream, mask, out, N, H, W, C);
    TCCT_LAUNCH_OK();
}

// ----------------------------------------------------------------------- mean squared error on fp32 vectors (nn.MSELoss, reg.py:108)
__global__ void k_mse_fwd(const float* __restrict__ a, const float* __restrict__ b, int64_t n, double* __restrict__ acc) {
    __shared__ float sm[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float d = a[i] - b[i];
        s += d * d;
    }
    s = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(acc, (double)s);
}
__global__ void k_mse_fin(const double* acc, int64_t n, float* out) { if (threadIdx.x == 0) *out = (float)(*acc / (double)n); }
// da = coef * (a-b), db = -da  with coef = 2/n * gscale * (*gout); either output may be NULL
__global__ void k_mse_bwd(const float* __restrict__ a, const float* __restrict__ b, int64_t n, const float* __restrict__ gout,
                          float gscale, float* __restrict__ da, float* __restrict__ db) {
    const float coef = 2.f / (float)n * gscale * (gout ? *gout : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float d = coef * (a[i] - b[i]);
        if (da) da[i] = d;
        if (db) db[i] = -d;
    }
}
extern "C" int tcct_mse_fwd(const float* a, const float* b, int64_t n, double* acc, float* out, tcct_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, sizeof(double), st) != hipSuccess) { tcct_set_error("mse_fwd: memset failed"); return -2; }
    hipLaunchKernelGGL(k_mse_fwd, dim3(tcct_grid(n, LB, 1024)), dim3(LB), 0, st, a, b, n, acc);
    hipLaunchKernelGGL(k_mse_fin, dim3(1), dim3(64), 0, st, acc, n, out);
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_mse_bwd(const float* a, const float* b, int64_t n, const float* grad_out, float grad_scale, float* da,
                            float* db, tcct_stream_t stream) {
    hipLaunchKernelGGL(k_mse_bwd, dim3(tcct_grid(n, LB, 1 << 16)), dim3(LB), 0, (hipStream_t)stream, a, b, n, grad_out, grad_scale, da, db);
    TCCT_LAUNCH_OK();
}
