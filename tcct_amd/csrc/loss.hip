// Loss-side kernels: fused softmax + batch-global Dice sums (+ backward), the column-wise (over H) pieces of the
// boundary-regression loss, and small fp32 helpers.  Labels are class indices (uint8), never one-hot.
#include "common.h"

#define LB 256
// The class-count bound MAXC sizes every per-class register array and loop of loss_classes.inc (entries beyond C are predicated off, not skipped): the
// benchmark's 5 classes ran the 8-wide code with 3 of 8 exponentials / accumulators wasted, so 5 has its own instantiation (round 4)
#define MAXC 5
#define MCNS mc5
#include "loss_classes.inc"
#undef MCNS
#undef MAXC
#define MAXC 8

#define MCNS mc8
#include "loss_classes.inc"
#undef MCNS
#undef MAXC
#define MAXC 16
#define MCNS mc16
#include "loss_classes.inc"
#undef MCNS

extern "C" int tcct_softmax_dice_fwd(const void* logits, const uint8_t* labels, int64_t M, int C, double* sums, float* loss, int dtype, tcct_stream_t stream) {
    return C == 5 ? mc5::tcct_softmax_dice_fwd_impl(logits, labels, M, C, sums, loss, dtype, stream) : (C <= 8 ? mc8::tcct_softmax_dice_fwd_impl(logits, labels, M, C, sums, loss, dtype, stream) : mc16::tcct_softmax_dice_fwd_impl(logits, labels, M, C, sums, loss, dtype, stream));
}
extern "C" int tcct_softmax_dice_bwd(const void* logits, const uint8_t* labels, int64_t M, int C, const double* sums, const float* grad_out, float grad_scale, void* dlogits, int dtype, tcct_stream_t stream) {
    return C == 5 ? mc5::tcct_softmax_dice_bwd_impl(logits, labels, M, C, sums, grad_out, grad_scale, dlogits, dtype, stream) : (C <= 8 ? mc8::tcct_softmax_dice_bwd_impl(logits, labels, M, C, sums, grad_out, grad_scale, dlogits, dtype, stream) : mc16::tcct_softmax_dice_bwd_impl(logits, labels, M, C, sums, grad_out, grad_scale, dlogits, dtype, stream));
}
extern "C" int tcct_updice_fwd(const float* low, const uint8_t* labels, int B, int h, int w, int H, int W, int C, double* sums, float* loss, tcct_stream_t stream) {
    return C == 5 ? mc5::tcct_updice_fwd_impl(low, labels, B, h, w, H, W, C, sums, loss, stream) : (C <= 8 ? mc8::tcct_updice_fwd_impl(low, labels, B, h, w, H, W, C, sums, loss, stream) : mc16::tcct_updice_fwd_impl(low, labels, B, h, w, H, W, C, sums, loss, stream));
}
extern "C" int tcct_updice_bwd(const float* low, const uint8_t* labels, int B, int h, int w, int H, int W, int C, const double* sums, const float* grad_out, float grad_scale, float* ws, float* dlow, tcct_stream_t stream) {
    return C == 5 ? mc5::tcct_updice_bwd_impl(low, labels, B, h, w, H, W, C, sums, grad_out, grad_scale, ws, dlow, stream) : (C <= 8 ? mc8::tcct_updice_bwd_impl(low, labels, B, h, w, H, W, C, sums, grad_out, grad_scale, ws, dlow, stream) : mc16::tcct_updice_bwd_impl(low, labels, B, h, w, H, W, C, sums, grad_out, grad_scale, ws, dlow, stream));
}
/* KiteSeg.grad_calc with MultiLoss(DiceLoss) and deep supervision (reference kite/loopback.py:62-73, kite/losses/loss.py:15-32,83-99):
 *   loss = sum_{i = 3, 2, 1} coff * Dice(resize(low_i)) + Dice(logits)      (fp32 scalar arithmetic in that order)
 * logits [B,H,W,C] (dtype), low_i fp32 [B,h_i,w_i,C] (nullable from the back: low3 == NULL -> two heads, ...), labels uint8 [B,H,W];
 * sums fp64 [4 * 3C] (cleared here; head i at offset i * 3C, kept for tcct_softmax_dice_bwd / tcct_updice_bwd with grad_scale = coff). */
extern "C" int tcct_dice_ds_fwd(const void* logits, int dtype, const uint8_t* labels, int B, int H, int W, int C, const float* low1, int h1, int w1,
                                const float* low2, int h2, int w2, const float* low3, int h3, int w3, float coff, double* sums, float* loss,
                                tcct_stream_t stream) {
    const float* lows[3] = {low1, low2, low3};
    const int lh[3] = {h1, h2, h3}, lw[3] = {w1, w2, w3};
    const int nlow = low1 ? (low2 ? (low3 ? 3 : 2) : 1) : 0;
    return C == 5 ? mc5::tcct_dice_ds_fwd_impl(logits, dtype, labels, B, H, W, C, lows, lh, lw, nlow, coff, sums, loss, stream) : (C <= 8 ? mc8::tcct_dice_ds_fwd_impl(logits, dtype, labels, B, H, W, C, lows, lh, lw, nlow, coff, sums, loss, stream) : mc16::tcct_dice_ds_fwd_impl(logits, dtype, labels, B, H, W, C, lows, lh, lw, nlow, coff, sums, loss, stream));
}
extern "C" int tcct_softmax_pick(const void* logits, const uint8_t* labels, int64_t M, int C, float* prob_lab, uint8_t* argmax, int dtype, tcct_stream_t stream) {
    return C <= 8 ? mc8::tcct_softmax_pick_impl(logits, labels, M, C, prob_lab, argmax, dtype, stream) : mc16::tcct_softmax_pick_impl(logits, labels, M, C, prob_lab, argmax, dtype, stream);
}
extern "C" int tcct_confusion_counts(const uint8_t* pred, const uint8_t* labels, int N, int64_t HW, int C, float* out, tcct_stream_t stream) {
    return C <= 8 ? mc8::tcct_confusion_counts_impl(pred, labels, N, HW, C, out, stream) : mc16::tcct_confusion_counts_impl(pred, labels, N, HW, C, out, stream);
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
// rows of its column (online max/sum), segment partials are merged through LDS.  z is recomputed in the second pass (2 x logf per element
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
    // Rows in groups of four: a thread walks ~100 rows that lie 17.6 KB apart, and with one load pair outstanding per wave the walk is a chain
    // of memory latencies.  The four rows of a group are loaded before any of them is used (hipcc refuses `#pragma unroll` on these loops).
    auto zrow = [&](int h) { return gumbel_z(xp[(int64_t)h * WC], ep[(int64_t)h * WC]); };
    int hq = h0;
    for (; hq + 4 <= h1; hq += 4) {
        const float z0 = zrow(hq), z1 = zrow(hq + 1), z2 = zrow(hq + 2), z3 = zrow(hq + 3);
        const float mn = fmaxf(fmaxf(m, fmaxf(z0, z1)), fmaxf(z2, z3));
        if (mn > -INFINITY) Z = Z * __expf(m - mn) + ((__expf(z0 - mn) + __expf(z1 - mn)) + (__expf(z2 - mn) + __expf(z3 - mn)));
        m = mn;
    }
    for (int h = hq; h < h1; ++h) {
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
    // S = sum_h g[h] with g = exp(z - m) / Z is the sum of a softmax column: sum_h exp(z_h - m) IS Z, so S = 1 up to the rounding of whichever order
    // the terms are added in (the reference's own value is 1 +- 1e-7).  Rounds 1-3 re-read both tensors to add the terms up again (a third of the
    // kernel's traffic: 699 MB moved for 254 MB of inputs and outputs at the bench shape); the column total of the first pass is used instead.
    const float S = 1.f;
    if (ok && seg == 0) { stats[cc * 3] = m; stats[cc * 3 + 1] = Z; stats[cc * 3 + 2] = S; }
    const float den = 1.f / (1e-6f + S);
    for (hq = h0; hq + 4 <= h1; hq += 4) {
        float v[4] = {zrow(hq), zrow(hq + 1), zrow(hq + 2), zrow(hq + 3)};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = __expf(v[j] - m) / Z * den;
#pragma unroll
            for (int o = CH >> 1; o > 0; o >>= 1) v[j] += __shfl_xor(v[j], o, 64);
            if (ok && (wc % CH) == 0) out[(n * H + hq + j) * (int64_t)W + wc / CH] = v[j];
        }
    }
    for (int h = hq; h < h1; ++h) {
        float v = __expf(zrow(h) - m) / Z * den;
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
    auto grow = [&](int h) { return __expf(gumbel_z(xp[(int64_t)h * WC], ep[(int64_t)h * WC]) - m) / Z; };
    int hq = h0;
    for (; hq + 4 <= h1; hq += 4) {        // groups of four rows: see k_gumbel_fwd
        const float x0 = xp[(int64_t)hq * WC], x1 = xp[(int64_t)(hq + 1) * WC], x2 = xp[(int64_t)(hq + 2) * WC], x3 = xp[(int64_t)(hq + 3) * WC];
        const float e0 = ep[(int64_t)hq * WC], e1 = ep[(int64_t)(hq + 1) * WC], e2 = ep[(int64_t)(hq + 2) * WC], e3 = ep[(int64_t)(hq + 3) * WC];
        const float d0 = dp[(int64_t)hq * W], d1 = dp[(int64_t)(hq + 1) * W], d2 = dp[(int64_t)(hq + 2) * W], d3 = dp[(int64_t)(hq + 3) * W];
        D += (d0 * (__expf(gumbel_z(x0, e0) - m) / Z) + d1 * (__expf(gumbel_z(x1, e1) - m) / Z))
           + (d2 * (__expf(gumbel_z(x2, e2) - m) / Z) + d3 * (__expf(gumbel_z(x3, e3) - m) / Z));
    }
    for (int h = hq; h < h1; ++h) D += dp[(int64_t)h * W] * grow(h);
    sd[seg][lane] = D;
    __syncthreads();
    D = 0.f;
#pragma unroll
    for (int g = 0; g < GSEG; ++g) D += sd[g][lane];
    const float den = 1.f / (1e-6f + S);
    const float k2 = D * (1.f - S) * den * den;
    if (ok) {
        for (hq = h0; hq + 4 <= h1; hq += 4) {
            const float g0 = grow(hq), g1 = grow(hq + 1), g2 = grow(hq + 2), g3 = grow(hq + 3);
            const float d0 = dp[(int64_t)hq * W], d1 = dp[(int64_t)(hq + 1) * W], d2 = dp[(int64_t)(hq + 2) * W], d3 = dp[(int64_t)(hq + 3) * W];
            dxp[(int64_t)hq * WC] = g0 * ((d0 - D) * den - k2);
            dxp[(int64_t)(hq + 1) * WC] = g1 * ((d1 - D) * den - k2);
            dxp[(int64_t)(hq + 2) * WC] = g2 * ((d2 - D) * den - k2);
            dxp[(int64_t)(hq + 3) * WC] = g3 * ((d3 - D) * den - k2);
        }
        for (int h = hq; h < h1; ++h) dxp[(int64_t)h * WC] = grow(h) * ((dp[(int64_t)h * W] - D) * den - k2);
    }
}
extern "C" int tcct_gumbel_colsoftmax_fwd(const float* x, const float* eps, float* out, float* stats, int N, int H, int W,
                                          int CH, tcct_stream_t stream) {
    // CH = classes - 1 (reference nets/reg.py:110: pred[:,1:]): 4 for the 5-class GOALS set, 8 for the 9-class Duke / HCMS models; the channel sum
    // is an xor-butterfly over adjacent lanes, so CH must be a power of two <= 16
    TCCT_CHECK(CH == 2 || CH == 4 || CH == 8 || CH == 16, "gumbel_colsoftmax_fwd: CH=%d unsupported (2, 4, 8, 16)", CH);
    int64_t cols = (int64_t)N * W * CH;
    const dim3 grid((unsigned)((cols + 63) / 64)), block(64 * GSEG);
    hipStream_t st = (hipStream_t)stream;
    if (CH == 2) hipLaunchKernelGGL(k_gumbel_fwd<2>, grid, block, 0, st, x, eps, out, stats, N, H, W);
    else if (CH == 4) hipLaunchKernelGGL(k_gumbel_fwd<4>, grid, block, 0, st, x, eps, out, stats, N, H, W);
    else if (CH == 8) hipLaunchKernelGGL(k_gumbel_fwd<8>, grid, block, 0, st, x, eps, out, stats, N, H, W);
    else hipLaunchKernelGGL(k_gumbel_fwd<16>, grid, block, 0, st, x, eps, out, stats, N, H, W);
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_gumbel_colsoftmax_bwd(const float* x, const float* eps, const float* stats, const float* dout, float* dx,
                                          int N, int H, int W, int CH, tcct_stream_t stream) {
    TCCT_CHECK(CH == 2 || CH == 4 || CH == 8 || CH == 16, "gumbel_colsoftmax_bwd: CH=%d unsupported (2, 4, 8, 16)", CH);
    int64_t cols = (int64_t)N * W * CH;
    const dim3 grid((unsigned)((cols + 63) / 64)), block(64 * GSEG);
    hipStream_t st = (hipStream_t)stream;
    if (CH == 2) hipLaunchKernelGGL(k_gumbel_bwd<2>, grid, block, 0, st, x, eps, stats, dout, dx, N, H, W);
    else if (CH == 4) hipLaunchKernelGGL(k_gumbel_bwd<4>, grid, block, 0, st, x, eps, stats, dout, dx, N, H, W);
    else if (CH == 8) hipLaunchKernelGGL(k_gumbel_bwd<8>, grid, block, 0, st, x, eps, stats, dout, dx, N, H, W);
    else hipLaunchKernelGGL(k_gumbel_bwd<16>, grid, block, 0, st, x, eps, stats, dout, dx, N, H, W);
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
    __shared__ int sc[CSEG][MAXC - 1][64];        // (MAXC is 16 here: the second instantiation of loss_classes.inc leaves it defined)
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
    for (int k = 1; k < MAXC; ++k) sc[p.seg][k - 1][p.lane] = cnt[k];
    __syncthreads();
    if (p.ok)
        for (int k = 1 + p.seg; k < C; k += CSEG) {
            int s = 0;
#pragma unroll
            for (int g = 0; g < CSEG; ++g) s += sc[g][k - 1][p.lane];
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
