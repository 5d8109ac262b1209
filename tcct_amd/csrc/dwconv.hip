// Depthwise 3x3 convolution (pad 1, stride 1 or 2) on NHWC, HBM-bound VALU stencil.  VEC=4 channels per lane when
// C % 4 == 0, scalar otherwise (the 1-channel lap_map convs of the boundary-regression loss).
#include "common.h"

#define DB 256

template <typename T, int VEC>
__device__ __forceinline__ void ldv(const T* p, float* o) {
    if (VEC == 4) { f4 a = ld4(p); o[0] = a.v[0]; o[1] = a.v[1]; o[2] = a.v[2]; o[3] = a.v[3]; }
    else o[0] = ldf(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void stv(T* p, const float* o) {
    if (VEC == 4) { f4 a; a.v[0] = o[0]; a.v[1] = o[1]; a.v[2] = o[2]; a.v[3] = o[3]; st4(p, a); }
    else stf(p, o[0]);
}

// Sliding-window form: thread = (channel vector, strip of SEG output pixels along W); the 3x3 window of input columns and
// the 9 per-channel taps stay in registers, so each output costs `stride` new column loads (3 rows) instead of 9 loads.
// FLIP=1 uses w[2-ky][2-kx]: with stride 1 that is exactly the input-gradient convolution.
#define DW_SEG 16
template <typename T, int VEC, int STRIDE, bool FLIP>
__global__ void k_dw_fwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                         T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo, int add_input) {
    const int CV = C / VEC;
    const int segs = (Wo + DW_SEG - 1) / DW_SEG;
    const int64_t total = (int64_t)N * Ho * segs * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * VEC;
        int64_t q = i / CV;
        const int sg = (int)(q % segs);
        q /= segs;
        const int ho = (int)(q % Ho);
        const int64_t n = q / Ho;
        float wk[9][VEC], bv[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            bv[k] = bias ? bias[c + k] : 0.f;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) wk[tp][k] = w[(c + k) * 9 + (FLIP ? 8 - tp : tp)];
        }
        const int wo0 = sg * DW_SEG, wo1 = min(Wo, wo0 + DW_SEG);
        const int hi0 = ho * STRIDE - 1;
        const T* rows[3];
        bool rok[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int hi = hi0 + ky;
            rok[ky] = hi >= 0 && hi < H;
            rows[ky] = x + ((n * H + (rok[ky] ? hi : 0)) * (int64_t)W) * C + c;
        }
        float col[3][3][VEC];      // [kx][ky]
        auto load_col = [&](int kxslot, int wi) {
            bool cok = wi >= 0 && wi < W;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                if (cok && rok[ky]) ldv<T, VEC>(rows[ky] + (int64_t)wi * C, col[kxslot][ky]);
                else {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) col[kxslot][ky][k] = 0.f;
                }
            }
        };
        int wi = wo0 * STRIDE - 1;
        load_col(0, wi); load_col(1, wi + 1);
        for (int wo = wo0; wo < wo1; ++wo) {
            load_col(2, wo * STRIDE + 1);
            float acc[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float a = bv[k];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) a += col[kx][ky][k] * wk[ky * 3 + kx][k];
                if (add_input) a += col[1][1][k];
                acc[k] = a;
            }
            stv<T, VEC>(y + ((n * Ho + ho) * (int64_t)Wo + wo) * C + c, acc);
            if (STRIDE == 1) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) { col[0][ky][k] = col[1][ky][k]; col[1][ky][k] = col[2][ky][k]; }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int k = 0; k < VEC; ++k) col[0][ky][k] = col[2][ky][k];
                load_col(1, (wo + 1) * STRIDE);
            }
        }
    }
}

template <typename T, int VEC, bool FLIP>
static void dw_fwd_launch(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C, int stride,
                          int Ho, int Wo, int add_input, hipStream_t st) {
    int segs = (Wo + DW_SEG - 1) / DW_SEG;
    int64_t total = (int64_t)N * Ho * segs * (C / VEC);
    dim3 g(tcct_grid(total, DB, 1 << 16)), b(DB);
    if (stride == 1) hipLaunchKernelGGL((k_dw_fwd<T, VEC, 1, FLIP>), g, b, 0, st, (const T*)x, w, bias, (T*)y, N, H, W, C, Ho, Wo, add_input);
    else hipLaunchKernelGGL((k_dw_fwd<T, VEC, 2, FLIP>), g, b, 0, st, (const T*)x, w, bias, (T*)y, N, H, W, C, Ho, Wo, add_input);
}

extern "C" int tcct_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C,
                                  int stride, int add_input, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "dwconv3x3_fwd: stride %d", stride);
    TCCT_CHECK(!(add_input && stride != 1), "dwconv3x3_fwd: add_input needs stride 1");
    int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    int vec = (C % 4 == 0) ? 4 : 1;
    hipStream_t st = (hipStream_t)stream;
    if (vec == 4) { TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 4, false>(x, w, bias, y, N, H, W, C, stride, Ho, Wo, add_input, st))); }
    else { TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 1, false>(x, w, bias, y, N, H, W, C, stride, Ho, Wo, add_input, st))); }
    TCCT_LAUNCH_OK();
}

// dx[hi,wi] = sum over taps (ky,kx) with (hi+1-ky) % stride == 0 ... of dy[(hi+1-ky)/stride, (wi+1-kx)/stride] * w[ky,kx]
template <typename T, int VEC>
__global__ void k_dw_dgrad(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int N, int H, int W,
                           int C, int stride, int Ho, int Wo, int add_input) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % CV) * VEC;
        int64_t p = i / CV;
        int wi = (int)(p % W);
        int64_t r = p / W;
        int hi = (int)(r % H);
        int64_t n = r / H;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            int th = hi + 1 - ky;
            if (th < 0 || (th % stride) != 0) continue;
            int ho = th / stride;
            if (ho >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                int tw = wi + 1 - kx;
                if (tw < 0 || (tw % stride) != 0) continue;
                int wo = tw / stride;
                if (wo >= Wo) continue;
                float v[VEC];
                ldv<T, VEC>(dy + ((n * Ho + ho) * (int64_t)Wo + wo) * C + c, v);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    acc[k] += v[k] * w[(c + k) * 9 + ky * 3 + kx];
                    if (add_input && ky == 1 && kx == 1) acc[k] += v[k];
                }
            }
        }
        stv<T, VEC>(dx + p * C + c, acc);
    }
}

extern "C" int tcct_dwconv3x3_dgrad(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride,
                                    int add_input, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "dwconv3x3_dgrad: stride %d", stride);
    TCCT_CHECK(!(add_input && stride != 1), "dwconv3x3_dgrad: add_input needs stride 1");
    int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    int vec = (C % 4 == 0) ? 4 : 1;
    int64_t total = (int64_t)N * H * W * (C / vec);
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1) {      // dx = conv(dy, flipped taps) (+ dy when the forward added its input)
        if (vec == 4) { TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 4, true>(dy, w, nullptr, dx, N, H, W, C, 1, H, W, add_input, st))); }
        else { TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 1, true>(dy, w, nullptr, dx, N, H, W, C, 1, H, W, add_input, st))); }
        TCCT_LAUNCH_OK();
    }
    if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_dw_dgrad<T, 4>), dim3(tcct_grid(total, DB, 1 << 16)), dim3(DB), 0, st, (const T*)dy, w, (T*)dx, N, H, W, C, stride, Ho, Wo, add_input)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_dw_dgrad<T, 1>), dim3(tcct_grid(total, DB, 1 << 16)), dim3(DB), 0, st, (const T*)dy, w, (T*)dx, N, H, W, C, stride, Ho, Wo, add_input)); }
    TCCT_LAUNCH_OK();
}

// dw[c][ky][kx] = sum_p x[p@tap][c] * dy[p][c];  dbias[c] = sum_p dy[p][c].  Thread = (row slot, channel vector),
// 10*VEC register sums, LDS combine per block, fp32 atomics out.
template <typename T, int VEC, int STRIDE>
__global__ void k_dw_wgrad(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
                           int N, int H, int W, int C, int Ho, int Wo) {
    extern __shared__ float sm[];   // [DB][10*VEC]
    const int CV = C / VEC;
    const int R = DB / CV;
    const int t = threadIdx.x;
    const bool active = t < R * CV;
    const int cv = t % CV, r = t / CV;
    const int c = cv * VEC;
    const int segs = (Wo + DW_SEG - 1) / DW_SEG;
    const int64_t strips = (int64_t)N * Ho * segs;
    float acc[10][VEC];
#pragma unroll
    for (int a = 0; a < 10; ++a)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[a][k] = 0.f;
    if (active) {
        for (int64_t sidx = (int64_t)blockIdx.x * R + r; sidx < strips; sidx += (int64_t)gridDim.x * R) {
            const int sg = (int)(sidx % segs);
            int64_t q = sidx / segs;
            const int ho = (int)(q % Ho);
            const int64_t n = q / Ho;
            const int wo0 = sg * DW_SEG, wo1 = min(Wo, wo0 + DW_SEG);
            const int hi0 = ho * STRIDE - 1;
            const T* rows[3];
            bool rok[3];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                int hi = hi0 + ky;
                rok[ky] = hi >= 0 && hi < H;
                rows[ky] = x + ((n * H + (rok[ky] ? hi : 0)) * (int64_t)W) * C + c;
            }
            float col[3][3][VEC];
            auto load_col = [&](int slot, int wi) {
                bool cok = wi >= 0 && wi < W;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    if (cok && rok[ky]) ldv<T, VEC>(rows[ky] + (int64_t)wi * C, col[slot][ky]);
                    else {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) col[slot][ky][k] = 0.f;
                    }
                }
            };
            load_col(0, wo0 * STRIDE - 1); load_col(1, wo0 * STRIDE);
            for (int wo = wo0; wo < wo1; ++wo) {
                load_col(2, wo * STRIDE + 1);
                float g[VEC];
                ldv<T, VEC>(dy + ((n * Ho + ho) * (int64_t)Wo + wo) * C + c, g);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    acc[9][k] += g[k];
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx][k] += col[kx][ky][k] * g[k];
                }
                if (STRIDE == 1) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) { col[0][ky][k] = col[1][ky][k]; col[1][ky][k] = col[2][ky][k]; }
                } else {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) col[0][ky][k] = col[2][ky][k];
                    load_col(1, (wo + 1) * STRIDE);
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < 10; ++a)
#pragma unroll
        for (int k = 0; k < VEC; ++k) sm[(t * 10 + a) * VEC + k] = acc[a][k];
    __syncthreads();
    for (int o = t; o < C * 10; o += DB) {
        int ch = o / 10, a = o % 10;
        int cvv = ch / VEC, k = ch % VEC;
        float s = 0.f;
        for (int rr = 0; rr < R; ++rr) s += sm[((rr * CV + cvv) * 10 + a) * VEC + k];
        if (a < 9) atomicAdd(&dw[ch * 9 + a], s);
        else if (dbias) atomicAdd(&dbias[ch], s);
    }
}

extern "C" int tcct_dwconv3x3_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C,
                                    int stride, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "dwconv3x3_wgrad: stride %d", stride);
    TCCT_CHECK(C >= 1 && C <= DB * 4, "dwconv3x3_wgrad: C=%d", C);
    int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    int vec = (C % 4 == 0) ? 4 : 1;
    TCCT_CHECK(C / vec <= DB, "dwconv3x3_wgrad: C=%d too large", C);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * C * 9, st) != hipSuccess) { tcct_set_error("dwconv3x3_wgrad: memset failed"); return -2; }
    if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * C, st) != hipSuccess) { tcct_set_error("dwconv3x3_wgrad: memset failed"); return -2; }
    int R = DB / (C / vec);
    int64_t strips = (int64_t)N * Ho * ((Wo + DW_SEG - 1) / DW_SEG);
    int grid = tcct_grid(strips, R, 2048);
    size_t lds = sizeof(float) * DB * 10 * vec;
#define DWG(V, S) hipLaunchKernelGGL((k_dw_wgrad<T, V, S>), dim3(grid), dim3(DB), lds, st, (const T*)x, (const T*)dy, dw, dbias, N, H, W, C, Ho, Wo)
    if (vec == 4) { TCCT_DISPATCH(dtype, if (stride == 1) DWG(4, 1); else DWG(4, 2)); }
    else { TCCT_DISPATCH(dtype, if (stride == 1) DWG(1, 1); else DWG(1, 2)); }
#undef DWG
    TCCT_LAUNCH_OK();
}
