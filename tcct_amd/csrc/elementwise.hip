// Elementwise / layout kernels (HBM-bound; 16 B (fp32) or 8 B (bf16) per lane, grid-stride).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void tcct_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* tcct_last_error(void) { return g_err; }

// Accumulation outputs (weight/bias gradients, BN/LN partial sums) are normally cleared by the entry point itself (one
// hipMemsetAsync per call, ~270 per training step).  A caller that hands out slices of ONE buffer it clears once per step
// can switch that off.  Process-wide on purpose: torch runs backward kernels from its autograd worker thread.
static volatile int g_skip_zero = 0;
int tcct_skip_zero_fill() { return g_skip_zero; }
extern "C" int tcct_set_outputs_prezeroed(int on) { g_skip_zero = on ? 1 : 0; return 0; }
extern "C" int tcct_version(void) { return 100; }
static int64_t g_census[TCCT_CENSUS_N];
void tcct_census_hit(int which) { __atomic_fetch_add(&g_census[which], 1, __ATOMIC_RELAXED); }
/* launches of kernel family `which` (0 k_conv32_chain33, 1 k_conv32_wgradk_stream, 2 k_conv32_wgrad33_stream, 3 k_conv32_fwd33_stream) since the last reset;
 * reset != 0 clears the counter after reading.  Host-side bookkeeping only (tests assert which kernel a shape was routed to). */
extern "C" int64_t tcct_kernel_census(int which, int reset) {
    if (which < 0 || which >= TCCT_CENSUS_N) return -1;
    const int64_t v = __atomic_load_n(&g_census[which], __ATOMIC_RELAXED);
    if (reset) __atomic_store_n(&g_census[which], 0, __ATOMIC_RELAXED);
    return v;
}

#define EW_BLOCK 256

// ------------------------------------------------------------------------------------------- layout
template <typename T>
__global__ void k_image_to_nhwc4(const float* __restrict__ img, T* __restrict__ out, int N, int Csrc, int H,
                                 int Wsrc, int Wdst) {
    int64_t total = (int64_t)N * H * Wdst;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int w = (int)(i % Wdst);
        int64_t t = i / Wdst;
        int h = (int)(t % H);
        int n = (int)(t / H);
        f4 v = f4zero();
        if (w < Wsrc) {
            int64_t plane = (int64_t)H * Wsrc;
            const float* p = img + (int64_t)n * Csrc * plane + (int64_t)h * Wsrc + w;
            if (Csrc == 1) { v.v[0] = v.v[1] = v.v[2] = p[0]; }
            else { v.v[0] = p[0]; v.v[1] = p[plane]; v.v[2] = p[2 * plane]; }
        }
        st4(out + i * 4, v);
    }
}

extern "C" int tcct_image_to_nhwc4(const float* img, void* out, int N, int Csrc, int H, int Wsrc, int Wdst,
                                   int dtype, tcct_stream_t stream) {
    TCCT_CHECK(Csrc == 1 || Csrc == 3, "image_to_nhwc4: Csrc must be 1 or 3, got %d", Csrc);
    TCCT_CHECK(Wdst >= Wsrc && N > 0 && H > 0, "image_to_nhwc4: bad shape");
    int64_t total = (int64_t)N * H * Wdst;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_image_to_nhwc4<T>, dim3(tcct_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0,
                                            (hipStream_t)stream, img, (T*)out, N, Csrc, H, Wsrc, Wdst));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------- GOALS image preprocessing (uint8)
// One gather covers the reference's OpenCV/albumentations steps on B-scans and label maps (data/octnpy.py:82-85,91-112,119-130,
// data/octgen.py:9-24): row/column crop (source ROI), cv2.INTER_NEAREST resize (ROI -> dh x dw), horizontal/vertical flip, the label
// code change v*mul/div (gray level // 30 on the way in, class * 30 on the way out), and pasting into a larger canvas (destination
// offset, everything else = fill).  Nearest-neighbour index exactly as OpenCV's resizeNN: s = min(floor(d * (1.0 / (dn / sn))), sn-1)
// evaluated in double precision.  src [N,SH,SW,C] -> dst [N,DH,DW,C], both uint8 HWC (what cv2.imread returns).
__global__ void k_u8_gather2d(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int N, int SH, int SW, int C, int sy0, int sx0,
                              int sh, int sw, int DH, int DW, int dy0, int dx0, int dh, int dw, int flipy, int flipx, int mul, int div,
                              int fill) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= DW) return;
    const double ify = 1.0 / ((double)dh / (double)sh), ifx = 1.0 / ((double)dw / (double)sw);
    int rx = x - dx0;
    const bool xin = rx >= 0 && rx < dw;
    if (flipx) rx = dw - 1 - rx;
    const int sx = xin ? sx0 + min((int)floor((double)rx * ifx), sw - 1) : 0;
    for (int row = blockIdx.y; row < N * DH; row += gridDim.y) {
        const int n = row / DH, y = row - n * DH;
        int ry = y - dy0;
        const bool in = xin && ry >= 0 && ry < dh;
        if (flipy) ry = dh - 1 - ry;
        uint8_t* o = dst + ((int64_t)row * DW + x) * C;
        if (in) {
            const int sy = sy0 + min((int)floor((double)ry * ify), sh - 1);
            const uint8_t* p = src + (((int64_t)n * SH + sy) * SW + sx) * C;
            for (int c = 0; c < C; ++c) o[c] = (uint8_t)((int)p[c] * mul / div);
        } else {
            for (int c = 0; c < C; ++c) o[c] = (uint8_t)fill;
        }
    }
}
extern "C" int tcct_u8_gather2d(const uint8_t* src, uint8_t* dst, int N, int SH, int SW, int C, int sy0, int sx0, int sh, int sw, int DH,
                                int DW, int dy0, int dx0, int dh, int dw, int flipy, int flipx, int mul, int div, int fill,
                                tcct_stream_t stream) {
    TCCT_CHECK(N >= 1 && SH >= 1 && SW >= 1 && C >= 1 && C <= 4 && DH >= 1 && DW >= 1, "u8_gather2d: bad shape");
    TCCT_CHECK(sy0 >= 0 && sx0 >= 0 && sh >= 1 && sw >= 1 && sy0 + sh <= SH && sx0 + sw <= SW, "u8_gather2d: source ROI outside the image");
    TCCT_CHECK(dh >= 1 && dw >= 1 && dy0 >= 0 && dx0 >= 0 && dy0 + dh <= DH && dx0 + dw <= DW, "u8_gather2d: destination ROI outside the canvas");
    TCCT_CHECK(mul >= 1 && div >= 1 && mul <= 255 && fill >= 0 && fill <= 255, "u8_gather2d: bad value map");      // v*mul/div wraps to uint8 like numpy astype
    const int64_t rows = (int64_t)N * DH;
    dim3 g((unsigned)((DW + EW_BLOCK - 1) / EW_BLOCK), (unsigned)(rows < 4096 ? rows : 4096));
    hipLaunchKernelGGL(k_u8_gather2d, g, dim3(EW_BLOCK), 0, (hipStream_t)stream, src, dst, N, SH, SW, C, sy0, sx0, sh, sw, DH, DW, dy0, dx0,
                       dh, dw, flipy, flipx, mul, div, fill);
    TCCT_LAUNCH_OK();
}

__global__ void k_onehot_to_index(const int64_t* __restrict__ oh, uint8_t* __restrict__ lab, int N, int C, int64_t HW) {
    int64_t total = (int64_t)N * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t n = i / HW, p = i % HW;
        int best = 0;
        for (int c = 0; c < C; ++c)
            if (oh[(n * C + c) * HW + p] > 0) { best = c; break; }
        lab[i] = (uint8_t)best;
    }
}
extern "C" int tcct_onehot_to_index(const int64_t* onehot, uint8_t* lab, int N, int C, int64_t HW, tcct_stream_t stream) {
    TCCT_CHECK(C > 0 && C < 256, "onehot_to_index: bad C");
    hipLaunchKernelGGL(k_onehot_to_index, dim3(tcct_grid((int64_t)N * HW, EW_BLOCK)), dim3(EW_BLOCK), 0,
                       (hipStream_t)stream, onehot, lab, N, C, HW);
    TCCT_LAUNCH_OK();
}

__global__ void k_labels_to_u8(const int64_t* __restrict__ lab, uint8_t* __restrict__ out, int N, int H, int Wsrc, int Wdst) {
    int64_t total = (int64_t)N * H * Wdst;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int w = (int)(i % Wdst);
        int64_t r = i / Wdst;
        out[i] = w < Wsrc ? (uint8_t)lab[r * Wsrc + w] : (uint8_t)0;
    }
}
extern "C" int tcct_labels_to_u8(const int64_t* lab, uint8_t* out, int N, int H, int Wsrc, int Wdst, tcct_stream_t stream) {
    TCCT_CHECK(Wdst >= Wsrc, "labels_to_u8: Wdst < Wsrc");
    hipLaunchKernelGGL(k_labels_to_u8, dim3(tcct_grid((int64_t)N * H * Wdst, EW_BLOCK)), dim3(EW_BLOCK), 0,
                       (hipStream_t)stream, lab, out, N, H, Wsrc, Wdst);
    TCCT_LAUNCH_OK();
}

// NHWC -> NCHW fp32 through an LDS tile (32 pixels x C)
template <typename T>
__global__ void k_nhwc_to_nchw(const T* __restrict__ x, float* __restrict__ y, int64_t HW, int C) {
    extern __shared__ float tile[];   // [64][C+1]
    int n = blockIdx.y;
    int64_t p0 = (int64_t)blockIdx.x * 64;
    int np = (int)min((int64_t)64, HW - p0);
    const T* xs = x + ((int64_t)n * HW + p0) * C;
    for (int i = threadIdx.x; i < np * C; i += blockDim.x) tile[(i / C) * (C + 1) + (i % C)] = ldf(xs + i);
    __syncthreads();
    for (int i = threadIdx.x; i < np * C; i += blockDim.x) {
        int c = i / np, p = i % np;
        y[((int64_t)n * C + c) * HW + p0 + p] = tile[p * (C + 1) + c];
    }
}
extern "C" int tcct_nhwc_to_nchw_f32(const void* x, float* y, int N, int64_t HW, int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C > 0 && C <= 512, "nhwc_to_nchw: bad C %d", C);
    dim3 grid((unsigned)((HW + 63) / 64), N);
    size_t lds = (size_t)64 * (C + 1) * sizeof(float);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_nhwc_to_nchw<T>, grid, dim3(256), lds, (hipStream_t)stream,
                                            (const T*)x, y, HW, C));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------ generic vec4 maps
// OP(i, lambda over 4 lanes).  n must be a multiple of 4 for the vector body; the tail is scalar.
template <typename T, typename F>
__global__ void k_map1(const T* __restrict__ x, T* __restrict__ y, int64_t n, F f) {
    int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f4 a = ld4(x + i * 4), r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[k] = f(a.v[k]);
        st4(y + i * 4, r);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = (n4 << 2) + threadIdx.x;
        stf(y + i, f(ldf(x + i)));
    }
}
template <typename T, typename F>
__global__ void k_map2(const T* __restrict__ x, const T* __restrict__ z, T* __restrict__ y, int64_t n, F f) {
    int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f4 a = ld4(x + i * 4), b = ld4(z + i * 4), r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[k] = f(a.v[k], b.v[k]);
        st4(y + i * 4, r);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = (n4 << 2) + threadIdx.x;
        stf(y + i, f(ldf(x + i), ldf(z + i)));
    }
}
template <typename T, typename F>
__global__ void k_map3(const T* __restrict__ x, const T* __restrict__ z, const T* __restrict__ u, T* __restrict__ y,
                       int64_t n, F f) {
    int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f4 a = ld4(x + i * 4), b = ld4(z + i * 4), c = ld4(u + i * 4), r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[k] = f(a.v[k], b.v[k], c.v[k]);
        st4(y + i * 4, r);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        int64_t i = (n4 << 2) + threadIdx.x;
        stf(y + i, f(ldf(x + i), ldf(z + i), ldf(u + i)));
    }
}

struct FAct { int kind; __device__ float operator()(float x) const { return act_fwd(kind, x); } };
struct FActBwd { int kind; __device__ float operator()(float x, float dy) const { return dy * act_grad(kind, x); } };
// the activation the training step actually maps over whole tensors (Mlp: GELU behind fc1, reference nets/tcct.py:29-53) with a compile-time
// kind: act_fwd / act_grad resolve a run-time kind per element with a chain of scalar branches
struct FGelu { __device__ float operator()(float x) const { return act_c<TCCT_ACT_GELU>(x); } };
struct FGeluBwd { __device__ float operator()(float x, float dy) const { float c, p; gauss_cdf_pdf(x, c, p); return dy * (c + x * p); } };
struct FAdd { __device__ float operator()(float a, float b) const { return a + b; } };
struct FAddAct { int kind; __device__ float operator()(float a, float b) const { return act_fwd(kind, a + b); } };
struct FAddActBwd { int kind; __device__ float operator()(float a, float b, float dy) const { return dy * act_grad(kind, a + b); } };
struct FAdd3 { float alpha; __device__ float operator()(float a, float b, float c) const { return alpha * (a + b + c); } };
struct FScale { float alpha; __device__ float operator()(float a) const { return alpha * a; } };

#define EW_GRID(n) dim3(tcct_grid(((n) >> 2) + 1, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream

extern "C" int tcct_act_fwd(const void* x, void* y, int64_t n, int kind, int dtype, tcct_stream_t stream) {
    if (kind == TCCT_ACT_GELU) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map1<T, FGelu>), EW_GRID(n), (const T*)x, (T*)y, n, FGelu{})); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map1<T, FAct>), EW_GRID(n), (const T*)x, (T*)y, n, FAct{kind})); }
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int kind, int dtype, tcct_stream_t stream) {
    if (kind == TCCT_ACT_GELU) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map2<T, FGeluBwd>), EW_GRID(n), (const T*)x, (const T*)dy, (T*)dx, n, FGeluBwd{})); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map2<T, FActBwd>), EW_GRID(n), (const T*)x, (const T*)dy, (T*)dx, n, FActBwd{kind})); }
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_add(const void* a, const void* b, void* y, int64_t n, int dtype, tcct_stream_t stream) {
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map2<T, FAdd>), EW_GRID(n), (const T*)a, (const T*)b, (T*)y, n, FAdd{}));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_add_act_fwd(const void* a, const void* b, void* y, int64_t n, int kind, int dtype, tcct_stream_t stream) {
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map2<T, FAddAct>), EW_GRID(n), (const T*)a, (const T*)b, (T*)y, n, FAddAct{kind}));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_add_act_bwd(const void* a, const void* b, const void* dy, void* dx, int64_t n, int kind, int dtype,
                                tcct_stream_t stream) {
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map3<T, FAddActBwd>), EW_GRID(n), (const T*)a, (const T*)b, (const T*)dy,
                                            (T*)dx, n, FAddActBwd{kind}));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_add3_scale(const void* a, const void* b, const void* c, void* y, int64_t n, float alpha, int dtype,
                               tcct_stream_t stream) {
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map3<T, FAdd3>), EW_GRID(n), (const T*)a, (const T*)b, (const T*)c, (T*)y, n,
                                            FAdd3{alpha}));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_scale(const void* x, void* y, int64_t n, float alpha, int dtype, tcct_stream_t stream) {
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_map1<T, FScale>), EW_GRID(n), (const T*)x, (T*)y, n, FScale{alpha}));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------ per-sample scaled residual
template <typename T>
__global__ void k_residual(const T* __restrict__ x, const T* __restrict__ z, const float* __restrict__ scale,
                           T* __restrict__ y, int64_t per4, int64_t total4) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        float s = scale ? scale[i / per4] : 1.f;
        f4 b = ld4(z + i * 4), r;
        if (x) {
            f4 a = ld4(x + i * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.v[k] = a.v[k] + s * b.v[k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) r.v[k] = s * b.v[k];
        }
        st4(y + i * 4, r);
    }
}
extern "C" int tcct_residual_fwd(const void* x, const void* z, const float* scale, void* y, int B, int64_t per_sample,
                                 int dtype, tcct_stream_t stream) {
    TCCT_CHECK(per_sample % 4 == 0, "residual_fwd: per_sample %% 4 != 0");
    int64_t total4 = (int64_t)B * per_sample / 4;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_residual<T>, dim3(tcct_grid(total4, EW_BLOCK)), dim3(EW_BLOCK), 0,
                                            (hipStream_t)stream, (const T*)x, (const T*)z, scale, (T*)y, per_sample / 4, total4));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_scale_rows(const void* x, const float* scale, void* y, int B, int64_t per_sample, int dtype,
                               tcct_stream_t stream) {
    TCCT_CHECK(per_sample % 4 == 0, "scale_rows: per_sample %% 4 != 0");
    int64_t total4 = (int64_t)B * per_sample / 4;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_residual<T>, dim3(tcct_grid(total4, EW_BLOCK)), dim3(EW_BLOCK), 0,
                                            (hipStream_t)stream, (const T*)nullptr, (const T*)x, scale, (T*)y, per_sample / 4, total4));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------- concat / split
template <typename T, bool SPLIT>
__global__ void k_concat2(T* __restrict__ a, T* __restrict__ b, T* __restrict__ y, int64_t M, int Ca4, int Cb4) {
    int C4 = Ca4 + Cb4;
    int64_t total = M * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t m = i / C4;
        int c = (int)(i % C4);
        T* src = c < Ca4 ? a + (m * Ca4 + c) * 4 : b + (m * Cb4 + (c - Ca4)) * 4;
        if (SPLIT) st4(src, ld4(y + i * 4)); else st4(y + i * 4, ld4(src));
    }
}
extern "C" int tcct_concat2(const void* a, const void* b, void* y, int64_t M, int Ca, int Cb, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(Ca % 4 == 0 && Cb % 4 == 0, "concat2: channels must be multiples of 4");
    int64_t total = M * (Ca + Cb) / 4;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_concat2<T, false>), dim3(tcct_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0,
                                            (hipStream_t)stream, (T*)a, (T*)b, (T*)y, M, Ca / 4, Cb / 4));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_split2(const void* dy, void* da, void* db, int64_t M, int Ca, int Cb, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(Ca % 4 == 0 && Cb % 4 == 0, "split2: channels must be multiples of 4");
    int64_t total = M * (Ca + Cb) / 4;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_concat2<T, true>), dim3(tcct_grid(total, EW_BLOCK)), dim3(EW_BLOCK), 0,
                                            (hipStream_t)stream, (T*)da, (T*)db, (T*)dy, M, Ca / 4, Cb / 4));
    TCCT_LAUNCH_OK();
}

__global__ void k_axpy_f32(const float* __restrict__ x, float* __restrict__ y, int64_t n, float alpha) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] += alpha * x[i];
}
extern "C" int tcct_axpy_f32(const float* x, float* y, int64_t n, float alpha, tcct_stream_t stream) {
    hipLaunchKernelGGL(k_axpy_f32, dim3(tcct_grid(n, EW_BLOCK)), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, y, n, alpha);
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------- streaming-copy yardstick
// z = (a1[c] y1 + b1[c]) + (a2[c] y2 + b2[c]): the encoder fusion f_j = BN(tran_vit(v)) + BN(tran_cnn(c)) (SimpleFusion, reference nets/tcct.py:1016-1024) with
// BOTH train-mode BatchNorms applied in one pass (round 4: before, one normalisation pass wrote BN(tran_vit(v)) and the second read it back as a residual).
// ab1, ab2 = {a[C], b[C]} of the two BatchNorms; bf16: 8 channels per thread (C % 8 == 0), fp32: 4.
template <typename T, int VEC>
__global__ void k_affine2_add(const T* __restrict__ y1, const float* __restrict__ ab1, const T* __restrict__ y2, const float* __restrict__ ab2, T* __restrict__ z,
                              int64_t nvec, int C) {
    const int CV = C / VEC;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const int c0 = (int)(i % CV) * VEC;
        float u[VEC], v[VEC], o[VEC];
        ldv<VEC>(y1 + i * VEC, u); ldv<VEC>(y2 + i * VEC, v);
#pragma unroll
        for (int k = 0; k < VEC; ++k) o[k] = (ab1[c0 + k] * u[k] + ab1[C + c0 + k]) + (ab2[c0 + k] * v[k] + ab2[C + c0 + k]);
        stv<VEC>(z + i * VEC, o);
    }
}
extern "C" int tcct_affine2_add(const void* y1, const float* ab1, const void* y2, const float* ab2, void* z, int64_t M, int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(y1 && ab1 && y2 && ab2 && z && M > 0 && C >= 8 && C % 8 == 0, "affine2_add: NULL argument or C=%d (multiple of 8)", C);
    const int64_t n = M * C;
    if (dtype == TCCT_BF16) hipLaunchKernelGGL((k_affine2_add<bf16, 8>), dim3(tcct_grid(n / 8, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, (const bf16*)y1, ab1,
                                               (const bf16*)y2, ab2, (bf16*)z, n / 8, C);
    else hipLaunchKernelGGL((k_affine2_add<float, 4>), dim3(tcct_grid(n / 4, 256, 256 * 32)), dim3(256), 0, (hipStream_t)stream, (const float*)y1, ab1,
                            (const float*)y2, ab2, (float*)z, n / 4, C);
    TCCT_LAUNCH_OK();
}

// The copy rate the streaming kernels of this library are compared with (bench.py `roofline.copy_ceiling`): 16-byte accesses, one block-contiguous
// 8 KB chunk per block, a grid as large as the tensor (tools/probe/stream_probe.hip: 6.0 TB/s on the MI355X against 5.3 for persistent blocks and
// 4.9 for hipMemcpyDtoD; MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).
__global__ void __launch_bounds__(256) k_stream_copy(const uint4* __restrict__ x, uint4* __restrict__ y, int64_t n16) {
    const int64_t base = (int64_t)blockIdx.x * 512 + threadIdx.x;
    uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
    if (base < n16) v0 = x[base];
    if (base + 256 < n16) v1 = x[base + 256];
    if (base < n16) y[base] = v0;
    if (base + 256 < n16) y[base + 256] = v1;
}
// Attribution marker (tools/attrib_trace.py): an empty launch whose GRID SIZE carries an id.  rocprofv3's kernel trace and PMC tables record the grid of every
// dispatch but no arguments; a marker in front of each C-ABI call of a traced step cuts the dispatch sequence into calls, and the Python side knows
// which call (symbol, shapes, module scope) carries which id.  Measurement tooling only: nothing in the product path launches it.
__global__ void __launch_bounds__(64) k_marker() {}
extern "C" int tcct_marker(int id, tcct_stream_t stream) {
    TCCT_CHECK(id >= 1 && id < (1 << 20), "marker: id %d", id);
    hipLaunchKernelGGL(k_marker, dim3((unsigned)id), dim3(64), 0, (hipStream_t)stream);
    TCCT_LAUNCH_OK();
}
/* dst[0..nbytes) = src[0..nbytes), nbytes a multiple of 16, both 16-byte aligned: the achievable-bandwidth yardstick (no reference counterpart) */
extern "C" int tcct_stream_copy(const void* src, void* dst, int64_t nbytes, tcct_stream_t stream) {
    TCCT_CHECK(nbytes > 0 && nbytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "stream_copy: %lld bytes / alignment", (long long)nbytes);
    const int64_t n16 = nbytes / 16, blocks = (n16 + 511) / 512;
    TCCT_CHECK(blocks < 0x7fffffffLL, "stream_copy: too large");
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst, n16);
    TCCT_LAUNCH_OK();
}
