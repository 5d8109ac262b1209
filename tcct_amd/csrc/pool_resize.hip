// MetaPool (3x3 box over the (token, channel) plane minus identity), MaxPool2d(2), bilinear resize (both
// align_corners conventions) and per-pixel L2 normalisation.  All HBM-bound gather stencils on NHWC, atomic-free
// (backward passes are written output-stationary).
#include "common.h"

#define PB 256

// ------------------------------------------------------------------------------------------ MetaPool
// x [B, N, C]; BWD=false: y = box3x3_validcount(x) - x ; BWD=true: dx = box^T(dy) - dy
template <typename T, bool BWD>
__global__ void k_metapool(const T* __restrict__ x, T* __restrict__ y, int B, int64_t N, int C) {
    const int C4 = C >> 2;
    const int64_t total = (int64_t)B * N * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c0 = (int)(i % C4) * 4;
        int64_t bn = i / C4;
        int64_t n = bn % N;
        const T* base = x + bn * C;            // row n
        float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f4 ctr = f4zero();
#pragma unroll
        for (int dn = -1; dn <= 1; ++dn) {
            int64_t nn = n + dn;
            if (nn < 0 || nn >= N) continue;
            const T* row = base + (int64_t)dn * C;
            float rw = 1.f;
            if (BWD) { int rn = 1 + (nn > 0) + (nn < N - 1); rw = 1.f / (float)rn; }
            f4 m = ld4(row + c0);
            float l = c0 > 0 ? ldf(row + c0 - 1) : 0.f;
            float r = c0 + 4 < C ? ldf(row + c0 + 4) : 0.f;
            if (dn == 0) ctr = m;
            float g[6] = {l, m.v[0], m.v[1], m.v[2], m.v[3], r};
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float s = rw;
                if (BWD) { int c = c0 - 1 + j; int rc = 1 + (c > 0) + (c < C - 1); s = rw / (float)rc; }
                v[j] += g[j] * s;
            }
        }
        f4 o;
        int rn = 1 + (n > 0) + (n < N - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = v[k] + v[k + 1] + v[k + 2];
            if (!BWD) { int c = c0 + k; int rc = 1 + (c > 0) + (c < C - 1); s /= (float)(rn * rc); }
            o.v[k] = s - ctr.v[k];
        }
        st4(y + bn * C + c0, o);
    }
}
extern "C" int tcct_metapool_fwd(const void* x, void* y, int B, int64_t N, int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && C >= 4, "metapool_fwd: C=%d", C);
    int64_t total = (int64_t)B * N * (C / 4);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_metapool<T, false>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, N, C));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_metapool_bwd(const void* dy, void* dx, int B, int64_t N, int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && C >= 4, "metapool_bwd: C=%d", C);
    int64_t total = (int64_t)B * N * (C / 4);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_metapool<T, true>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, B, N, C));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------ MaxPool2d(2)
template <typename T, bool BWD>
__global__ void k_maxpool2(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, int N, int H, int W, int C) {
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const int64_t total = (int64_t)N * Ho * Wo * C4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c0 = (int)(i % C4) * 4;
        int64_t p = i / C4;
        int wo = (int)(p % Wo);
        int64_t r = p / Wo;
        int ho = (int)(r % Ho);
        int64_t n = r / Ho;
        int64_t b00 = ((n * H + 2 * ho) * (int64_t)W + 2 * wo) * C + c0;
        int64_t offs[4] = {b00, b00 + C, b00 + (int64_t)W * C, b00 + (int64_t)W * C + C};
        f4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ld4(x + offs[q]);
        f4 m = v[0];
        int am[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 1; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (v[q].v[k] > m.v[k] || v[q].v[k] != v[q].v[k]) { m.v[k] = v[q].v[k]; am[k] = q; }
        if (!BWD) { st4(out + p * C + c0, m); }
        else {
            f4 g = ld4(dy + p * C + c0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) o.v[k] = am[k] == q ? g.v[k] : 0.f;
                st4(out + offs[q], o);
            }
        }
    }
}
extern "C" int tcct_maxpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool2_fwd: needs C%%4==0 and even H,W (got %d,%d,%d)", C, H, W);
    int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_maxpool2<T, false>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)nullptr, (T*)y, N, H, W, C));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_maxpool2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0, "maxpool2_bwd: needs C%%4==0 and even H,W (got %d,%d,%d)", C, H, W);
    int64_t total = (int64_t)N * (H / 2) * (W / 2) * (C / 4);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_maxpool2<T, true>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, N, H, W, C));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------ bilinear
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp src_index(int o, float scale, int in, int align) {
    float s = align ? scale * (float)o : fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
    Lerp r;
    r.i0 = min((int)s, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = s - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

template <typename T, int VEC>
__global__ void k_bilinear_fwd(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo,
                               float sh, float sw, int align) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * Ho * Wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % CV) * VEC;
        int64_t p = i / CV;
        int wo = (int)(p % Wo);
        int64_t r = p / Wo;
        int ho = (int)(r % Ho);
        int64_t n = r / Ho;
        Lerp a = src_index(ho, sh, H, align), b = src_index(wo, sw, W, align);
        const T* r0 = x + ((n * H + a.i0) * (int64_t)W) * C + c;
        const T* r1 = x + ((n * H + a.i1) * (int64_t)W) * C + c;
        float o[VEC];
        if (VEC == 4) {
            f4 v00 = ld4(r0 + (int64_t)b.i0 * C), v01 = ld4(r0 + (int64_t)b.i1 * C);
            f4 v10 = ld4(r1 + (int64_t)b.i0 * C), v11 = ld4(r1 + (int64_t)b.i1 * C);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                o[k] = a.l0 * (b.l0 * v00.v[k] + b.l1 * v01.v[k]) + a.l1 * (b.l0 * v10.v[k] + b.l1 * v11.v[k]);
            f4 t; t.v[0] = o[0]; t.v[1] = o[1]; t.v[2] = o[2]; t.v[3] = o[3];
            st4(y + p * C + c, t);
        } else {
            float v00 = ldf(r0 + (int64_t)b.i0 * C), v01 = ldf(r0 + (int64_t)b.i1 * C);
            float v10 = ldf(r1 + (int64_t)b.i0 * C), v11 = ldf(r1 + (int64_t)b.i1 * C);
            stf(y + p * C + c, a.l0 * (b.l0 * v00 + b.l1 * v01) + a.l1 * (b.l0 * v10 + b.l1 * v11));
        }
    }
}

__device__ __forceinline__ void cand_range(int i, float scale, int out, int align, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
    float a, b;
    if (align) { a = ((float)i - 1.f) / scale; b = ((float)i + 1.f) / scale; }
    else { a = ((float)i - 0.5f) / scale - 0.5f; b = ((float)i + 1.5f) / scale - 0.5f; }
    lo = max(0, (int)floorf(a) - 1);
    hi = min(out - 1, (int)ceilf(b) + 1);
}

#define BL_MAXC 40   // candidate outputs per axis (scale >= 1/16)
template <typename T, int VEC>
__global__ void k_bilinear_bwd(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo,
                               float sh, float sw, int align) {
    const int CV = C / VEC;
    const int64_t total = (int64_t)N * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % CV) * VEC;
        int64_t p = i / CV;
        int wi = (int)(p % W);
        int64_t r = p / W;
        int hi = (int)(r % H);
        int64_t n = r / H;
        int hlo, hhi, wlo, whi;
        cand_range(hi, sh, Ho, align, hlo, hhi);
        cand_range(wi, sw, Wo, align, wlo, whi);
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        for (int ho = hlo; ho <= hhi; ++ho) {
            Lerp a = src_index(ho, sh, H, align);
            float wh = (a.i0 == hi ? a.l0 : 0.f) + (a.i1 == hi ? a.l1 : 0.f);
            if (wh == 0.f) continue;
            const T* row = dy + ((n * Ho + ho) * (int64_t)Wo) * C + c;
            for (int wo = wlo; wo <= whi; ++wo) {
                Lerp b = src_index(wo, sw, W, align);
                float ww = (b.i0 == wi ? b.l0 : 0.f) + (b.i1 == wi ? b.l1 : 0.f);
                if (ww == 0.f) continue;
                float g = wh * ww;
                if (VEC == 4) {
                    f4 v = ld4(row + (int64_t)wo * C);
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] += g * v.v[k];
                } else acc[0] += g * ldf(row + (int64_t)wo * C);
            }
        }
        if (VEC == 4) { f4 t; t.v[0] = acc[0]; t.v[1] = acc[1]; t.v[2] = acc[2]; t.v[3] = acc[3]; st4(dx + p * C + c, t); }
        else stf(dx + p * C + c, acc[0]);
    }
}

extern "C" int tcct_bilinear_fwd(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int align_corners,
                                 int dtype, tcct_stream_t stream) {
    TCCT_CHECK(H > 0 && W > 0 && Ho > 0 && Wo > 0, "bilinear_fwd: bad sizes");
    float sh = align_corners ? (Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f) : (float)H / (float)Ho;
    float sw = align_corners ? (Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f) : (float)W / (float)Wo;
    int vec = (C % 4 == 0) ? 4 : 1;
    int64_t total = (int64_t)N * Ho * Wo * (C / vec);
    hipStream_t st = (hipStream_t)stream;
    if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_fwd<T, 4>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, st, (const T*)x, (T*)y, N, H, W, C, Ho, Wo, sh, sw, align_corners)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_fwd<T, 1>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, st, (const T*)x, (T*)y, N, H, W, C, Ho, Wo, sh, sw, align_corners)); }
    TCCT_LAUNCH_OK();
}
/* dy [N,Ho,Wo,C] -> dx [N,H,W,C] (H,W = forward input size) */
extern "C" int tcct_bilinear_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int align_corners,
                                 int dtype, tcct_stream_t stream) {
    TCCT_CHECK(H > 0 && W > 0 && Ho > 0 && Wo > 0, "bilinear_bwd: bad sizes");
    float sh = align_corners ? (Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f) : (float)H / (float)Ho;
    float sw = align_corners ? (Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f) : (float)W / (float)Wo;
    int vec = (C % 4 == 0) ? 4 : 1;
    int64_t total = (int64_t)N * H * W * (C / vec);
    hipStream_t st = (hipStream_t)stream;
    if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_bwd<T, 4>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, st, (const T*)dy, (T*)dx, N, H, W, C, Ho, Wo, sh, sw, align_corners)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_bwd<T, 1>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, st, (const T*)dy, (T*)dx, N, H, W, C, Ho, Wo, sh, sw, align_corners)); }
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------ L2 normalise over C
// LP = C/4 lanes per pixel (power of two <= 64)
template <typename T, bool BWD>
__global__ void k_l2norm(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, int64_t M, int C, float eps) {
    const int LP = C >> 2;
    const int64_t total = M * LP;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;     // multiple of 64 -> lanes of a pixel stay together
    const int64_t rounds = (total + stride - 1) / stride;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t it = 0; it < rounds; ++it, i += stride) {
        bool ok = i < total;
        int64_t ii = ok ? i : 0;
        f4 v = ld4(x + ii * 4);
        float ss = v.v[0] * v.v[0] + v.v[1] * v.v[1] + v.v[2] * v.v[2] + v.v[3] * v.v[3];
        f4 g = f4zero();
        float dot = 0.f;
        if (BWD) { g = ld4(dy + ii * 4); dot = v.v[0] * g.v[0] + v.v[1] * g.v[1] + v.v[2] * g.v[2] + v.v[3] * g.v[3]; }
        for (int o = LP >> 1; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); if (BWD) dot += __shfl_xor(dot, o, 64); }
        float nrm = sqrtf(ss);
        float d = fmaxf(nrm, eps);
        f4 r;
        if (!BWD) {
#pragma unroll
            for (int k = 0; k < 4; ++k) r.v[k] = v.v[k] / d;
        } else {
            // y = x/d ; dx = dy/d - x * (x.dy) / (d^2 * nrm)   when nrm > eps, else dy/eps
            float coef = nrm > eps ? dot / (d * d * nrm) : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) r.v[k] = g.v[k] / d - v.v[k] * coef;
        }
        if (ok) st4(out + ii * 4, r);
    }
}
extern "C" int tcct_l2norm_fwd(const void* x, void* y, int64_t M, int C, float eps, int dtype, tcct_stream_t stream) {
    int LP = C / 4;
    TCCT_CHECK(C % 4 == 0 && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0, "l2norm_fwd: C=%d unsupported", C);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_l2norm<T, false>), dim3(tcct_grid(M * LP, PB, 1 << 16)), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)nullptr, (T*)y, M, C, eps));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_l2norm_bwd(const void* x, const void* dy, void* dx, int64_t M, int C, float eps, int dtype, tcct_stream_t stream) {
    int LP = C / 4;
    TCCT_CHECK(C % 4 == 0 && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0, "l2norm_bwd: C=%d unsupported", C);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_l2norm<T, true>), dim3(tcct_grid(M * LP, PB, 1 << 16)), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, M, C, eps));
    TCCT_LAUNCH_OK();
}
