// Dense conv2d on NHWC activations, fp32-accumulating VALU formulation (generic: any KHxKW, stride, Cin%4==0,
// any Cout).  v1 kernels: correctness-first, used for every dense conv / Linear of the path; the MFMA
// implicit-GEMM kernels for the 32->32 cross/3x3 convolutions live in conv_mfma.hip and take precedence there.
//   fwd  : thread = (pixel slot, group of 8 output channels); weights staged K-chunk-wise in LDS as
//          [k = tap*Cin+ci][cout] fp32 (gathered from the OIHW parameter, optionally flipped+transposed for dgrad)
//   wgrad: thread = (group of 4 k, group of 8 cout) register tile, loop over a pixel range, fp32 atomics out.
#include "common.h"

#define CB 256
#define COT 8      // output channels per thread
#define PPT 4      // pixels per thread (fwd)
#define WLDS_FLOATS 8192

template <typename Tin, typename Tout>
__global__ void __launch_bounds__(CB)
k_conv_fwd(const Tin* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, Tout* __restrict__ y,
           int N, int H, int W, int Cin, int Cin_w, int Cout, int KH, int KW, int stride, int padh, int padw, int Ho,
           int Wo, int transposed, int KC, int CoutC) {
    // CoutC: output channels handled per block (multiple of 8, <= 256); blockIdx.y selects the channel chunk
    __shared__ __attribute__((aligned(16))) float wl[WLDS_FLOATS];
    const int CG = CoutC / COT;
    const int CoutP = CoutC;
    const int co_base = blockIdx.y * CoutC;
    const int PIXB = CB / CG;
    const int t = threadIdx.x;
    const int cg = t % CG, pl = t / CG;
    const bool tactive = pl < PIXB;
    const int64_t NP = (int64_t)N * Ho * Wo;
    const int64_t tile0 = (int64_t)blockIdx.x * (PIXB * PPT);
    const int Ktot = KH * KW * Cin;

    int hb[PPT], wb[PPT];
    int64_t nb[PPT];
    bool pv[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int64_t p = tile0 + pl + (int64_t)PIXB * j;
        pv[j] = tactive && p < NP;
        int64_t pp = pv[j] ? p : 0;
        int wo = (int)(pp % Wo);
        int64_t r = pp / Wo;
        int ho = (int)(r % Ho);
        nb[j] = (r / Ho) * (int64_t)H * W;
        hb[j] = ho * stride - padh;
        wb[j] = wo * stride - padw;
    }
    float acc[PPT][COT];
#pragma unroll
    for (int j = 0; j < PPT; ++j)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[j][c] = 0.f;

    for (int k0 = 0; k0 < Ktot; k0 += KC) {
        const int kc = min(KC, Ktot - k0);
        __syncthreads();
        for (int i = t; i < kc * CoutP; i += CB) {
            int kk = i / CoutP, co = co_base + i % CoutP;
            int k = k0 + kk;
            int tap = k / Cin, ci = k % Cin;
            int dy = tap / KW, dx = tap % KW;
            float v = 0.f;
            if (co < Cout && ci < Cin_w) {
                if (!transposed) v = w[(((int64_t)co * Cin_w + ci) * KH + dy) * KW + dx];
                else v = w[(((int64_t)ci * Cout + co) * KH + (KH - 1 - dy)) * KW + (KW - 1 - dx)];
            }
            wl[i] = v;
        }
        __syncthreads();
        for (int kk = 0; kk < kc; kk += 4) {
            const int k = k0 + kk;
            const int tap = k / Cin, ci = k % Cin;
            const int dy = tap / KW, dx = tap % KW;
            f4 xv[PPT];
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                int hi = hb[j] + dy, wi = wb[j] + dx;
                bool ok = pv[j] && hi >= 0 && hi < H && wi >= 0 && wi < W;
                xv[j] = ok ? ld4(x + (nb[j] + (int64_t)hi * W + wi) * Cin + ci) : f4zero();
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 w0 = *reinterpret_cast<const float4*>(&wl[(kk + q) * CoutP + cg * COT]);
                const float4 w1 = *reinterpret_cast<const float4*>(&wl[(kk + q) * CoutP + cg * COT + 4]);
#pragma unroll
                for (int j = 0; j < PPT; ++j) {
                    float a = xv[j].v[q];
                    acc[j][0] += a * w0.x; acc[j][1] += a * w0.y; acc[j][2] += a * w0.z; acc[j][3] += a * w0.w;
                    acc[j][4] += a * w1.x; acc[j][5] += a * w1.y; acc[j][6] += a * w1.z; acc[j][7] += a * w1.w;
                }
            }
        }
    }
    const int co0 = co_base + cg * COT;
    float bv[COT];
#pragma unroll
    for (int c = 0; c < COT; ++c) bv[c] = (bias && co0 + c < Cout) ? bias[co0 + c] : 0.f;
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        if (!pv[j]) continue;
        int64_t p = tile0 + pl + (int64_t)PIXB * j;
        Tout* yp = y + p * Cout + co0;
        if ((Cout & 3) == 0 && co0 + COT <= Cout) {
            f4 a, b;
#pragma unroll
            for (int c = 0; c < 4; ++c) { a.v[c] = acc[j][c] + bv[c]; b.v[c] = acc[j][4 + c] + bv[4 + c]; }
            st4(yp, a); st4(yp + 4, b);
        } else {
#pragma unroll
            for (int c = 0; c < COT; ++c)
                if (co0 + c < Cout) stf(yp + c, acc[j][c] + bv[c]);
        }
    }
}

// 1x1 conv whose *input* channel count is not a multiple of 4 (only the dgrad of the 32->5 aux heads): thread = fixed
// vector of 8 output channels (no per-element index division, 16-byte bf16 stores), pixels strided over the grid two at a
// time; the <= 16 x Cout weights sit in LDS as [ci][co] so a lane reads its taps of one input channel with two ds_read_b128.
template <typename Tout> __device__ __forceinline__ void st8(Tout* p, const float* a);
template <> __device__ __forceinline__ void st8<float>(float* p, const float* a) {
    *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(a[4], a[5], a[6], a[7]);
}
template <> __device__ __forceinline__ void st8<bf16>(bf16* p, const float* a) {
    uint4 v;
    v.x = pack_bf16x2(a[0], a[1]); v.y = pack_bf16x2(a[2], a[3]); v.z = pack_bf16x2(a[4], a[5]); v.w = pack_bf16x2(a[6], a[7]);
    *reinterpret_cast<uint4*>(p) = v;
}
template <typename Tin, typename Tout>
__global__ void k_conv1x1_smallcin(const Tin* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                   Tout* __restrict__ y, int64_t NP, int Cin, int Cout, int transposed) {
    __shared__ __attribute__((aligned(16))) float sw[16 * 256];
    for (int i = threadIdx.x; i < Cin * Cout; i += blockDim.x) {
        int ci = i / Cout, co = i - ci * Cout;
        sw[i] = transposed ? w[i] : w[co * Cin + ci];
    }
    __syncthreads();
    const int C8 = Cout >> 3, R = CB / C8, t = threadIdx.x;
    if (t >= R * C8) return;
    const int co = (t % C8) * 8, r = t / C8;
    float b8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) b8[c] = bias ? bias[co + c] : 0.f;
    const int64_t step = (int64_t)gridDim.x * R;
    for (int64_t p = (int64_t)blockIdx.x * R + r; p < NP; p += 2 * step) {
        const int64_t p2 = p + step;
        const bool ok2 = p2 < NP;
        float a[8], a2[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) a[c] = a2[c] = b8[c];
        const Tin* x1 = x + p * Cin;
        const Tin* x2 = x + (ok2 ? p2 : p) * Cin;
        for (int ci = 0; ci < Cin; ++ci) {
            const float xv = ldf(x1 + ci), xv2 = ldf(x2 + ci);
            const float4 w0 = *reinterpret_cast<const float4*>(sw + ci * Cout + co);
            const float4 w1 = *reinterpret_cast<const float4*>(sw + ci * Cout + co + 4);
            const float wv[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int c = 0; c < 8; ++c) { a[c] += xv * wv[c]; a2[c] += xv2 * wv[c]; }
        }
        st8<Tout>(y + p * Cout + co, a);
        if (ok2) st8<Tout>(y + p2 * Cout + co, a2);
    }
}

static int conv_fwd_launch(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin,
                           int Cin_w, int Cout, int KH, int KW, int stride, int padh, int padw, int transposed,
                           int in_dtype, int out_dtype, hipStream_t st, const char* who) {
    if (Cin % 4 != 0) {
        if (!(KH == 1 && KW == 1 && stride == 1 && padh == 0 && padw == 0 && Cin_w == Cin && Cout % 8 == 0 && Cin <= 16 && Cout <= 256)) {
            tcct_set_error("%s: Cin=%d not a multiple of 4 is only supported for small 1x1 convs", who, Cin);
            return -1;
        }
        int64_t NPs = (int64_t)N * H * W;
        dim3 g(tcct_grid(NPs, 2 * (CB / (Cout / 8)), 256 * 16)), b(CB);
#define LAUNCHS(TI, TO) hipLaunchKernelGGL((k_conv1x1_smallcin<TI, TO>), g, b, 0, st, (const TI*)x, w, bias, (TO*)y, NPs, Cin, Cout, transposed)
        if (in_dtype == TCCT_F32 && out_dtype == TCCT_F32) LAUNCHS(float, float);
        else if (in_dtype == TCCT_BF16 && out_dtype == TCCT_BF16) LAUNCHS(bf16, bf16);
        else if (in_dtype == TCCT_BF16 && out_dtype == TCCT_F32) LAUNCHS(bf16, float);
        else if (in_dtype == TCCT_F32 && out_dtype == TCCT_BF16) LAUNCHS(float, bf16);
        else { tcct_set_error("%s: bad dtypes", who); return -1; }
#undef LAUNCHS
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) { tcct_set_error("%s: launch failed: %s", who, hipGetErrorString(e0)); return -2; }
        return 0;
    }
    if (!(Cin % 4 == 0 && Cin_w <= Cin && Cin_w > 0 && Cout > 0 && Cout <= 1024 && stride >= 1)) {
        tcct_set_error("%s: unsupported shape Cin=%d Cin_w=%d Cout=%d stride=%d", who, Cin, Cin_w, Cout, stride);
        return -1;
    }
    int Ho = (H + 2 * padh - KH) / stride + 1, Wo = (W + 2 * padw - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || N <= 0) { tcct_set_error("%s: empty output", who); return -1; }
    int nchunk = (Cout + 255) / 256;
    int CoutC = (((Cout + nchunk - 1) / nchunk) + COT - 1) / COT * COT;
    int CG = CoutC / COT, PIXB = CB / CG;
    int KC = (WLDS_FLOATS / CoutC) & ~3;
    int64_t NP = (int64_t)N * Ho * Wo;
    int64_t tiles = (NP + PIXB * PPT - 1) / (PIXB * PPT);
    if (tiles > 0x7fffffffLL) { tcct_set_error("%s: grid too large", who); return -1; }
    dim3 grid((unsigned)tiles, (unsigned)nchunk), block(CB);
#define LAUNCH(TI, TO) hipLaunchKernelGGL((k_conv_fwd<TI, TO>), grid, block, 0, st, (const TI*)x, w, bias, (TO*)y, N, H, W, Cin, Cin_w, Cout, KH, KW, stride, padh, padw, Ho, Wo, transposed, KC, CoutC)
    if (in_dtype == TCCT_F32 && out_dtype == TCCT_F32) LAUNCH(float, float);
    else if (in_dtype == TCCT_BF16 && out_dtype == TCCT_BF16) LAUNCH(bf16, bf16);
    else if (in_dtype == TCCT_BF16 && out_dtype == TCCT_F32) LAUNCH(bf16, float);
    else if (in_dtype == TCCT_F32 && out_dtype == TCCT_BF16) LAUNCH(float, bf16);
    else { tcct_set_error("%s: bad dtypes %d,%d", who, in_dtype, out_dtype); return -1; }
#undef LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { tcct_set_error("%s: launch failed: %s", who, hipGetErrorString(e)); return -2; }
    return 0;
}

extern "C" int tcct_conv2d_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin,
                               int Cin_w, int Cout, int KH, int KW, int stride, int padh, int padw, int in_dtype,
                               int out_dtype, tcct_stream_t stream) {
    return conv_fwd_launch(x, w, bias, y, N, H, W, Cin, Cin_w, Cout, KH, KW, stride, padh, padw, 0, in_dtype, out_dtype,
                           (hipStream_t)stream, "conv2d_fwd");
}

extern "C" int tcct_conv2d_dgrad(const void* dy, const float* w, void* dx, int N, int H, int W, int Cin, int Cout, int KH,
                                 int KW, int padh, int padw, int dy_dtype, int dx_dtype, tcct_stream_t stream) {
    // dx = conv(dy, flip(w)^T) with pad' = K-1-pad; as a forward conv: "Cin" := Cout, "Cout" := Cin
    return conv_fwd_launch(dy, w, nullptr, dx, N, H, W, /*Cin*/ Cout, /*Cin_w*/ Cout, /*Cout*/ Cin, KH, KW, 1, KH - 1 - padh,
                           KW - 1 - padw, 1, dy_dtype, dx_dtype, (hipStream_t)stream, "conv2d_dgrad");
}

// ---------------------------------------------------------------------------------------------- wgrad
template <typename Tx, typename Tdy>
__global__ void __launch_bounds__(CB)
k_conv_wgrad(const Tx* __restrict__ x, const Tdy* __restrict__ dy, float* __restrict__ dw, int N, int H, int W, int Cin,
             int Cin_w, int Cout, int KH, int KW, int stride, int padh, int padw, int Ho, int Wo, int64_t prange) {
    const int CG = (Cout + COT - 1) / COT;
    const int KGB = CB / CG;                    // k-groups (of 4) per block
    const int t = threadIdx.x;
    const int cg = t % CG, kgl = t / CG;
    const int KG = KH * KW * Cin / 4;
    const int kg = blockIdx.y * KGB + kgl;
    const bool active = kgl < KGB && kg < KG;
    const int k = (active ? kg : 0) * 4;
    const int tap = k / Cin, ci = k % Cin;
    const int dyy = tap / KW - padh, dxx = tap % KW - padw;
    const int co0 = cg * COT;
    const int64_t NP = (int64_t)N * Ho * Wo;
    const int64_t p0 = (int64_t)blockIdx.x * prange;
    const int64_t p1 = min(NP, p0 + prange);
    float acc[4][COT];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < COT; ++c) acc[i][c] = 0.f;
    int wo = (int)(p0 % Wo);
    int64_t r = p0 / Wo;
    int ho = (int)(r % Ho);
    int64_t n = r / Ho;
    const bool vec_dy = ((Cout & 3) == 0) && (co0 + COT <= Cout);
    if (active) {
        for (int64_t p = p0; p < p1; ++p) {
            int hi = ho * stride + dyy, wi = wo * stride + dxx;
            if (hi >= 0 && hi < H && wi >= 0 && wi < W) {
                f4 xv = ld4(x + ((n * H + hi) * (int64_t)W + wi) * Cin + ci);
                float d[COT];
                const Tdy* dp = dy + p * Cout + co0;
                if (vec_dy) {
                    f4 a = ld4(dp), b = ld4(dp + 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { d[c] = a.v[c]; d[4 + c] = b.v[c]; }
                } else {
#pragma unroll
                    for (int c = 0; c < COT; ++c) d[c] = (co0 + c < Cout) ? ldf(dp + c) : 0.f;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int c = 0; c < COT; ++c) acc[i][c] += xv.v[i] * d[c];
            }
            if (++wo == Wo) { wo = 0; if (++ho == Ho) { ho = 0; ++n; } }
        }
        const int ky = tap / KW, kx = tap % KW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (ci + i >= Cin_w) continue;
#pragma unroll
            for (int c = 0; c < COT; ++c)
                if (co0 + c < Cout)
                    atomicAdd(&dw[(((int64_t)(co0 + c) * Cin_w + ci + i) * KH + ky) * KW + kx], acc[i][c]);
        }
    }
}

template <typename T>
__global__ void k_colsum(const T* __restrict__ x, int64_t M, int C, float* __restrict__ out) {
    // out[c] += sum_m x[m][c]; thread t handles channel t % C, row slot t / C
    __shared__ float sm[CB];
    const int R = CB / C;
    const int t = threadIdx.x, c = t % C, r = t / C;
    float s = 0.f;
    if (r < R)
        for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += (int64_t)gridDim.x * R) s += ldf(x + m * C + c);
    sm[t] = s;
    __syncthreads();
    if (t < C) {
        float a = 0.f;
        for (int rr = 0; rr < R; ++rr) a += sm[rr * C + t];
        atomicAdd(&out[t], a);
    }
}

// the same for C > CB channels: thread t owns channels t, t + CB, ...
template <typename T>
__global__ void k_colsum_wide(const T* __restrict__ x, int64_t M, int C, float* __restrict__ out) {
    for (int c = threadIdx.x; c < C; c += CB) {
        float s = 0.f;
        for (int64_t m = blockIdx.x; m < M; m += gridDim.x) s += ldf(x + m * C + c);
        atomicAdd(&out[c], s);
    }
}

extern "C" int tcct_conv2d_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int Cin,
                                 int Cin_w, int Cout, int KH, int KW, int stride, int padh, int padw, int x_dtype,
                                 int dy_dtype, tcct_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    TCCT_CHECK(Cin % 4 == 0 && Cin_w <= Cin && Cout > 0 && Cout <= 1024, "conv2d_wgrad: unsupported shape Cin=%d Cout=%d", Cin, Cout);
    int Ho = (H + 2 * padh - KH) / stride + 1, Wo = (W + 2 * padw - KW) / stride + 1;
    TCCT_CHECK(Ho > 0 && Wo > 0 && N > 0, "conv2d_wgrad: empty output");
    int64_t NP = (int64_t)N * Ho * Wo;
    size_t wbytes = sizeof(float) * (size_t)Cout * Cin_w * KH * KW;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, wbytes, st) != hipSuccess) { tcct_set_error("conv2d_wgrad: memset failed"); return -2; }
    int CG = (Cout + COT - 1) / COT, KGB = CB / CG, KG = KH * KW * Cin / 4;
    int gy = (KG + KGB - 1) / KGB;
    int64_t prange = 1024;
    int64_t gx = (NP + prange - 1) / prange;
    while (gx > 8192) { prange *= 2; gx = (NP + prange - 1) / prange; }
    dim3 grid((unsigned)gx, (unsigned)gy), block(CB);
#define LAUNCH(TX, TD) hipLaunchKernelGGL((k_conv_wgrad<TX, TD>), grid, block, 0, st, (const TX*)x, (const TD*)dy, dw, N, H, W, Cin, Cin_w, Cout, KH, KW, stride, padh, padw, Ho, Wo, prange)
    if (x_dtype == TCCT_F32 && dy_dtype == TCCT_F32) LAUNCH(float, float);
    else if (x_dtype == TCCT_BF16 && dy_dtype == TCCT_BF16) LAUNCH(bf16, bf16);
    else if (x_dtype == TCCT_BF16 && dy_dtype == TCCT_F32) LAUNCH(bf16, float);
    else if (x_dtype == TCCT_F32 && dy_dtype == TCCT_BF16) LAUNCH(float, bf16);
    else { tcct_set_error("conv2d_wgrad: bad dtypes"); return -1; }
#undef LAUNCH
    if (dbias) {
        if (!tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * Cout, st) != hipSuccess) { tcct_set_error("conv2d_wgrad: memset failed"); return -2; }
        if (Cout > CB) {
            int gw = tcct_grid(NP, 1, 512);
            if (dy_dtype == TCCT_F32) hipLaunchKernelGGL(k_colsum_wide<float>, dim3(gw), dim3(CB), 0, st, (const float*)dy, NP, Cout, dbias);
            else hipLaunchKernelGGL(k_colsum_wide<bf16>, dim3(gw), dim3(CB), 0, st, (const bf16*)dy, NP, Cout, dbias);
            TCCT_LAUNCH_OK();
        }
        int R = CB / Cout;
        int g = tcct_grid(NP, R, 2048);
        if (dy_dtype == TCCT_F32) hipLaunchKernelGGL(k_colsum<float>, dim3(g), dim3(CB), 0, st, (const float*)dy, NP, Cout, dbias);
        else hipLaunchKernelGGL(k_colsum<bf16>, dim3(g), dim3(CB), 0, st, (const bf16*)dy, NP, Cout, dbias);
    }
    TCCT_LAUNCH_OK();
}
