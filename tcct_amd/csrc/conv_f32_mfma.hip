// fp32 MFMA convolution for the PARITY mode (compute_dtype = float32): dense 32 -> 32 channels, stride 1, 'same' padding, any KH x KW
// (3x3, 1xk, kx1; reference nets/tcct.py:808-822,892,978), fp32 NHWC in and out, fp32 products and fp32 accumulation on the matrix pipes:
//   v_mfma_f32_32x32x2_f32, 157 TFLOP/s dense on MI355X -- 16x below the bf16 rate, but ~5x above what the VALU convolution of conv.hip
//   reaches, so the mode in which the 1e-3 contract against the reference is asserted no longer runs at 41 B-scans/s.
// Same GEMM view as conv_mfma.hip: per tap D[co][pixel] += W[tap][co][ci] X[pixel@tap][ci], A = weights (rows = co), B = pixels: a lane owns
// one pixel and 16 of its output channels.  The K order of an MFMA step is free as long as A and B agree: step s of lane half hh takes input
// channel 16 hh + s, so a lane reads 16 CONTIGUOUS floats of its pixel / weight row per tap (four ds_read_b128) instead of 16 strided words.
// LDS rows are 144 bytes (128 + 16 pad: 8 consecutive lanes hit 32 distinct banks with 16-byte reads).  Tiles are 8 M-tiles of 32 pixels
// (8 rows x 32 columns; kx1: 32 rows x 8 columns, column-major image), two per wave; weights are staged once per block.
// Correctness-first kernels: no BatchNorm-statistics epilogues (the fp32 mode runs tcct_bn_stats), one block per CU.
#include "common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CFB 256
#define CF_IPS 144          // LDS bytes per pixel / per weight row
#define CF_MAXL 12          // staging slots per thread: (8 + 12) x 32 ... (8 + 2) x (32 + 12) pixels x 8 chunks / 256

// OIHW fp32 -> [tap][co][ci] fp32 (transposed = 1: flipped taps, co/ci swapped -- the weights of the input-gradient convolution)
__global__ void k_pack_w32f(const float* __restrict__ w, float* __restrict__ wp, int KH, int KW, int transposed) {
    const int total = KH * KW * 1024;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
        const int dy = tap / KW, dx = tap % KW;
        wp[i] = transposed ? w[(((int64_t)ci * 32 + co) * KH + (KH - 1 - dy)) * KW + (KW - 1 - dx)] : w[(((int64_t)co * 32 + ci) * KH + dy) * KW + dx];
    }
}
extern "C" int tcct_conv32f_pack_weights(const float* w, float* wp, int KH, int KW, int transposed, tcct_stream_t stream) {
    const int total = KH * KW * 1024;
    hipLaunchKernelGGL(k_pack_w32f, dim3((total + CFB - 1) / CFB), dim3(CFB), 0, (hipStream_t)stream, w, wp, KH, KW, transposed);
    TCCT_LAUNCH_OK();
}

template <bool VERT>
__global__ void __launch_bounds__(CFB)
k_conv32f_mfma(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
               const float* __restrict__ yadd, int N, int H, int W, int KH, int KW, int PH, int PW, int tilesH, int tilesW, int ntiles) {
    constexpr int TH = VERT ? 32 : 8, TW = VERT ? 8 : 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LH = TH + KH - 1, LW = TW + KW - 1, TAPS = KH * KW;
    unsigned char* sW = smem;
    unsigned char* sX = smem + TAPS * 32 * CF_IPS;
    float* sB = reinterpret_cast<float*>(sX + LH * LW * CF_IPS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    for (int i = tid; i < TAPS * 32 * 8; i += CFB) {        // weight rows: 32 floats = 8 chunks of 16 bytes
        const int row = i >> 3, c = i & 7;
        *reinterpret_cast<float4*>(sW + row * CF_IPS + c * 16) = reinterpret_cast<const float4*>(wp)[i];
    }
    if (tid < 32) sB[tid] = bias ? bias[tid] : 0.f;
    const int npix = LH * LW;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tilesW, t2 = tile / tilesW, th = t2 % tilesH, n = t2 / tilesH;
        const int h0 = th * TH, w0 = tw * TW;
        __syncthreads();                    // the previous tile's fragment reads are done (and, first time, the weights are staged)
        const float* xb = x + (int64_t)n * H * W * 32;
        for (int i = tid; i < npix * 8; i += CFB) {
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / LW, lc = pl - lr * LW;
            const int hi = h0 - PH + lr, wi = w0 - PW + lc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = *reinterpret_cast<const float4*>(xb + ((int64_t)hi * W + wi) * 32 + c * 4);
            *reinterpret_cast<float4*>(sX + (VERT ? lc * LH + lr : pl) * CF_IPS + c * 16) = v;
        }
        __syncthreads();
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
        // M-tile mt = 2 wave + t: HORZ row mt of the tile (pixels = its 32 columns); VERT column mt (pixels = its 32 rows)
        const unsigned char* xB[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int mt = 2 * wave + t;
            xB[t] = sX + (VERT ? mt * LH + r : mt * LW + r) * CF_IPS + hh * 64;
        }
        const unsigned char* wA = sW + r * CF_IPS + hh * 64;
        for (int dy = 0; dy < KH; ++dy)
            for (int dx = 0; dx < KW; ++dx) {
                const unsigned char* wa = wA + (dy * KW + dx) * 32 * CF_IPS;
                const int poff = (VERT ? dx * LH + dy : dy * LW + dx) * CF_IPS;
                float a[16], b0[16], b1[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 va = *reinterpret_cast<const float4*>(wa + q * 16);
                    const float4 v0 = *reinterpret_cast<const float4*>(xB[0] + poff + q * 16);
                    const float4 v1 = *reinterpret_cast<const float4*>(xB[1] + poff + q * 16);
                    a[4 * q] = va.x; a[4 * q + 1] = va.y; a[4 * q + 2] = va.z; a[4 * q + 3] = va.w;
                    b0[4 * q] = v0.x; b0[4 * q + 1] = v0.y; b0[4 * q + 2] = v0.z; b0[4 * q + 3] = v0.w;
                    b1[4 * q] = v1.x; b1[4 * q + 1] = v1.y; b1[4 * q + 2] = v1.z; b1[4 * q + 3] = v1.w;
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b0[s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b1[s], acc[1], 0, 0, 0);
                }
            }
        // epilogue: lane owns pixel r of each M-tile and output channels co = 8 q + 4 hh + {0..3}: four 16-byte stores
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int mt = 2 * wave + t;
            const int ho = VERT ? h0 + r : h0 + mt, wo = VERT ? w0 + mt : w0 + r;
            if (ho < H && wo < W) {
                const int64_t o = (((int64_t)n * H + ho) * W + wo) * 32 + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(acc[t][4 * q] + sB[8 * q + 4 * hh], acc[t][4 * q + 1] + sB[8 * q + 4 * hh + 1],
                                           acc[t][4 * q + 2] + sB[8 * q + 4 * hh + 2], acc[t][4 * q + 3] + sB[8 * q + 4 * hh + 3]);
                    if (yadd) { const float4 u = *reinterpret_cast<const float4*>(yadd + o + 8 * q); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
                    *reinterpret_cast<float4*>(y + o + 8 * q) = v;
                }
            }
        }
    }
}

/* x, y: fp32 NHWC [N,H,W,32]; wp: packed fp32 [KH*KW][32][32] (tcct_conv32f_pack_weights); stride 1, 'same' padding (2 PH = KH - 1, 2 PW = KW - 1);
 * yadd (nullable, fp32 like y, may not overlap y): y = conv + bias + yadd -- the input gradient of a convolution whose input has a second consumer */
extern "C" int tcct_conv32f_fwd(const float* x, const float* wp, const float* bias, const float* yadd, float* y, int N, int H, int W, int KH, int KW,
                                int PH, int PW, tcct_stream_t stream) {
    TCCT_CHECK(KH >= 1 && KW >= 1 && (KH == 1 || KW == 1 || (KH == 3 && KW == 3)) && KH * KW <= 13, "conv32f_fwd: kernel %dx%d unsupported", KH, KW);
    TCCT_CHECK(2 * PH == KH - 1 && 2 * PW == KW - 1, "conv32f_fwd: only 'same' padding (got pad %d,%d for %dx%d)", PH, PW, KH, KW);
    TCCT_CHECK(yadd == nullptr || yadd != y, "conv32f_fwd: yadd must be a separate tensor");
    const bool vert = (KW == 1 && KH > 1);
    const int TH = vert ? 32 : 8, TW = vert ? 8 : 32;
    const int LH = TH + KH - 1, LW = TW + KW - 1;
    const size_t lds = (size_t)KH * KW * 32 * CF_IPS + (size_t)LH * LW * CF_IPS + 128;
    TCCT_CHECK(lds <= 160 * 1024, "conv32f_fwd: %dx%d needs %zu B of LDS", KH, KW, lds);
    const int tilesH = (H + TH - 1) / TH, tilesW = (W + TW - 1) / TW;
    const int64_t nt = (int64_t)N * tilesH * tilesW;
    TCCT_CHECK(nt > 0 && nt < (1LL << 31), "conv32f_fwd: bad tile count");
    const int grid = (int)(nt < 256 ? nt : 256);
    hipStream_t st = (hipStream_t)stream;
    static bool attr[2] = {false, false};
    if (vert) {
        if (!attr[1]) { (void)hipFuncSetAttribute((const void*)k_conv32f_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[1] = true; }
        hipLaunchKernelGGL(k_conv32f_mfma<true>, dim3(grid), dim3(CFB), lds, st, x, wp, bias, y, yadd, N, H, W, KH, KW, PH, PW, tilesH, tilesW, (int)nt);
    } else {
        if (!attr[0]) { (void)hipFuncSetAttribute((const void*)k_conv32f_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[0] = true; }
        hipLaunchKernelGGL(k_conv32f_mfma<false>, dim3(grid), dim3(CFB), lds, st, x, wp, bias, y, yadd, N, H, W, KH, KW, PH, PW, tilesH, tilesW, (int)nt);
    }
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[tap][co][ci] = sum_p dy[p][co] x[p@tap][ci]: per tap one 32 x 32 accumulator, K = pixels, two per MFMA (lane half hh takes pixel 2 j + hh):
// a lane reads dy[pixel][co = r] and x[pixel @ tap][ci = r] -- 32 consecutive floats per half wave, conflict-free plain ds_read_b32, no
// transpose needed for 4-byte elements.  Taps are dealt to the four waves round-robin (<= 4 accumulators per wave); every block accumulates
// over all its tiles in registers and ends with one fp32 atomic per weight element (and wave 0 with the bias gradient).
template <bool VERT>
__global__ void __launch_bounds__(CFB)
k_conv32f_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias, int N, int H, int W,
                int KH, int KW, int PH, int PW, int tilesH, int tilesW, int ntiles) {
    constexpr int TH = VERT ? 32 : 8, TW = VERT ? 8 : 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LH = TH + KH - 1, LW = TW + KW - 1, TAPS = KH * KW;
    float* sX = reinterpret_cast<float*>(smem);                      // [LH*LW][32] (row-major; VERT: column-major pixel order)
    float* sD = sX + LH * LW * 32;                                   // [256][32] in M-tile order: pixel index = 32 mt + r
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float bsum = 0.f;
    int poff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tap = wave + 4 * t;
        const int dy_ = tap / KW, dx_ = tap - dy_ * KW;
        poff[t] = tap < TAPS ? (VERT ? dx_ * LH + dy_ : dy_ * LW + dx_) * 32 : 0;
    }
    const int npix = LH * LW;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tilesW, t2 = tile / tilesW, th = t2 % tilesH, n = t2 / tilesH;
        const int h0 = th * TH, w0 = tw * TW;
        __syncthreads();
        const float* xb = x + (int64_t)n * H * W * 32;
        const float* db = dy + (int64_t)n * H * W * 32;
        for (int i = tid; i < npix * 8; i += CFB) {
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / LW, lc = pl - lr * LW;
            const int hi = h0 - PH + lr, wi = w0 - PW + lc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) v = *reinterpret_cast<const float4*>(xb + ((int64_t)hi * W + wi) * 32 + c * 4);
            *reinterpret_cast<float4*>(sX + (VERT ? lc * LH + lr : pl) * 32 + c * 4) = v;
        }
        for (int i = tid; i < TH * TW * 8; i += CFB) {
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / TW, lc = pl - lr * TW;              // tile-local output pixel (row-major)
            const int ho = h0 + lr, wo = w0 + lc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ho < H && wo < W) v = *reinterpret_cast<const float4*>(db + ((int64_t)ho * W + wo) * 32 + c * 4);
            *reinterpret_cast<float4*>(sD + (VERT ? lc * TH + lr : pl) * 32 + c * 4) = v;      // M-tile order: HORZ row-major, VERT column-major
        }
        __syncthreads();
        // pixel q of the tile in M-tile order: HORZ (row = q / 32, col = q % 32) -> image pixel row * LW + col; VERT (col = q / 32, row = q % 32)
        // -> image pixel col * LH + row: consecutive q inside an M-tile are consecutive image pixels in both layouts
        for (int mt = 0; mt < 8; ++mt) {
            const float* dbase = sD + mt * 32 * 32;
            const float* xbase = sX + (VERT ? mt * LH : mt * LW) * 32;
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const int q = 2 * j + hh;
                const float a = dbase[q * 32 + r];
                if (wave == 0) bsum += a;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if (wave + 4 * t < TAPS) {          // wave-uniform
                        const float b = xbase[q * 32 + poff[t] + r];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int tap = wave + 4 * t;
        if (tap < TAPS) {
            const int dy_ = tap / KW, dx_ = tap - dy_ * KW;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int co = (k & 3) + 8 * (k >> 2) + 4 * hh;
                atomicAdd(&dw[(((int64_t)co * 32 + r) * KH + dy_) * KW + dx_], acc[t][k]);
            }
        }
    }
    if (dbias && wave == 0) {
        const float sb = bsum + __shfl_xor(bsum, 32, 64);
        if (lane < 32) atomicAdd(&dbias[r], sb);
    }
}

/* dw OIHW fp32 [32,32,KH,KW] and dbias [32] (nullable) are cleared here (unless tcct_set_outputs_prezeroed) and accumulated into */
extern "C" int tcct_conv32f_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH, int PW,
                                  tcct_stream_t stream) {
    TCCT_CHECK(KH >= 1 && KW >= 1 && (KH == 1 || KW == 1 || (KH == 3 && KW == 3)) && KH * KW <= 13, "conv32f_wgrad: kernel %dx%d unsupported", KH, KW);
    TCCT_CHECK(2 * PH == KH - 1 && 2 * PW == KW - 1, "conv32f_wgrad: only 'same' padding");
    const bool vert = (KW == 1 && KH > 1);
    const int TH = vert ? 32 : 8, TW = vert ? 8 : 32;
    const int LH = TH + KH - 1, LW = TW + KW - 1;
    const size_t lds = ((size_t)LH * LW + 256) * 128;
    TCCT_CHECK(lds <= 160 * 1024, "conv32f_wgrad: %dx%d needs %zu B of LDS", KH, KW, lds);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill()) {
        if (hipMemsetAsync(dw, 0, sizeof(float) * 1024 * KH * KW, st) != hipSuccess) { tcct_set_error("conv32f_wgrad: memset failed"); return -2; }
        if (dbias && hipMemsetAsync(dbias, 0, sizeof(float) * 32, st) != hipSuccess) { tcct_set_error("conv32f_wgrad: memset failed"); return -2; }
    }
    const int tilesH = (H + TH - 1) / TH, tilesW = (W + TW - 1) / TW;
    const int64_t nt = (int64_t)N * tilesH * tilesW;
    TCCT_CHECK(nt > 0 && nt < (1LL << 31), "conv32f_wgrad: bad tile count");
    const int grid = (int)(nt < 256 ? nt : 256);
    static bool attr[2] = {false, false};
    if (vert) {
        if (!attr[1]) { (void)hipFuncSetAttribute((const void*)k_conv32f_wgrad<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[1] = true; }
        hipLaunchKernelGGL(k_conv32f_wgrad<true>, dim3(grid), dim3(CFB), lds, st, x, dy, dw, dbias, N, H, W, KH, KW, PH, PW, tilesH, tilesW, (int)nt);
    } else {
        if (!attr[0]) { (void)hipFuncSetAttribute((const void*)k_conv32f_wgrad<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[0] = true; }
        hipLaunchKernelGGL(k_conv32f_wgrad<false>, dim3(grid), dim3(CFB), lds, st, x, dy, dw, dbias, N, H, W, KH, KW, PH, PW, tilesH, tilesW, (int)nt);
    }
    TCCT_LAUNCH_OK();
}
