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
// the 32 x 32 block (o_off.., i_off..) of an OIHW weight with `ldi` input channels
__global__ void k_pack_w32f_sub(const float* __restrict__ w, float* __restrict__ wp, int KH, int KW, int transposed, int ldi, int o_off, int i_off) {
    const int total = KH * KW * 1024;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
        const int dy = tap / KW, dx = tap % KW;
        wp[i] = transposed ? w[(((int64_t)(o_off + ci) * ldi + i_off + co) * KH + (KH - 1 - dy)) * KW + (KW - 1 - dx)]
                           : w[(((int64_t)(o_off + co) * ldi + i_off + ci) * KH + dy) * KW + dx];
    }
}
extern "C" int tcct_conv32f_pack_weights_sub(const float* w, float* wp, int KH, int KW, int transposed, int cin_total, int o_off, int i_off,
                                             tcct_stream_t stream) {
    const int total = KH * KW * 1024;
    hipLaunchKernelGGL(k_pack_w32f_sub, dim3((total + CFB - 1) / CFB), dim3(CFB), 0, (hipStream_t)stream, w, wp, KH, KW, transposed, cin_total, o_off, i_off);
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_conv32f_pack_weights(const float* w, float* wp, int KH, int KW, int transposed, tcct_stream_t stream) {
    const int total = KH * KW * 1024;
    hipLaunchKernelGGL(k_pack_w32f, dim3((total + CFB - 1) / CFB), dim3(CFB), 0, (hipStream_t)stream, w, wp, KH, KW, transposed);
    TCCT_LAUNCH_OK();
}

template <bool VERT>
__global__ void __launch_bounds__(CFB)
k_conv32f_mfma(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* y,
               const float* yadd, int N, int H, int W, int KH, int KW, int PH, int PW, int tilesH, int tilesW, int ntiles,
               int xs, int xo, int ys, int yo) {
    // xs / xo, ys / yo: channels per pixel in memory and first channel of the 32-channel slab read / written (32, 0: plain 32-channel tensors);
    // wider convolutions run as 32 x 32 sub-GEMMs over slabs (MPViT stem[1] 32 -> 64, the wide CNN encoder), accumulating through yadd == y
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
    // the NEXT tile's pixels are requested into registers before the MFMA phase of the current one (576 MFMAs per wave and tile: long enough to
    // hide the whole memory latency) and written to LDS after it
    float4 pre[CF_MAXL];
    auto prefetch = [&](int tile) {
        const int tw = tile % tilesW, t2 = tile / tilesW, th = t2 % tilesH, n = t2 / tilesH;
        const int h0 = th * TH, w0 = tw * TW;
        const float* xb = x + (int64_t)n * H * W * xs + xo;
#pragma unroll
        for (int j = 0; j < CF_MAXL; ++j) {
            const int i = tid + j * CFB;
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / LW, lc = pl - lr * LW;
            const int hi = h0 - PH + lr, wi = w0 - PW + lc;
            pre[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pl < npix && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                pre[j] = *reinterpret_cast<const float4*>(xb + ((int64_t)hi * W + wi) * xs + c * 4);
        }
    };
    if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tilesW, t2 = tile / tilesW, th = t2 % tilesH, n = t2 / tilesH;
        const int h0 = th * TH, w0 = tw * TW;
        __syncthreads();                    // the previous tile's fragment reads are done (and, first time, the weights are staged)
#pragma unroll
        for (int j = 0; j < CF_MAXL; ++j) {
            const int i = tid + j * CFB;
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / LW, lc = pl - lr * LW;
            if (pl < npix) *reinterpret_cast<float4*>(sX + (VERT ? lc * LH + lr : pl) * CF_IPS + c * 16) = pre[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
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
                const int64_t o = (((int64_t)n * H + ho) * W + wo) * ys + yo + 4 * hh;
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
static int conv32f_fwd_impl(const float* x, const float* wp, const float* bias, const float* yadd, float* y, int N, int H, int W, int KH, int KW,
                            int PH, int PW, int xs, int xo, int ys, int yo, tcct_stream_t stream);
extern "C" int tcct_conv32f_fwd(const float* x, const float* wp, const float* bias, const float* yadd, float* y, int N, int H, int W, int KH, int KW,
                                int PH, int PW, tcct_stream_t stream) {
    TCCT_CHECK(yadd == nullptr || yadd != y, "conv32f_fwd: yadd must be a separate tensor");
    return conv32f_fwd_impl(x, wp, bias, yadd, y, N, H, W, KH, KW, PH, PW, 32, 0, 32, 0, stream);
}
/* the same on 32-channel slabs of wider fp32 tensors: x has xs channels per pixel (slab at xo), y has ys (slab at yo); accumulate = 1 adds into y.
 * wp: the packed 32 x 32 block of the weight (tcct_conv32f_pack_weights_sub) */
extern "C" int tcct_conv32f_fwd_strided(const float* x, const float* wp, const float* bias, float* y, int N, int H, int W, int KH, int KW, int PH, int PW,
                                        int xs, int xo, int ys, int yo, int accumulate, tcct_stream_t stream) {
    TCCT_CHECK(xs % 4 == 0 && xo % 4 == 0 && ys % 4 == 0 && yo % 4 == 0 && xo + 32 <= xs && yo + 32 <= ys, "conv32f_fwd_strided: bad slab");
    return conv32f_fwd_impl(x, wp, bias, accumulate ? y : nullptr, y, N, H, W, KH, KW, PH, PW, xs, xo, ys, yo, stream);
}
static int conv32f_fwd_impl(const float* x, const float* wp, const float* bias, const float* yadd, float* y, int N, int H, int W, int KH, int KW,
                            int PH, int PW, int xs, int xo, int ys, int yo, tcct_stream_t stream) {
    TCCT_CHECK(KH >= 1 && KW >= 1 && (KH == 1 || KW == 1 || (KH == 3 && KW == 3)) && KH * KW <= 13, "conv32f_fwd: kernel %dx%d unsupported", KH, KW);
    TCCT_CHECK(2 * PH == KH - 1 && 2 * PW == KW - 1, "conv32f_fwd: only 'same' padding (got pad %d,%d for %dx%d)", PH, PW, KH, KW);
    const bool vert = (KW == 1 && KH > 1);
    const int TH = vert ? 32 : 8, TW = vert ? 8 : 32;
    const int LH = TH + KH - 1, LW = TW + KW - 1;
    const size_t lds = (size_t)KH * KW * 32 * CF_IPS + (size_t)LH * LW * CF_IPS + 128;
    TCCT_CHECK(lds <= 160 * 1024, "conv32f_fwd: %dx%d needs %zu B of LDS", KH, KW, lds);
    TCCT_CHECK(LH * LW * 8 <= CF_MAXL * CFB, "conv32f_fwd: %dx%d tile image exceeds the staging slots", KH, KW);
    const int tilesH = (H + TH - 1) / TH, tilesW = (W + TW - 1) / TW;
    const int64_t nt = (int64_t)N * tilesH * tilesW;
    TCCT_CHECK(nt > 0 && nt < (1LL << 31), "conv32f_fwd: bad tile count");
    const int grid = (int)(nt < 256 ? nt : 256);
    hipStream_t st = (hipStream_t)stream;
    static bool attr[2] = {false, false};
    if (vert) {
        if (!attr[1]) { (void)hipFuncSetAttribute((const void*)k_conv32f_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[1] = true; }
        hipLaunchKernelGGL(k_conv32f_mfma<true>, dim3(grid), dim3(CFB), lds, st, x, wp, bias, y, yadd, N, H, W, KH, KW, PH, PW, tilesH, tilesW, (int)nt, xs, xo, ys, yo);
    } else {
        if (!attr[0]) { (void)hipFuncSetAttribute((const void*)k_conv32f_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr[0] = true; }
        hipLaunchKernelGGL(k_conv32f_mfma<false>, dim3(grid), dim3(CFB), lds, st, x, wp, bias, y, yadd, N, H, W, KH, KW, PH, PW, tilesH, tilesW, (int)nt, xs, xo, ys, yo);
    }
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[tap][co][ci] = sum_p dy[p][co] x[p@tap][ci]: per tap one 32 x 32 accumulator, K = pixels, two per MFMA (lane half hh takes pixel 2 j + hh):
// a lane reads dy[pixel][co = r] and x[pixel @ tap][ci = r] -- 32 consecutive floats per half wave, conflict-free plain ds_read_b32, no
// transpose needed for 4-byte elements.  Every wave keeps one accumulator per tap for its own quarter of the tile's pixels; a block accumulates
// over all its tiles in registers and ends with one fp32 atomic per weight element and wave (and the bias gradient).
template <bool VERT, int NTAP>
__global__ void __launch_bounds__(CFB)
k_conv32f_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias, int N, int H, int W,
                int KH, int KW, int PH, int PW, int tilesH, int tilesW, int ntiles, int xs, int xo, int ds, int dof, int ldi, int o_off, int i_off) {
    // xs / xo, ds / dof: channel strides / slab offsets of x and dy; the 32 x 32 block goes to rows o_off.., input channels i_off.. of a weight with ldi input channels
    constexpr int TH = VERT ? 32 : 8, TW = VERT ? 8 : 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LH = TH + KH - 1, LW = TW + KW - 1, TAPS = KH * KW;
    float* sX = reinterpret_cast<float*>(smem);                      // [LH*LW][32] (row-major; VERT: column-major pixel order)
    float* sD = sX + LH * LW * 32;                                   // [256][32] in M-tile order: pixel index = 32 mt + r
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // every wave takes ALL taps for its own two M-tiles (64 of the tile's 256 pixels): 9 or 13 accumulators -- one wave per SIMD, so the whole
    // 512-register file is available -- and the same number of MFMAs on every wave (dealing taps to waves gave wave 0 three of nine taps)
    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float bsum = 0.f;
    int poff[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        const int dy_ = t / KW, dx_ = t - dy_ * KW;
        poff[t] = t < TAPS ? (VERT ? dx_ * LH + dy_ : dy_ * LW + dx_) * 32 : 0;
    }
    const int npix = LH * LW;
    float4 prex[CF_MAXL], pred[8];          // the next tile's x / dy pixels, requested before the MFMA phase of the current tile
    auto prefetch = [&](int tile) {
        const int tw = tile % tilesW, t2 = tile / tilesW, th = t2 % tilesH, n = t2 / tilesH;
        const int h0 = th * TH, w0 = tw * TW;
        const float* xb = x + (int64_t)n * H * W * xs + xo;
        const float* db = dy + (int64_t)n * H * W * ds + dof;
#pragma unroll
        for (int j = 0; j < CF_MAXL; ++j) {
            const int i = tid + j * CFB;
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / LW, lc = pl - lr * LW;
            const int hi = h0 - PH + lr, wi = w0 - PW + lc;
            prex[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pl < npix && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                prex[j] = *reinterpret_cast<const float4*>(xb + ((int64_t)hi * W + wi) * xs + c * 4);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + j * CFB;
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / TW, lc = pl - lr * TW;              // tile-local output pixel (row-major)
            const int ho = h0 + lr, wo = w0 + lc;
            pred[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ho < H && wo < W) pred[j] = *reinterpret_cast<const float4*>(db + ((int64_t)ho * W + wo) * ds + c * 4);
        }
    };
    if ((int)blockIdx.x < ntiles) prefetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < CF_MAXL; ++j) {
            const int i = tid + j * CFB;
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / LW, lc = pl - lr * LW;
            if (pl < npix) *reinterpret_cast<float4*>(sX + (VERT ? lc * LH + lr : pl) * 32 + c * 4) = prex[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + j * CFB;
            const int pl = i >> 3, c = i & 7;
            const int lr = pl / TW, lc = pl - lr * TW;
            *reinterpret_cast<float4*>(sD + (VERT ? lc * TH + lr : pl) * 32 + c * 4) = pred[j];      // M-tile order: HORZ row-major, VERT column-major
        }
        __syncthreads();
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
        // pixel q of the tile in M-tile order: HORZ (row = q / 32, col = q % 32) -> image pixel row * LW + col; VERT (col = q / 32, row = q % 32)
        // -> image pixel col * LH + row: consecutive q inside an M-tile are consecutive image pixels in both layouts
        for (int mt = 2 * wave; mt < 2 * wave + 2; ++mt) {
            const float* dbase = sD + mt * 32 * 32;
            const float* xbase = sX + (VERT ? mt * LH : mt * LW) * 32;
#pragma unroll 2
            for (int j = 0; j < 16; ++j) {
                const int q = 2 * j + hh;
                const float a = dbase[q * 32 + r];
                bsum += a;
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    if (t < TAPS) {                     // block-uniform
                        const float b = xbase[q * 32 + poff[t] + r];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                    }
                }
            }
        }
    }
    // the four waves' partial sums are combined in LDS (the tiles are dead), wave after wave into the same slots, before ONE atomic per weight
    // element and block: with every wave adding for itself, 256 blocks x 4 waves hit each of the 9 216 addresses 1 024 times -- same-address
    // float atomics serialise at the memory side and cost a fixed ~0.3 ms per call, i.e. more than the whole kernel on the coarse levels
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                // [TAPS][16][64]
    for (int wv = 0; wv < 4; ++wv) {
        if (wave == wv) {
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                if (t < TAPS) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        float* slot = red + (t * 16 + k) * 64 + lane;
                        *slot = wv == 0 ? acc[t][k] : *slot + acc[t][k];
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < TAPS * 1024; i += CFB) {
        const int ln = i & 63, k = (i >> 6) & 15, t = i >> 10;
        const int co = (k & 3) + 8 * (k >> 2) + 4 * (ln >> 5), ci = ln & 31;
        const int dy_ = t / KW, dx_ = t - dy_ * KW;
        atomicAdd(&dw[(((int64_t)(o_off + co) * ldi + i_off + ci) * KH + dy_) * KW + dx_], red[i]);
    }
    if (dbias) {
        const float sb = bsum + __shfl_xor(bsum, 32, 64);
        if (lane < 32) atomicAdd(&dbias[o_off + r], sb);
    }
}

/* dw OIHW fp32 [32,32,KH,KW] and dbias [32] (nullable) are cleared here (unless tcct_set_outputs_prezeroed) and accumulated into */
static int conv32f_wgrad_impl(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH, int PW, int xs, int xo,
                              int ds, int dof, int ldi, int o_off, int i_off, bool clear, tcct_stream_t stream);
extern "C" int tcct_conv32f_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH, int PW,
                                  tcct_stream_t stream) {
    return conv32f_wgrad_impl(x, dy, dw, dbias, N, H, W, KH, KW, PH, PW, 32, 0, 32, 0, 32, 0, 0, true, stream);
}
/* one 32 x 32 block of a wider convolution's weight gradient: accumulates into dw [Cout][ldi][KH][KW] at (o_off, i_off) and, dbias non-NULL, into
 * dbias[o_off..] -- the caller clears dw / dbias once before the first slab */
extern "C" int tcct_conv32f_wgrad_strided(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH, int PW,
                                          int xs, int xo, int ds, int dof, int ldi, int o_off, int i_off, tcct_stream_t stream) {
    TCCT_CHECK(xs % 4 == 0 && xo % 4 == 0 && ds % 4 == 0 && dof % 4 == 0 && xo + 32 <= xs && dof + 32 <= ds, "conv32f_wgrad_strided: bad slab");
    return conv32f_wgrad_impl(x, dy, dw, dbias, N, H, W, KH, KW, PH, PW, xs, xo, ds, dof, ldi, o_off, i_off, false, stream);
}
static int conv32f_wgrad_impl(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH, int PW, int xs, int xo,
                              int ds, int dof, int ldi, int o_off, int i_off, bool clear, tcct_stream_t stream) {
    TCCT_CHECK(KH >= 1 && KW >= 1 && (KH == 1 || KW == 1 || (KH == 3 && KW == 3)) && KH * KW <= 13, "conv32f_wgrad: kernel %dx%d unsupported", KH, KW);
    TCCT_CHECK(2 * PH == KH - 1 && 2 * PW == KW - 1, "conv32f_wgrad: only 'same' padding");
    const bool vert = (KW == 1 && KH > 1);
    const int TH = vert ? 32 : 8, TW = vert ? 8 : 32;
    const int LH = TH + KH - 1, LW = TW + KW - 1;
    size_t lds = ((size_t)LH * LW + 256) * 128;
    if (lds < (size_t)KH * KW * 4096) lds = (size_t)KH * KW * 4096;             // the cross-wave reduction of the accumulators reuses the tile area
    TCCT_CHECK(lds <= 160 * 1024, "conv32f_wgrad: %dx%d needs %zu B of LDS", KH, KW, lds);
    TCCT_CHECK(LH * LW * 8 <= CF_MAXL * CFB, "conv32f_wgrad: %dx%d tile image exceeds the staging slots", KH, KW);
    hipStream_t st = (hipStream_t)stream;
    if (clear && !tcct_skip_zero_fill()) {
        if (hipMemsetAsync(dw, 0, sizeof(float) * 1024 * KH * KW, st) != hipSuccess) { tcct_set_error("conv32f_wgrad: memset failed"); return -2; }
        if (dbias && hipMemsetAsync(dbias, 0, sizeof(float) * 32, st) != hipSuccess) { tcct_set_error("conv32f_wgrad: memset failed"); return -2; }
    }
    const int tilesH = (H + TH - 1) / TH, tilesW = (W + TW - 1) / TW;
    const int64_t nt = (int64_t)N * tilesH * tilesW;
    TCCT_CHECK(nt > 0 && nt < (1LL << 31), "conv32f_wgrad: bad tile count");
    int64_t g8 = (nt + 7) / 8;                  // >= 8 tiles per block: every block ends with KH*KW*1024 same-address atomics
    const int grid = (int)(g8 < 1 ? 1 : (g8 > 256 ? 256 : g8));
#define WG_L(V, NT) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_conv32f_wgrad<V, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_conv32f_wgrad<V, NT>), dim3(grid), dim3(CFB), lds, st, x, dy, dw, dbias, N, H, W, KH, KW, PH, PW, tilesH, tilesW, (int)nt, xs, xo, ds, dof, ldi, o_off, i_off); }
    const int taps = KH * KW;
    if (vert) { if (taps <= 9) WG_L(true, 9) else WG_L(true, 13) }
    else { if (taps <= 9) WG_L(false, 9) else WG_L(false, 13) }
#undef WG_L
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ pointwise (1x1 / nn.Linear), fp32
// Y[M,N] = X[M,K] W^T + bias with fp32 rows (the parity mode of nets/tcct.py:41-43,124,532-546,600,966-997), K and N multiples of 32.
// 128-pixel tiles (32 per wave), up to PF_NTB 32-channel output tiles per block (blockIdx.y walks the rest), K in chunks of 32 staged through LDS
// (rows of 144 bytes as above).  transposed = 1: the weight is read as [K][N], i.e. the input gradient dx = dy W of the same layer.
#define PF_NTB 4
__global__ void __launch_bounds__(CFB)
k_pwf_mfma(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y, int64_t M, int K, int N,
           int transposed, int64_t tiles) {
    __shared__ __attribute__((aligned(16))) unsigned char sX[128 * CF_IPS];
    __shared__ __attribute__((aligned(16))) unsigned char sW[PF_NTB * 32 * CF_IPS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.y * PF_NTB * 32;
    const int ntb = min(PF_NTB, (N - n0) / 32);
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        f32x16 acc[PF_NTB];
#pragma unroll
        for (int t = 0; t < PF_NTB; ++t)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
        const int64_t m0 = tile * 128;
        for (int k0 = 0; k0 < K; k0 += 32) {
            __syncthreads();
            for (int i = tid; i < 128 * 8; i += CFB) {
                const int p = i >> 3, c = i & 7;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m0 + p < M) v = *reinterpret_cast<const float4*>(x + (m0 + p) * K + k0 + c * 4);
                *reinterpret_cast<float4*>(sX + p * CF_IPS + c * 16) = v;
            }
            for (int i = tid; i < ntb * 32 * 32; i += CFB) {
                const int row = i >> 5, col = i & 31;           // row = output channel n0 + row, col = input channel k0 + col
                const float v = transposed ? w[(int64_t)(k0 + col) * N + n0 + row] : w[(int64_t)(n0 + row) * K + k0 + col];
                *reinterpret_cast<float*>(sW + row * CF_IPS + col * 4) = v;
            }
            __syncthreads();
            float b[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(sX + (32 * wave + r) * CF_IPS + hh * 64 + q * 16);
                b[4 * q] = v.x; b[4 * q + 1] = v.y; b[4 * q + 2] = v.z; b[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < PF_NTB; ++t) {
                if (t < ntb) {
                    float a[16];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = *reinterpret_cast<const float4*>(sW + (32 * t + r) * CF_IPS + hh * 64 + q * 16);
                        a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
                    }
#pragma unroll
                    for (int s = 0; s < 16; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[t], 0, 0, 0);
                }
            }
        }
        const int64_t m = m0 + 32 * wave + r;
        if (m < M) {
#pragma unroll
            for (int t = 0; t < PF_NTB; ++t) {
                if (t < ntb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int co = n0 + 32 * t + 8 * q + 4 * hh;
                        float4 v = make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
                        if (bias) { v.x += bias[co]; v.y += bias[co + 1]; v.z += bias[co + 2]; v.w += bias[co + 3]; }
                        *reinterpret_cast<float4*>(y + m * N + co) = v;
                    }
                }
            }
        }
    }
}
/* y [M,N] = x [M,K] W^T + bias (fp32; w [N,K], or read as [K,N] when transposed = 1; bias nullable); K, N multiples of 32 */
extern "C" int tcct_pwf_fwd(const float* x, const float* w, const float* bias, float* y, int64_t M, int K, int N, int transposed, tcct_stream_t stream) {
    TCCT_CHECK(K % 32 == 0 && N % 32 == 0 && K >= 32 && N >= 32 && M > 0, "pwf_fwd: K=%d N=%d unsupported (multiples of 32)", K, N);
    const int64_t tiles = (M + 127) / 128;
    const int gy = (N / 32 + PF_NTB - 1) / PF_NTB;
    int64_t gx = 1024 / gy;
    if (gx > tiles) gx = tiles;
    hipLaunchKernelGGL(k_pwf_mfma, dim3((unsigned)gx, gy), dim3(CFB), 0, (hipStream_t)stream, x, w, bias, y, M, K, N, transposed, tiles);
    TCCT_LAUNCH_OK();
}

// dW[N][K] += dy^T x, dbias[N] += sum dy over 128-pixel tiles: a block owns all N/32 output-channel tiles x PF_KTB input-channel tiles
// (blockIdx.y walks the rest of K), tile i of that set belongs to wave i % 4 (<= PF_WT accumulators per wave); K-dim = pixels, two per MFMA.
#define PF_KTB 2
#define PF_WT 3
__global__ void __launch_bounds__(CFB)
k_pwf_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias, int64_t M, int K, int N,
            int64_t tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sD = reinterpret_cast<float*>(smem);                  // [128][N]
    float* sX = sD + 128 * N;                                    // [128][32 * PF_KTB]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int NT = N / 32;
    const int k0 = blockIdx.y * PF_KTB * 32;
    const int ktb = min(PF_KTB, (K - k0) / 32);
    const int XW = 32 * PF_KTB;
    const int ntl = NT * ktb;                                    // tiles of this block: index i = nt * ktb + kt
    const int S = ntl == 1 ? 4 : (ntl == 2 ? 2 : 1);             // pixel splits per tile
    f32x16 acc[PF_WT];
#pragma unroll
    for (int t = 0; t < PF_WT; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float bsum[PF_WT] = {0.f, 0.f, 0.f};
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const int64_t m0 = tile * 128;
        __syncthreads();
        for (int i = tid; i < 128 * (N / 4); i += CFB) {
            const int p = i / (N / 4), c = i - p * (N / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + p < M) v = *reinterpret_cast<const float4*>(dy + (m0 + p) * N + c * 4);
            *reinterpret_cast<float4*>(sD + p * N + c * 4) = v;
        }
        for (int i = tid; i < 128 * (ktb * 8); i += CFB) {
            const int p = i / (ktb * 8), c = i - p * (ktb * 8);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (m0 + p < M) v = *reinterpret_cast<const float4*>(x + (m0 + p) * K + k0 + c * 4);
            *reinterpret_cast<float4*>(sX + p * XW + c * 4) = v;
        }
        __syncthreads();
        // units = tiles x pixel splits: with fewer than four tiles (32 -> 32: ONE) the 128 pixels are cut into S = 4 / ntl parts so that every wave works
#pragma unroll
        for (int t = 0; t < PF_WT; ++t) {
            const int u = wave + 4 * t;
            if (u < ntl * S) {                          // wave-uniform
                const int ti = u % ntl, part = u / ntl;
                const int nt = ti / ktb, kt = ti - nt * ktb;
                const int j0 = part * (64 / S), j1 = j0 + 64 / S;
#pragma unroll 4
                for (int j = j0; j < j1; ++j) {
                    const int q = 2 * j + hh;
                    const float a = sD[q * N + 32 * nt + r];
                    const float b = sX[q * XW + 32 * kt + r];
                    if (kt == 0) bsum[t] += a;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < PF_WT; ++t) {
        const int u = wave + 4 * t;
        if (u < ntl * S) {
            const int ti = u % ntl;
            const int nt = ti / ktb, kt = ti - nt * ktb;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int co = 32 * nt + (k & 3) + 8 * (k >> 2) + 4 * hh;
                atomicAdd(&dw[(int64_t)co * K + k0 + 32 * kt + r], acc[t][k]);
            }
            if (dbias && kt == 0 && blockIdx.y == 0) {
                const float sb = bsum[t] + __shfl_xor(bsum[t], 32, 64);
                if (lane < 32) atomicAdd(&dbias[32 * nt + r], sb);
            }
        }
    }
}
/* dw [N,K] fp32 and dbias [N] (nullable) are cleared here (unless tcct_set_outputs_prezeroed) and accumulated into; N <= 160 */
extern "C" int tcct_pwf_wgrad(const float* x, const float* dy, float* dw, float* dbias, int64_t M, int K, int N, tcct_stream_t stream) {
    TCCT_CHECK(K % 32 == 0 && N % 32 == 0 && K >= 32 && N >= 32 && N <= 160 && M > 0, "pwf_wgrad: K=%d N=%d unsupported", K, N);
    TCCT_CHECK((N / 32) * PF_KTB <= 4 * PF_WT, "pwf_wgrad: N=%d needs more accumulators than a block has", N);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill()) {
        if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)N * K, st) != hipSuccess) { tcct_set_error("pwf_wgrad: memset failed"); return -2; }
        if (dbias && hipMemsetAsync(dbias, 0, sizeof(float) * N, st) != hipSuccess) { tcct_set_error("pwf_wgrad: memset failed"); return -2; }
    }
    const size_t lds = (size_t)128 * (N + 32 * PF_KTB) * 4;
    TCCT_CHECK(lds <= 160 * 1024, "pwf_wgrad: %zu B of LDS", lds);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_pwf_wgrad, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    const int64_t tiles = (M + 127) / 128;
    const int gy = (K / 32 + PF_KTB - 1) / PF_KTB;
    // no register prefetch in this kernel: several resident blocks per CU hide the load latency instead (48 KB of LDS at N = 32: three per CU;
    // one block per CU left the level-0 32 -> 32 weight gradient at 2.1 ms for 0.13 ms of MFMA work)
    int per_cu = (int)((160 * 1024) / (lds + 1024));
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    int64_t gx = 256 * per_cu / gy;
    if (gx < 32) gx = 32;
    if (gx > (tiles + 7) / 8) gx = (tiles + 7) / 8;        // >= 8 tiles per block: every block ends with same-address atomics on all of dw
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(k_pwf_wgrad, dim3((unsigned)gx, gy), dim3(CFB), lds, st, x, dy, dw, dbias, M, K, N, tiles);
    TCCT_LAUNCH_OK();
}
