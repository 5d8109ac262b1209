// MetaPool (3x3 box over the (token, channel) plane minus identity), MaxPool2d(2), bilinear resize (both
// align_corners conventions) and per-pixel L2 normalisation.  All HBM-bound gather stencils on NHWC, atomic-free
// (backward passes are written output-stationary).
#include "common.h"
#include <cstdlib>

#define PB 256

// ------------------------------------------------------------------------------------------ MetaPool
// x [B, N, C]; BWD=false: y = box3x3_validcount(x) - x ; BWD=true: dx = box^T(dy) - dy
// thread = fixed vector of 4 channels (the per-column divisors are computed once), rows strided over grid.x, batch on grid.y:
// no per-element index division (the first version spent most of its time in 64-bit div/mod).
// res / scale (both nullable): forward y = res + scale[b] * (box(x) - x) -- the token-mixer branch of MHCABlock with its residual add
// and DropPath scale folded in (reference nets/tcct.py:464-465); backward dx = scale[b] * (box^T(dy) - dy) (the residual itself
// passes dy through unchanged).
// Round 3: marching window.  A thread owns a channel vector and a STRIP of consecutive tokens, keeps the weighted rows n-1, n, n+1 in registers
// and loads every row once (the first version loaded the three rows of every token afresh -- L1 hits, but three requests per element: 0.20 ms
// for the 452 MB of stage 0, 2.3 TB/s).  All lanes of a wave run the same trip count (tokens beyond N load nothing and store nothing), so the
// lane shuffles that fetch the channel neighbours stay convergent.
#define MP_STRIP 32
template <typename T, bool BWD>
__global__ void k_metapool(const T* __restrict__ x, T* __restrict__ y, int B, int N, int C, const T* __restrict__ res,
                           const float* __restrict__ scale) {
    const int C4 = C >> 2, R = PB / C4, t = threadIdx.x;
    if (t >= R * C4) return;
    const int c0 = (t % C4) * 4, r = t / C4, lane = t & 63;
    float cs[6];                 // BWD: 1/rc of source columns c0-1..c0+4 ; FWD: 1/rc of output columns c0..c0+3 in cs[0..3]
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        int c = BWD ? c0 - 1 + j : c0 + j;
        int rc = 1 + (c > 0) + (c < C - 1);
        cs[j] = 1.f / (float)rc;
    }
    const bool hasl = c0 > 0, hasr = c0 + 4 < C;
    const T* xb = x + (int64_t)blockIdx.y * N * C + c0;
    T* yb = y + (int64_t)blockIdx.y * N * C + c0;
    const T* rb = res ? res + (int64_t)blockIdx.y * N * C + c0 : nullptr;
    const float sc = scale ? scale[blockIdx.y] : 1.f;
    struct Row { float g[6]; f4 ctr; };
    auto load_row = [&](int nn) {           // the six weighted values (left neighbour, 4 own channels, right neighbour) of token nn; zeros outside
        Row o;
        const bool in = nn >= 0 && nn < N;
        const T* row = xb + (int64_t)(in ? nn : 0) * C;
        f4 m = f4zero();
        if (in) m = ld4(row);
        const float ls = __shfl_up(m.v[3], 1, 64), rs = __shfl_down(m.v[0], 1, 64);
        const float l = (hasl && in) ? (lane == 0 ? ldf(row - 1) : ls) : 0.f;
        const float rr = (hasr && in) ? (lane == 63 ? ldf(row + 4) : rs) : 0.f;
        float rw = 1.f;
        if (BWD) { const int rn = 1 + (nn > 0) + (nn < N - 1); rw = 1.f / (float)rn; }
        const float g[6] = {l, m.v[0], m.v[1], m.v[2], m.v[3], rr};
#pragma unroll
        for (int j = 0; j < 6; ++j) o.g[j] = in ? g[j] * (BWD ? rw * cs[j] : 1.f) : 0.f;
        o.ctr = m;
        return o;
    };
    const int n0 = (blockIdx.x * R + r) * MP_STRIP;
    Row prev = load_row(n0 - 1), cur = load_row(n0);
    for (int i = 0; i < MP_STRIP; ++i) {
        const int n = n0 + i;
        const Row nxt = load_row(n + 1);
        f4 o;
        const int rn = 1 + (n > 0) + (n < N - 1);
        const float rinv = 1.f / (float)rn;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s_ = (prev.g[k] + cur.g[k] + nxt.g[k]) + (prev.g[k + 1] + cur.g[k + 1] + nxt.g[k + 1]) + (prev.g[k + 2] + cur.g[k + 2] + nxt.g[k + 2]);
            if (!BWD) s_ *= rinv * cs[k];
            o.v[k] = sc * (s_ - cur.ctr.v[k]);
        }
        if (n < N) {
            if (rb) {
                const f4 e = ld4(rb + (int64_t)n * C);
#pragma unroll
                for (int k = 0; k < 4; ++k) o.v[k] += e.v[k];
            }
            st4(yb + (int64_t)n * C, o);
        }
        prev = cur;
        cur = nxt;
    }
}
static int metapool_launch(const void* x, void* y, int B, int64_t N, int C, int dtype, bool bwd, tcct_stream_t stream, const char* who,
                           const void* res = nullptr, const float* scale = nullptr) {
    if (!(C % 4 == 0 && C >= 4 && C / 4 <= PB)) { tcct_set_error("%s: C=%d", who, C); return -1; }
    if (!(B >= 1 && B <= 65535 && N >= 1 && N < (1LL << 30))) { tcct_set_error("%s: B=%d N=%lld out of range", who, B, (long long)N); return -1; }
    const int R = PB / (C / 4);
    const int64_t strips = (N + MP_STRIP - 1) / MP_STRIP;
    dim3 g((unsigned)((strips + R - 1) / R), (unsigned)B);
    if (bwd) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_metapool<T, true>), g, dim3(PB), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, (int)N, C, (const T*)res, scale)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_metapool<T, false>), g, dim3(PB), 0, (hipStream_t)stream, (const T*)x, (T*)y, B, (int)N, C, (const T*)res, scale)); }
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_metapool_fwd(const void* x, void* y, int B, int64_t N, int C, int dtype, tcct_stream_t stream) {
    return metapool_launch(x, y, B, N, C, dtype, false, stream, "metapool_fwd");
}
extern "C" int tcct_metapool_bwd(const void* dy, void* dx, int B, int64_t N, int C, int dtype, tcct_stream_t stream) {
    return metapool_launch(dy, dx, B, N, C, dtype, true, stream, "metapool_bwd");
}
/* y = res + scale[b] * metapool(x) (scale fp32 [B] nullable = 1): token mixer + DropPath scale + residual add in one pass */
extern "C" int tcct_metapool_residual_fwd(const void* x, const void* res, const float* scale, void* y, int B, int64_t N, int C, int dtype,
                                          tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr, "metapool_residual_fwd: res is NULL");
    return metapool_launch(x, y, B, N, C, dtype, false, stream, "metapool_residual_fwd", res, scale);
}
/* dx = scale[b] * metapool^T(dy): the input gradient of the mixer branch (the residual branch receives dy itself) */
extern "C" int tcct_metapool_scaled_bwd(const void* dy, const float* scale, void* dx, int B, int64_t N, int C, int dtype, tcct_stream_t stream) {
    return metapool_launch(dy, dx, B, N, C, dtype, true, stream, "metapool_scaled_bwd", nullptr, scale);
}

// ------------------------------------------------------------------------------------------ MaxPool2d(2)
// grid.y strides over output rows (n, ho); threads of a row cover (wo, channel vector): 32-bit index arithmetic only
template <typename T, bool BWD>
__global__ void k_maxpool2(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, int N, int H, int W, int C,
                           const T* __restrict__ res) {
    // BWD with res != NULL: dx = scatter(dy) + res -- the gradient that reaches x through its other consumers (the encoder level
    // also feeds the fusion / decoder skip, tcct.py:880-883,1012-1031) is added here instead of in a separate pass
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Wo * C4) return;
    const int wo = i / C4, c0 = (i - wo * C4) * 4;
    for (int row = blockIdx.y; row < N * Ho; row += gridDim.y) {
        const int n = row / Ho, ho = row - n * Ho;
        const int64_t p = (int64_t)row * Wo + wo;
        int64_t b00 = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + c0;
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
                if (res) {
                    const f4 rv = ld4(res + offs[q]);
#pragma unroll
                    for (int k = 0; k < 4; ++k) o.v[k] += rv.v[k];
                }
                st4(out + offs[q], o);
            }
        }
    }
}
// grid.x covers one row of (pixel, channel-vector) items, grid.y strides over rows: about 8192 blocks in total, so every block
// loops over several rows (one row per block left the resize kernels bound by block dispatch, not by HBM)
static inline dim3 row_grid(int per_row, int64_t rows, int target_blocks = 8192) {
    const int gx = (per_row + PB - 1) / PB;
    int64_t gy = (target_blocks + gx - 1) / gx;
    if (gy > rows) gy = rows;
    if (gy > 65535) gy = 65535;
    if (gy < 1) gy = 1;
    return dim3((unsigned)gx, (unsigned)gy);
}
extern "C" int tcct_maxpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2 && N >= 1, "maxpool2_fwd: needs C%%4==0 and even H,W (got %d,%d,%d)", C, H, W);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_maxpool2<T, false>), row_grid((W / 2) * (C / 4), (int64_t)N * (H / 2), 1 << 22), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)nullptr, (T*)y, N, H, W, C, (const T*)nullptr));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_maxpool2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2 && N >= 1, "maxpool2_bwd: needs C%%4==0 and even H,W (got %d,%d,%d)", C, H, W);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_maxpool2<T, true>), row_grid((W / 2) * (C / 4), (int64_t)N * (H / 2), 1 << 22), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, N, H, W, C, (const T*)nullptr));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_maxpool2_bwd_add(const void* x, const void* dy, const void* res, void* dx, int N, int H, int W, int C, int dtype,
                                     tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2 && N >= 1, "maxpool2_bwd_add: needs C%%4==0 and even H,W (got %d,%d,%d)", C, H, W);
    TCCT_CHECK(res != nullptr, "maxpool2_bwd_add: res is NULL");
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_maxpool2<T, true>), row_grid((W / 2) * (C / 4), (int64_t)N * (H / 2), 1 << 22), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, N, H, W, C, (const T*)res));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------ bilinear
// grid.y strides over output rows (n, ho) -- the row interpolation is block-uniform; threads of a row cover (wo, channel vector)
template <typename T, int VEC>
__global__ void k_bilinear_fwd(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo,
                               float sh, float sw, int align, const T* __restrict__ res) {
    // res != NULL: y = resize(x) + res  (decoder skip connection, tcct.py:908-912, without a separate add pass)
    // every input row feeds ~2 * scale output rows: with the hardware's block order they are fetched by as many different L2s (PMC: 1 282 MB read
    // for 750 MB of distinct input at the four decoder levels); xcd_band gives every XCD a contiguous band of output rows
    const int CV = C / VEC;
    const unsigned lb = xcd_band(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bx = (int)(lb % gridDim.x), by = (int)(lb / gridDim.x);
    const int i = bx * blockDim.x + threadIdx.x;
    if (i >= Wo * CV) return;
    const int wo = i / CV, c = (i - wo * CV) * VEC;
    const Lerp b = src_index(wo, sw, W, align);
    const int o0 = b.i0 * C + c, o1 = b.i1 * C + c;
    for (int row = by; row < N * Ho; row += gridDim.y) {
        const int n = row / Ho, ho = row - n * Ho;
        const Lerp a = src_index(ho, sh, H, align);
        const T* r0 = x + ((int64_t)n * H + a.i0) * W * C;
        const T* r1 = x + ((int64_t)n * H + a.i1) * W * C;
        T* yo = y + ((int64_t)row * Wo + wo) * C + c;
        if (VEC == 8) {             // bf16, 16 bytes per lane: half the load / store instructions of the 4-element form
            float v00[8], v01[8], v10[8], v11[8], t[8], e[8];
            ldv<8>(r0 + o0, v00); ldv<8>(r0 + o1, v01); ldv<8>(r1 + o0, v10); ldv<8>(r1 + o1, v11);
            if (res) ldv<8>(res + ((int64_t)row * Wo + wo) * C + c, e);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                t[k] = a.l0 * (b.l0 * v00[k] + b.l1 * v01[k]) + a.l1 * (b.l0 * v10[k] + b.l1 * v11[k]);
                if (res) t[k] += e[k];
            }
            stv<8>(yo, t);
        } else if (VEC == 4) {
            f4 v00 = ld4(r0 + o0), v01 = ld4(r0 + o1), v10 = ld4(r1 + o0), v11 = ld4(r1 + o1), t;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                t.v[k] = a.l0 * (b.l0 * v00.v[k] + b.l1 * v01.v[k]) + a.l1 * (b.l0 * v10.v[k] + b.l1 * v11.v[k]);
            if (res) {
                const f4 e = ld4(res + ((int64_t)row * Wo + wo) * C + c);       // one rounding for resize + add
#pragma unroll
                for (int k = 0; k < 4; ++k) t.v[k] += e.v[k];
            }
            st4(yo, t);
        } else {
            float v00 = ldf(r0 + o0), v01 = ldf(r0 + o1), v10 = ldf(r1 + o0), v11 = ldf(r1 + o1);
            float v = a.l0 * (b.l0 * v00 + b.l1 * v01) + a.l1 * (b.l0 * v10 + b.l1 * v11);
            if (res) v += ldf(res + ((int64_t)row * Wo + wo) * C + c);
            stf(yo, v);
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

// Tiled backward (the one the step uses): block = one DH x DW tile of dx.  The outputs that interpolate from a given input row
// (column) and their weights are found ONCE per tile row (column) and kept as small tables in LDS; every dx element then
// gathers exactly its contributing dy elements (rows x columns of the two tables) -- no per-element candidate search, no
// atomics.  The plain gather kernel above searches ~8x8 candidates with a float->int conversion each per dx element.
#define BT_DH 8
template <typename T, int VEC>
__global__ void __launch_bounds__(PB) k_bilinear_bwd_tab(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C,
                                                         int Ho, int Wo, float sh, float sw, int align, int DW, int KT,
                                                         int tilesW, int tilesH) {
    extern __shared__ int smem_i[];             // idx[(DH+DW)][KT], then weights[(DH+DW)][KT], then counts[DH+DW]
    int* tidx = smem_i;
    float* twt = reinterpret_cast<float*>(smem_i + (BT_DH + DW) * KT);
    int* tcnt = smem_i + 2 * (BT_DH + DW) * KT;
    const int CV = C / VEC, t = threadIdx.x;
    int bid = blockIdx.x;
    const int tw = bid % tilesW; bid /= tilesW;
    const int th = bid % tilesH;
    const int n = bid / tilesH;
    const int hi0 = th * BT_DH, wi0 = tw * DW;
    if (t < BT_DH + DW) {
        const bool isrow = t < BT_DH;
        const int i = isrow ? hi0 + t : wi0 + (t - BT_DH);
        const int in = isrow ? H : W, out = isrow ? Ho : Wo;
        const float sc = isrow ? sh : sw;
        int cnt = 0;
        if (i < in) {
            int lo, hi;
            cand_range(i, sc, out, align, lo, hi);
            for (int o = lo; o <= hi && cnt < KT; ++o) {
                Lerp a = src_index(o, sc, in, align);
                float wgt = (a.i0 == i ? a.l0 : 0.f) + (a.i1 == i ? a.l1 : 0.f);
                if (wgt != 0.f) { tidx[t * KT + cnt] = o; twt[t * KT + cnt] = wgt; ++cnt; }
            }
        }
        tcnt[t] = cnt;
    }
    __syncthreads();
    const T* g = dy + (int64_t)n * Ho * Wo * C;
    T* out = dx + (int64_t)n * H * W * C;
    for (int i = t; i < BT_DH * DW * CV; i += PB) {
        const int cv = i % CV, pix = i / CV;
        const int r = pix / DW, cc = pix - r * DW;
        const int hi = hi0 + r, wi = wi0 + cc;
        if (hi >= H || wi >= W) continue;
        const int nr = tcnt[r], nc = tcnt[BT_DH + cc];
        const int* ri = tidx + r * KT;
        const float* rw = twt + r * KT;
        const int* ci = tidx + (BT_DH + cc) * KT;
        const float* cw = twt + (BT_DH + cc) * KT;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        T* o = out + ((int64_t)hi * W + wi) * C + cv * VEC;
        if (nr <= 4 && nc <= 4) {      // (VEC == 1 as well: the n_class-channel fp32 logits of the level-0 head, round 4)
            // x2 upsampling: at most 4 x 4 contributing outputs.  Fixed trip counts with zero-weight padding: the 16 loads are independent and
            // all in flight together; the run-time loops below wait for every load before the next one is issued (1.96 TB/s at level 0)
            int rI[4], cI[4]; float rW[4], cW[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                rI[a] = a < nr ? ri[a] : ri[0]; rW[a] = a < nr ? rw[a] : 0.f;
                cI[a] = a < nc ? ci[a] : ci[0]; cW[a] = a < nc ? cw[a] : 0.f;
            }
            float v[4][4][VEC];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) ldv<VEC>(g + ((int64_t)rI[a] * Wo + cI[b]) * C + cv * VEC, v[a][b]);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float gw = rW[a] * cW[b];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += gw * v[a][b][k];
                }
            stv<VEC>(o, acc);
            continue;
        }
        for (int a = 0; a < nr; ++a) {
            const T* row = g + (int64_t)ri[a] * Wo * C + cv * VEC;
            const float wh = rw[a];
            for (int b = 0; b < nc; ++b) {
                const float gw = wh * cw[b];
                if (VEC >= 4) {
                    float v[VEC];
                    ldv<VEC>(row + (int64_t)ci[b] * C, v);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += gw * v[k];
                } else acc[0] += gw * ldf(row + (int64_t)ci[b] * C);
            }
        }
        if (VEC >= 4) stv<VEC>(o, acc);
        else stf(o, acc[0]);
    }
}

// Exact x2, align_corners = False (the level-0 head's resize; an align_corners = True resize has position-dependent taps and keeps the gather above) -- round 6.  The transposed interpolation is separable with the fixed taps
// {1/4, 3/4, 3/4, 1/4} over dy columns 2 wi - 1 .. 2 wi + 2 (rows alike; at the borders the outer tap vanishes and its inner neighbour weighs 1).  The tiled gather above
// reads 16 dy vectors per dx vector (every dy element four times: L1 / L2 hits, but the CU takes only so many load instructions: 2.1-3.3 TB/s); here a lane owns low-resolution
// column wi (NCH channels) of a strip of BX_SH output rows, loads ITS two dy columns of the 2 BX_SH + 2 rows the strip needs -- all requests issued before the first use --
// and takes columns 2 wi - 1 / 2 wi + 2 from the adjacent lanes (`__shfl`; a lane at the edge of the wave's pixel run fetches that one column itself).  Every dy element
// is loaded once (+ the strip halo rows).  The sums are formed column-first: ((1/4 l + 3/4 a) + 3/4 b) + 1/4 r per dy row, then the same over four rows -- a different
// association than the gather's row-major sum of 16 products (fp32 accumulation either way: ~1e-7 relative).
#define BX_SH 4
#define BX_ROWS (2 * BX_SH + 2)
template <typename T, int NCH> struct BxRaw;           // NCH channels of one dy pixel as loaded
template <int NCH> struct BxRaw<float, NCH> {
    float v[NCH];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int k = 0; k < NCH; ++k) v[k] = 0.f;
    }
    __device__ __forceinline__ void load(const float* p) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) v[k] = p[k];
    }
    __device__ __forceinline__ float get(int k) const { return v[k]; }
    __device__ __forceinline__ BxRaw shfl(int delta, bool up) const {
        BxRaw r;
#pragma unroll
        for (int k = 0; k < NCH; ++k) r.v[k] = up ? __shfl_up(v[k], delta, 64) : __shfl_down(v[k], delta, 64);
        return r;
    }
};
template <> struct BxRaw<bf16, 8> {
    uint4 q;
    __device__ __forceinline__ void zero() { q = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void load(const bf16* p) { q = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ float get(int k) const {
        const uint32_t w = k < 2 ? q.x : (k < 4 ? q.y : (k < 6 ? q.z : q.w));
        return (k & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
    }
    __device__ __forceinline__ BxRaw shfl(int delta, bool up) const {
        BxRaw r;
        r.q.x = up ? __shfl_up(q.x, delta, 64) : __shfl_down(q.x, delta, 64);
        r.q.y = up ? __shfl_up(q.y, delta, 64) : __shfl_down(q.y, delta, 64);
        r.q.z = up ? __shfl_up(q.z, delta, 64) : __shfl_down(q.z, delta, 64);
        r.q.w = up ? __shfl_up(q.w, delta, 64) : __shfl_down(q.w, delta, 64);
        return r;
    }
};
__device__ __forceinline__ void bx_store(float* p, const float* o, int n) { for (int k = 0; k < n; ++k) p[k] = o[k]; }
__device__ __forceinline__ void bx_store(bf16* p, const float* o, int) {
    uint4 t; t.x = pack_bf16x2(o[0], o[1]); t.y = pack_bf16x2(o[2], o[3]); t.z = pack_bf16x2(o[4], o[5]); t.w = pack_bf16x2(o[6], o[7]);
    *reinterpret_cast<uint4*>(p) = t;
}
// the lane's two adjacent dy pixels: fp32 narrow tensors -- 2 NCH contiguous floats, 8-byte aligned (Wo = 2 W) -> NCH 8-byte loads; bf16 -- two 16-byte loads C apart
template <int NCH>
__device__ __forceinline__ void bx_load_pair(const float* row, int, BxRaw<float, NCH>& a, BxRaw<float, NCH>& b) {
    float tmp[2 * NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) { const float2 q = reinterpret_cast<const float2*>(row)[k]; tmp[2 * k] = q.x; tmp[2 * k + 1] = q.y; }
#pragma unroll
    for (int k = 0; k < NCH; ++k) { a.v[k] = tmp[k]; b.v[k] = tmp[NCH + k]; }
}
__device__ __forceinline__ void bx_load_pair(const bf16* row, int C, BxRaw<bf16, 8>& a, BxRaw<bf16, 8>& b) { a.load(row); b.load(row + C); }
// HALO (one channel group per column, CV = 1: the narrow fp32 tensors): a wave covers 62 columns + one halo lane at either end, which loads and exchanges but stores nothing
// (the adjacent wave owns its column) -- no edge fetches, 50 fewer registers; otherwise (bf16, CV channel vectors per column) the lanes at the ends of the wave's pixel run
// fetch the one column no neighbour lane holds.
template <typename T, int NCH, bool HALO>
__global__ void __launch_bounds__(PB) k_bilinear_bwd_x2(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int wcols, int hstrips) {
    const int CV = HALO ? 1 : C / NCH, PPW = 64 / CV;           // lanes of a wave: PPW consecutive columns x CV channel vectors
    const int lane = threadIdx.x & 63;
    const int wv = blockIdx.x * (PB / 64) + (int)(threadIdx.x >> 6);        // wave-uniform; no block-level synchronisation below
    if (wv >= N * hstrips * wcols) return;
    const int wc = wv % wcols, hs = (wv / wcols) % hstrips, n = wv / (wcols * hstrips);
    const int pl = lane / CV, cv = lane - pl * CV;
    const int wi = HALO ? wc * 62 - 1 + pl : wc * PPW + pl, c0 = cv * NCH;
    const bool live = wi >= 0 && wi < W;
    const bool writes = live && (!HALO || (pl >= 1 && pl <= 62));
    const int hA = hs * BX_SH, hB = min(H, hA + BX_SH);
    const int Ho = 2 * H, Wo = 2 * W;
    const T* g = dy + (int64_t)n * Ho * Wo * C + c0;
    const bool edge_l = !HALO && live && pl == 0 && wi > 0, edge_r = !HALO && live && pl == PPW - 1 && wi < W - 1;
    BxRaw<T, NCH> A[BX_ROWS], Bq[BX_ROWS], E[HALO ? 1 : BX_ROWS];
#pragma unroll
    for (int i = 0; i < BX_ROWS; ++i) {
        const int r = 2 * hA - 1 + i;
        const bool rok = r >= 0 && r < Ho && r <= 2 * hB;       // (wave-uniform)
        A[i].zero(); Bq[i].zero();
        if (!HALO) E[i].zero();
        if (rok && live) {
            const T* row = g + ((int64_t)r * Wo + 2 * wi) * C;
            bx_load_pair(row, C, A[i], Bq[i]);
            if (!HALO) {
                if (edge_l) E[i].load(row - C);
                if (edge_r) E[i].load(row + 2 * C);
            }
        }
    }
    const float cw0 = wi > 0 ? 0.25f : 0.f, cw1 = wi > 0 ? 0.75f : 1.f, cw2 = wi < W - 1 ? 0.75f : 1.f, cw3 = wi < W - 1 ? 0.25f : 0.f;
    float t[BX_ROWS][NCH];
#pragma unroll
    for (int i = 0; i < BX_ROWS; ++i) {
        BxRaw<T, NCH> L = Bq[i].shfl(CV, true), Rr = A[i].shfl(CV, false);      // column 2 wi - 1 = the left lane's second column, 2 wi + 2 = the right lane's first
        if (!HALO) {
            if (edge_l) L = E[i];
            if (edge_r) Rr = E[i];
        }
#pragma unroll
        for (int k = 0; k < NCH; ++k) t[i][k] = ((cw0 * L.get(k) + cw1 * A[i].get(k)) + cw2 * Bq[i].get(k)) + cw3 * Rr.get(k);
    }
    if (!writes) return;
#pragma unroll
    for (int j = 0; j < BX_SH; ++j) {
        const int hi = hA + j;
        if (hi >= hB) break;
        const float rw0 = hi > 0 ? 0.25f : 0.f, rw1 = hi > 0 ? 0.75f : 1.f, rw2 = hi < H - 1 ? 0.75f : 1.f, rw3 = hi < H - 1 ? 0.25f : 0.f;
        float o[NCH];
#pragma unroll
        for (int k = 0; k < NCH; ++k) o[k] = ((rw0 * t[2 * j][k] + rw1 * t[2 * j + 1][k]) + rw2 * t[2 * j + 2][k]) + rw3 * t[2 * j + 3][k];
        bx_store(dx + (((int64_t)n * H + hi) * W + wi) * C + c0, o, NCH);
    }
}

// Separable backward for narrow fp32 tensors (the 5-class aux logits resized x2 / x4 / x8 to the input size): the transposed
// interpolation factors into a pass along W (dy [N,Ho,Wo,C] -> tmp [N,Ho,W,C], every dy element read once, coalesced) and a pass
// along H (tmp -> dx [N,H,W,C]).  The 2-D gather reads (2/scale)^2 = 16 .. 256 dy elements per dx element one after another from a
// few hundred threads: 0.27 ms for the x8 head against ~0.05 ms here.  tmp is caller-provided workspace (fp32, N*Ho*W*C elements).
template <bool ALONG_W>
__global__ void k_bilinear_bwd_1d(const float* __restrict__ src, float* __restrict__ dst, int rows_out, int In, int Out, int Wn, int C,
                                  float scale, int align) {
    // ALONG_W: src [rows][Out][C] -> dst [rows][In][C]   (rows = N*Ho, In = W, Out = Wo; Wn unused)
    // else   : src [N][Out][Wn*C] -> dst [N][In][Wn*C]    (rows_out = N*In, Out = Ho, In = H, Wn*C contiguous elements per row)
    if (ALONG_W) {
        // one block per dy row (n, ho): the row (Out*C floats) is staged in LDS with coalesced loads, every thread then forms its
        // (wi, c) outputs from LDS -- gathering straight from global touched 20-byte pieces 160 bytes apart
        extern __shared__ float srow[];
        for (int row = blockIdx.x; row < rows_out; row += gridDim.x) {
            __syncthreads();
            const float* s = src + (int64_t)row * Out * C;
            for (int i = threadIdx.x; i < Out * C; i += blockDim.x) srow[i] = s[i];
            __syncthreads();
            for (int j = threadIdx.x; j < In * C; j += blockDim.x) {
                const int wi = j / C, c = j - wi * C;
                int lo, hi;
                cand_range(wi, scale, Out, align, lo, hi);
                float acc = 0.f;
                for (int o = lo; o <= hi; ++o) {
                    const Lerp a = src_index(o, scale, In, align);
                    const float w = (a.i0 == wi ? a.l0 : 0.f) + (a.i1 == wi ? a.l1 : 0.f);
                    acc += w * srow[o * C + c];
                }
                dst[((int64_t)row * In + wi) * C + c] = acc;
            }
        }
    } else {
        const int j = blockIdx.x * blockDim.x + threadIdx.x;          // element inside a row of Wn*C
        const int RW = Wn * C;
        if (j >= RW) return;
        for (int row = blockIdx.y; row < rows_out; row += gridDim.y) {
            const int n = row / In, hi_ = row - n * In;
            int lo, hi;
            cand_range(hi_, scale, Out, align, lo, hi);
            const float* s = src + (int64_t)n * Out * RW + j;
            float acc = 0.f;
            for (int o = lo; o <= hi; ++o) {                            // block-uniform candidate rows
                const Lerp a = src_index(o, scale, In, align);
                const float w = (a.i0 == hi_ ? a.l0 : 0.f) + (a.i1 == hi_ ? a.l1 : 0.f);
                if (w != 0.f) acc += w * s[(int64_t)o * RW];
            }
            dst[(int64_t)row * RW + j] = acc;
        }
    }
}
/* fp32 only; workspace: N*Ho*W*C floats */
extern "C" int tcct_bilinear_bwd_separable(const float* dy, float* dx, float* workspace, int N, int H, int W, int C, int Ho, int Wo,
                                           int align_corners, tcct_stream_t stream) {
    TCCT_CHECK(H > 0 && W > 0 && Ho > 0 && Wo > 0 && N > 0 && C > 0 && workspace, "bilinear_bwd_separable: bad arguments");
    float sh = align_corners ? (Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f) : (float)H / (float)Ho;
    float sw = align_corners ? (Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f) : (float)W / (float)Wo;
    TCCT_CHECK(sw > 0.f && (int)(2.f / sw) + 3 <= BL_MAXC, "bilinear_bwd_separable: horizontal factor out of range");
    hipStream_t st = (hipStream_t)stream;
    const size_t row_bytes = sizeof(float) * (size_t)Wo * C;
    TCCT_CHECK(row_bytes <= 64 * 1024, "bilinear_bwd_separable: a dy row of %zu bytes does not fit the LDS stage", row_bytes);
    const int64_t rows = (int64_t)N * Ho;
    hipLaunchKernelGGL(k_bilinear_bwd_1d<true>, dim3((unsigned)(rows < 8192 ? rows : 8192)), dim3(PB), row_bytes, st, dy, workspace, N * Ho, W, Wo, 0, C, sw, align_corners);
    hipLaunchKernelGGL(k_bilinear_bwd_1d<false>, row_grid(W * C, (int64_t)N * H), dim3(PB), 0, st, workspace, dx, N * H, H, Ho, W, C, sh, align_corners);
    TCCT_LAUNCH_OK();
}

// (Round 4 tried two other forms for the fp32 n_class-channel logits of the level-0 head, 0.129 ms in the element-per-lane form above: a pixel per thread
// (20-byte lane stride) 0.159 ms, flat runs of four floats with 16-byte accesses for the addend and the result 0.139 ms -- the four scalar source gathers per
// element, not the streams, are the cost; a separable two-pass form would halve them.)
static int bilinear_fwd_impl(const void* x, const void* res, void* y, int N, int H, int W, int C, int Ho, int Wo, int align_corners,
                             int dtype, tcct_stream_t stream);
extern "C" int tcct_bilinear_fwd(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int align_corners,
                                 int dtype, tcct_stream_t stream) {
    return bilinear_fwd_impl(x, nullptr, y, N, H, W, C, Ho, Wo, align_corners, dtype, stream);
}
/* y = resize(x) + res with res, y [N,Ho,Wo,C]: upsampling with the skip-connection add folded in */
extern "C" int tcct_bilinear_add_fwd(const void* x, const void* res, void* y, int N, int H, int W, int C, int Ho, int Wo,
                                     int align_corners, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr, "bilinear_add_fwd: res is NULL");
    return bilinear_fwd_impl(x, res, y, N, H, W, C, Ho, Wo, align_corners, dtype, stream);
}
static int bilinear_fwd_impl(const void* x, const void* res, void* y, int N, int H, int W, int C, int Ho, int Wo, int align_corners,
                             int dtype, tcct_stream_t stream) {
    TCCT_CHECK(H > 0 && W > 0 && Ho > 0 && Wo > 0, "bilinear_fwd: bad sizes");
    float sh = align_corners ? (Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f) : (float)H / (float)Ho;
    float sw = align_corners ? (Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f) : (float)W / (float)Wo;
    int vec = (C % 4 == 0) ? 4 : 1;
    hipStream_t st = (hipStream_t)stream;
    static int v8 = -1;         // compile-time A/B switch (0: 8-byte accesses for bf16 as well)
    if (v8 < 0) v8 = 1;
    if (v8 && dtype == TCCT_BF16 && C % 8 == 0) {
        hipLaunchKernelGGL((k_bilinear_fwd<bf16, 8>), row_grid(Wo * (C / 8), (int64_t)N * Ho), dim3(PB), 0, st, (const bf16*)x, (bf16*)y, N, H, W, C, Ho, Wo, sh, sw, align_corners, (const bf16*)res);
        TCCT_LAUNCH_OK();
    }
    if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_fwd<T, 4>), row_grid(Wo * (C / 4), (int64_t)N * Ho), dim3(PB), 0, st, (const T*)x, (T*)y, N, H, W, C, Ho, Wo, sh, sw, align_corners, (const T*)res)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_fwd<T, 1>), row_grid(Wo * C, (int64_t)N * Ho), dim3(PB), 0, st, (const T*)x, (T*)y, N, H, W, C, Ho, Wo, sh, sw, align_corners, (const T*)res)); }
    TCCT_LAUNCH_OK();
}
static int g_bilinear_x2 = 1;
/* kernel A/B (tools/bilinear_bwd_bench.py, tests): 0 = exact x2 resizes take the tiled gather kernel like every other scale; returns the previous value */
extern "C" int64_t tcct_bilinear_bwd_x2(int on) { const int old = g_bilinear_x2; g_bilinear_x2 = on ? 1 : 0; return old; }
/* dy [N,Ho,Wo,C] -> dx [N,H,W,C] (H,W = forward input size) */
extern "C" int tcct_bilinear_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int align_corners,
                                 int dtype, tcct_stream_t stream) {
    TCCT_CHECK(H > 0 && W > 0 && Ho > 0 && Wo > 0, "bilinear_bwd: bad sizes");
    float sh = align_corners ? (Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f) : (float)H / (float)Ho;
    float sw = align_corners ? (Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f) : (float)W / (float)Wo;
    int vec = (C % 4 == 0) ? 4 : 1;
    hipStream_t st = (hipStream_t)stream;
    // entries per table row: outputs within +-1 source pixel of an input index = 2/scale (+ slack); beyond BL_MAXC use the search kernel
    if (!align_corners && Ho == 2 * H && Wo == 2 * W && g_bilinear_x2) {        // exact x2: the separable lane-exchange kernel
        const int nch = (dtype == TCCT_BF16 && C % 8 == 0 && (C == 8 || C == 16 || C == 32 || C == 64)) ? 8 : ((dtype == TCCT_F32 && (C == 5 || C == 9)) ? C : 0);
        if (nch) {
            const int CV = C / nch, PPW = 64 / CV;
            const int wcols = nch == 8 ? (W + PPW - 1) / PPW : (W + 61) / 62, hstrips = (H + BX_SH - 1) / BX_SH;
            const int64_t waves = (int64_t)N * wcols * hstrips;
            TCCT_CHECK(waves < 0x7fffffffLL, "bilinear_bwd: grid too large");
            const dim3 grid((unsigned)((waves + PB / 64 - 1) / (PB / 64)));
            if (nch == 8) hipLaunchKernelGGL((k_bilinear_bwd_x2<bf16, 8, false>), grid, dim3(PB), 0, st, (const bf16*)dy, (bf16*)dx, N, H, W, C, wcols, hstrips);
            else if (nch == 5) hipLaunchKernelGGL((k_bilinear_bwd_x2<float, 5, true>), grid, dim3(PB), 0, st, (const float*)dy, (float*)dx, N, H, W, C, wcols, hstrips);
            else hipLaunchKernelGGL((k_bilinear_bwd_x2<float, 9, true>), grid, dim3(PB), 0, st, (const float*)dy, (float*)dx, N, H, W, C, wcols, hstrips);
            TCCT_LAUNCH_OK();
        }
    }
    const float smin = fminf(sh, sw);
    const int KT = smin > 0.f ? (int)(2.f / smin) + 3 : BL_MAXC + 1;
    if (KT <= BL_MAXC) {
        const int DW = 32;
        const int tilesW = (W + DW - 1) / DW, tilesH = (H + BT_DH - 1) / BT_DH;
        const int64_t blocks = (int64_t)N * tilesW * tilesH;
        TCCT_CHECK(blocks < 0x7fffffffLL, "bilinear_bwd: grid too large");
        const size_t lds = sizeof(int) * ((size_t)2 * (BT_DH + DW) * KT + BT_DH + DW);
        static int v8 = -1;         // compile-time A/B switch (0: 8-byte accesses for bf16 as well)
        if (v8 < 0) v8 = 1;
        if (v8 && dtype == TCCT_BF16 && C % 8 == 0) {
            hipLaunchKernelGGL((k_bilinear_bwd_tab<bf16, 8>), dim3((unsigned)blocks), dim3(PB), lds, st, (const bf16*)dy, (bf16*)dx, N, H, W, C, Ho, Wo, sh, sw, align_corners, DW, KT, tilesW, tilesH);
            TCCT_LAUNCH_OK();
        }
        if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_bwd_tab<T, 4>), dim3((unsigned)blocks), dim3(PB), lds, st, (const T*)dy, (T*)dx, N, H, W, C, Ho, Wo, sh, sw, align_corners, DW, KT, tilesW, tilesH)); }
        else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_bwd_tab<T, 1>), dim3((unsigned)blocks), dim3(PB), lds, st, (const T*)dy, (T*)dx, N, H, W, C, Ho, Wo, sh, sw, align_corners, DW, KT, tilesW, tilesH)); }
        TCCT_LAUNCH_OK();
    }
    int64_t total = (int64_t)N * H * W * (C / vec);
    if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_bwd<T, 4>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, st, (const T*)dy, (T*)dx, N, H, W, C, Ho, Wo, sh, sw, align_corners)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bilinear_bwd<T, 1>), dim3(tcct_grid(total, PB, 1 << 16)), dim3(PB), 0, st, (const T*)dy, (T*)dx, N, H, W, C, Ho, Wo, sh, sw, align_corners)); }
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------ GateFusion (training)
// reference nets/tcct.py:916-932: alpha = clamp(F.interpolate(rand(B,C,hs,ws), size=(H,W), mode='bicubic'), 0, 1) (align_corners=False,
// cubic convolution with A = -0.75, border indices clamped -- torch's upsample_bicubic2d); out = x1*alpha + x2*(1-alpha).  The small
// random field `a` ([B,hs,ws,C] fp32, NHWC, drawn by the caller) is expanded on the fly: alpha is never materialised.
// MODE 0: y = x1*alpha + x2*(1-alpha);  MODE 1 (backward): y = dy*alpha, y2 = dy*(1-alpha).
__device__ __forceinline__ void cubic_w(float t, float* w) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x1 = t, x2 = 1.f - t, x3 = 2.f - t;
    w[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    w[1] = ((A + 2.f) * x1 - (A + 3.f)) * x1 * x1 + 1.f;
    w[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    w[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}
template <typename T, int MODE>
__global__ void k_gate_fusion(const T* __restrict__ x1, const T* __restrict__ x2, const float* __restrict__ a, T* __restrict__ y,
                              T* __restrict__ y2, int N, int H, int W, int C, int hs, int ws, float sh, float sw) {
    const int C4 = C >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W * C4) return;
    const int wo = i / C4, c = (i - wo * C4) * 4;
    const float fx = sw * ((float)wo + 0.5f) - 0.5f;
    const int ix = (int)floorf(fx);
    float wx[4];
    cubic_w(fx - (float)ix, wx);
    int cx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cx[k] = min(max(ix - 1 + k, 0), ws - 1) * C + c;
    for (int row = blockIdx.y; row < N * H; row += gridDim.y) {
        const int n = row / H, ho = row - n * H;
        const float fy = sh * ((float)ho + 0.5f) - 0.5f;
        const int iy = (int)floorf(fy);
        float wy[4];
        cubic_w(fy - (float)iy, wy);
        float al[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* ar = a + ((int64_t)n * hs + min(max(iy - 1 + r, 0), hs - 1)) * ws * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 v = *reinterpret_cast<const float4*>(ar + cx[k]);
                const float wgt = wy[r] * wx[k];
                al[0] += wgt * v.x; al[1] += wgt * v.y; al[2] += wgt * v.z; al[3] += wgt * v.w;
            }
        }
        const int64_t off = ((int64_t)row * W + wo) * C + c;
        const f4 u = ld4(x1 + off);
        f4 o, o2;
        if (MODE == 0) {
            const f4 v = ld4(x2 + off);
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = fminf(fmaxf(al[k], 0.f), 1.f); o.v[k] = u.v[k] * t + v.v[k] * (1.f - t); }
            st4(y + off, o);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = fminf(fmaxf(al[k], 0.f), 1.f); o.v[k] = u.v[k] * t; o2.v[k] = u.v[k] * (1.f - t); }
            st4(y + off, o); st4(y2 + off, o2);
        }
    }
}
static int gate_launch(const void* x1, const void* x2, const float* a, void* y, void* y2, int N, int H, int W, int C, int hs, int ws,
                       int dtype, int mode, tcct_stream_t stream, const char* who) {
    if (!(C % 4 == 0 && C >= 4 && N >= 1 && H >= 1 && W >= 1 && hs >= 1 && ws >= 1)) { tcct_set_error("%s: bad shape", who); return -1; }
    const float sh = (float)hs / (float)H, sw = (float)ws / (float)W;        // align_corners=False: scale = in / out
    dim3 g = row_grid(W * (C / 4), (int64_t)N * H);
    if (mode == 0) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_gate_fusion<T, 0>), g, dim3(PB), 0, (hipStream_t)stream, (const T*)x1, (const T*)x2, a, (T*)y, (T*)y2, N, H, W, C, hs, ws, sh, sw)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_gate_fusion<T, 1>), g, dim3(PB), 0, (hipStream_t)stream, (const T*)x1, (const T*)x2, a, (T*)y, (T*)y2, N, H, W, C, hs, ws, sh, sw)); }
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_gate_fusion_fwd(const void* x1, const void* x2, const float* field, void* y, int N, int H, int W, int C, int hs, int ws,
                                    int dtype, tcct_stream_t stream) {
    return gate_launch(x1, x2, field, y, nullptr, N, H, W, C, hs, ws, dtype, 0, stream, "gate_fusion_fwd");
}
extern "C" int tcct_gate_fusion_bwd(const void* dy, const float* field, void* dx1, void* dx2, int N, int H, int W, int C, int hs, int ws,
                                    int dtype, tcct_stream_t stream) {
    return gate_launch(dy, nullptr, field, dx1, dx2, N, H, W, C, hs, ws, dtype, 1, stream, "gate_fusion_bwd");
}

// ------------------------------------------------------------------------------------------ L2 normalise over C
// LP = C/4 lanes per pixel (power of two <= 64)
template <typename T, bool BWD>
__global__ void k_l2norm(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, int64_t M, int C, float eps, float oscale,
                         const T* __restrict__ res = nullptr) {
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
            for (int k = 0; k < 4; ++k) r.v[k] = oscale * (g.v[k] / d - v.v[k] * coef);
            if (res) {          // + the gradient that reaches x through its other consumer (norm_add's inputs also feed the aux heads)
                const f4 e = ld4(res + ii * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) r.v[k] += e.v[k];
            }
        }
        if (ok) st4(out + ii * 4, r);
    }
}
extern "C" int tcct_l2norm_fwd(const void* x, void* y, int64_t M, int C, float eps, int dtype, tcct_stream_t stream) {
    int LP = C / 4;
    TCCT_CHECK(C % 4 == 0 && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0, "l2norm_fwd: C=%d unsupported", C);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_l2norm<T, false>), dim3(tcct_grid(M * LP, PB, 1 << 16)), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)nullptr, (T*)y, M, C, eps, 1.f));
    TCCT_LAUNCH_OK();
}
static int l2norm_bwd_impl(const void* x, const void* dy, void* dx, int64_t M, int C, float eps, float oscale, int dtype, tcct_stream_t stream,
                           const void* res = nullptr) {
    int LP = C / 4;
    TCCT_CHECK(C % 4 == 0 && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0, "l2norm_bwd: C=%d unsupported", C);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_l2norm<T, true>), dim3(tcct_grid(M * LP, PB, 1 << 16)), dim3(PB), 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, M, C, eps, oscale, (const T*)res));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_l2norm_bwd(const void* x, const void* dy, void* dx, int64_t M, int C, float eps, int dtype, tcct_stream_t stream) {
    return l2norm_bwd_impl(x, dy, dx, M, C, eps, 1.f, dtype, stream);
}
/* dx = scale * l2norm_bwd(x, dy): the 1/3 of norm_add's mean (nets/tcct.py:937-942) rides on the pass instead of a separate scaling */
extern "C" int tcct_l2norm_bwd_scaled(const void* x, const void* dy, void* dx, int64_t M, int C, float eps, float scale, int dtype,
                                      tcct_stream_t stream) {
    return l2norm_bwd_impl(x, dy, dx, M, C, eps, scale, dtype, stream);
}

/* dx = scale * l2norm_bwd(x, dy) + res (res: same shape and dtype as x, may not be NULL): norm_add's inputs g_i also feed the aux heads
 * (nets/tcct.py:1035-1040); the heads' gradient of g_i is added here instead of by an autograd accumulation pass */
extern "C" int tcct_l2norm_bwd_scaled_add(const void* x, const void* dy, const void* res, void* dx, int64_t M, int C, float eps, float scale,
                                          int dtype, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr && res != dx, "l2norm_bwd_scaled_add: res must be a separate tensor");
    return l2norm_bwd_impl(x, dy, dx, M, C, eps, scale, dtype, stream, res);
}

// ------------------------------------------------- backward of norm_add when its gradient comes from the feature-polarization loss (round 4)
// d loss / d feats of RegNet.regular_udh (reference nets/reg.py:86-105, nets/fcs.py:25-50) is a pure function of TWO BYTES per pixel:
//     dfeat[p][c] = g * dpro_over_n[label[p]][bin[p]][c]      (0 for pixels outside every bin; tcct_fpl_backward)
// with a [classes][32][32] fp32 table.  Round 3 wrote that 452 MB tensor (bench shape) and read it three times -- the L2-normalise backward of g0
// and the two bilinear backward passes to the coarser maps.  Here the three kernels look the rows up in LDS instead: dfeat is never written
// (k_fpl_bwd is gone from the step) and the two resize gradients read 2 bytes per contributing pixel instead of 64.  Values are rounded to the
// storage type first, exactly as the tensor used to hold them.
#define FG_BINS 32
template <typename T> __device__ __forceinline__ float fg_round(float v);
template <> __device__ __forceinline__ float fg_round<float>(float v) { return v; }
template <> __device__ __forceinline__ float fg_round<bf16>(float v) { return __bfloat162float(__float2bfloat16(v)); }

// dx = oscale * l2norm_bwd(x, dfeat) + res  (C = 32: eight lanes per pixel)
template <typename T>
__global__ void k_l2norm_bwd_fplgrad(const T* __restrict__ x, const uint8_t* __restrict__ lab, const uint8_t* __restrict__ binmap,
                                     const float* __restrict__ dpro, const float* __restrict__ gout, float gscale, int ncls, const T* __restrict__ res,
                                     T* __restrict__ out, int64_t M, float eps, float oscale) {
    extern __shared__ float stab[];         // [ncls][32][32]
    for (int i = threadIdx.x; i < ncls * FG_BINS * 32; i += blockDim.x) stab[i] = dpro[i];
    __syncthreads();
    const float gs = gscale * (gout ? *gout : 1.f);
    const int64_t total = M * 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (total + stride - 1) / stride;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t it = 0; it < rounds; ++it, i += stride) {
        const bool ok = i < total;
        const int64_t ii = ok ? i : 0, p = ii >> 3;
        const int sub = (int)(ii & 7);
        const f4 v = ld4(x + ii * 4);
        const int b = binmap[p], l = lab[p];
        f4 g = f4zero();
        if (b < FG_BINS && l < ncls) {
            const float* d = stab + (l * FG_BINS + b) * 32 + sub * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) g.v[k] = fg_round<T>(gs * d[k]);
        }
        float ss = v.v[0] * v.v[0] + v.v[1] * v.v[1] + v.v[2] * v.v[2] + v.v[3] * v.v[3];
        float dot = v.v[0] * g.v[0] + v.v[1] * g.v[1] + v.v[2] * g.v[2] + v.v[3] * g.v[3];
        for (int o = 4; o > 0; o >>= 1) { ss += __shfl_xor(ss, o, 64); dot += __shfl_xor(dot, o, 64); }
        const float nrm = sqrtf(ss), dn = fmaxf(nrm, eps);
        const float coef = nrm > eps ? dot / (dn * dn * nrm) : 0.f;
        f4 r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r.v[k] = oscale * (g.v[k] / dn - v.v[k] * coef);
        if (res) {
            const f4 e = ld4(res + ii * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) r.v[k] += e.v[k];
        }
        if (ok) st4(out + ii * 4, r);
    }
}
/* dx = scale * l2norm_bwd(x, dfeat) (+ res, nullable) with dfeat looked up from (labels, binmap, dpro_over_n [ncls][32][32]) instead of read: the
 * level-0 branch of norm_add's backward under the feature-polarization loss.  x, res, dx [M,32]; grad_out: device scalar (nullable) */
extern "C" int tcct_l2norm_bwd_fplgrad(const void* x, const uint8_t* labels, const uint8_t* binmap, const float* dpro_over_n, const float* grad_out,
                                       float grad_scale, int ncls, const void* res, void* dx, int64_t M, float eps, float scale, int dtype,
                                       tcct_stream_t stream) {
    TCCT_CHECK(ncls >= 1 && ncls <= 16 && labels && binmap && dpro_over_n, "l2norm_bwd_fplgrad: bad arguments (ncls=%d)", ncls);
    const size_t lds = sizeof(float) * (size_t)ncls * FG_BINS * 32;
    TCCT_DISPATCH(dtype, {
        static bool at_ = false;
        if (!at_) { (void)hipFuncSetAttribute((const void*)k_l2norm_bwd_fplgrad<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); at_ = true; }
        hipLaunchKernelGGL((k_l2norm_bwd_fplgrad<T>), dim3(tcct_grid(M * 8, PB, 2048)), dim3(PB), lds, (hipStream_t)stream, (const T*)x, labels, binmap,
                           dpro_over_n, grad_out, grad_scale, ncls, (const T*)res, (T*)dx, M, eps, scale); });
    TCCT_LAUNCH_OK();
}

// dx [N,H,W,32] = bilinear_bwd(dfeat [N,Ho,Wo,32]) with dfeat looked up: k_bilinear_bwd_tab with the contributor loads replaced by LDS reads.  Per tile
// the (label, bin) bytes of the contributing output region (a contiguous rectangle: the contributors of neighbouring inputs overlap) are read ONCE,
// coalesced, and kept as 16-bit row ids; the table is stored pre-scaled and pre-rounded (fg_round(g * dpro)), so the inner loop is two LDS reads and
// eight FMAs per contributor.  (First version: two global byte loads per contributor and item, four items per pixel: 0.38 ms for the two calls of
// the bench step, as slow as the dense gather it replaced.)
template <typename T>
__global__ void __launch_bounds__(PB) k_bilinear_bwd_fplgrad(const uint8_t* __restrict__ lab, const uint8_t* __restrict__ binmap, const float* __restrict__ dpro,
                                                             const float* __restrict__ gout, float gscale, int ncls, T* __restrict__ dx, int N, int H,
                                                             int W, int Ho, int Wo, float sh, float sw, int align, int DW, int KT, int tilesW,
                                                             int tilesH, int RMAX, int CMAX) {
    extern __shared__ int smem_i[];             // idx[(DH+DW)][KT], weights[(DH+DW)][KT], counts[DH+DW], the table [ncls][32][32], ids [RMAX][CMAX] u16
    int* tidx = smem_i;
    float* twt = reinterpret_cast<float*>(smem_i + (BT_DH + DW) * KT);
    int* tcnt = smem_i + 2 * (BT_DH + DW) * KT;
    float* stab = reinterpret_cast<float*>(smem_i + ((2 * (BT_DH + DW) * KT + BT_DH + DW + 3) & ~3));
    uint16_t* sid = reinterpret_cast<uint16_t*>(stab + ncls * FG_BINS * 32);
    __shared__ int s_org[4];                    // first row / column of the contributing region and its extent
    const int t = threadIdx.x;
    const float gs = gscale * (gout ? *gout : 1.f);
    for (int i = t; i < ncls * FG_BINS * 32; i += PB) stab[i] = fg_round<T>(gs * dpro[i]);
    int bid = blockIdx.x;
    const int tw = bid % tilesW; bid /= tilesW;
    const int th = bid % tilesH;
    const int n = bid / tilesH;
    const int hi0 = th * BT_DH, wi0 = tw * DW;
    if (t < BT_DH + DW) {
        const bool isrow = t < BT_DH;
        const int i = isrow ? hi0 + t : wi0 + (t - BT_DH);
        const int in = isrow ? H : W, out = isrow ? Ho : Wo;
        const float sc = isrow ? sh : sw;
        int cnt = 0;
        if (i < in) {
            int lo, hi;
            cand_range(i, sc, out, align, lo, hi);
            for (int o = lo; o <= hi && cnt < KT; ++o) {
                Lerp a = src_index(o, sc, in, align);
                float wgt = (a.i0 == i ? a.l0 : 0.f) + (a.i1 == i ? a.l1 : 0.f);
                if (wgt != 0.f) { tidx[t * KT + cnt] = o; twt[t * KT + cnt] = wgt; ++cnt; }
            }
        }
        tcnt[t] = cnt;
    }
    __syncthreads();
    if (t == 0) {       // contributors are ascending per input and the inputs are ascending: first of the first / last of the last non-empty entry
        int r0 = 0x7fffffff, r1 = -1, c0 = 0x7fffffff, c1 = -1;
        for (int q = 0; q < BT_DH; ++q) if (tcnt[q]) { r0 = min(r0, tidx[q * KT]); r1 = max(r1, tidx[q * KT + tcnt[q] - 1]); }
        for (int q = 0; q < DW; ++q) if (tcnt[BT_DH + q]) { c0 = min(c0, tidx[(BT_DH + q) * KT]); c1 = max(c1, tidx[(BT_DH + q) * KT + tcnt[BT_DH + q] - 1]); }
        s_org[0] = r0; s_org[1] = c0; s_org[2] = r1 >= r0 ? min(r1 - r0 + 1, RMAX) : 0; s_org[3] = c1 >= c0 ? min(c1 - c0 + 1, CMAX) : 0;
    }
    __syncthreads();
    const int r0 = s_org[0], c0 = s_org[1], RR = s_org[2], CC = s_org[3];
    const uint8_t* L = lab + (int64_t)n * Ho * Wo;
    const uint8_t* B = binmap + (int64_t)n * Ho * Wo;
    for (int i = t; i < RR * CC; i += PB) {
        const int rr = i / CC, cc = i - rr * CC;
        const int64_t q = (int64_t)(r0 + rr) * Wo + (c0 + cc);
        const int bb = B[q], ll = L[q];
        sid[rr * CMAX + cc] = (bb < FG_BINS && ll < ncls) ? (uint16_t)(ll * FG_BINS + bb) : (uint16_t)0xffff;
    }
    __syncthreads();
    T* out = dx + (int64_t)n * H * W * 32;
    for (int i = t; i < BT_DH * DW * 4; i += PB) {
        const int cv = i & 3, pix = i >> 2;
        const int r = pix / DW, cc = pix - r * DW;
        const int hi = hi0 + r, wi = wi0 + cc;
        if (hi >= H || wi >= W) continue;
        const int nr = tcnt[r], nc = tcnt[BT_DH + cc];
        const int* ri = tidx + r * KT;
        const float* rw = twt + r * KT;
        const int* ci = tidx + (BT_DH + cc) * KT;
        const float* cw = twt + (BT_DH + cc) * KT;
        float acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        for (int a = 0; a < nr; ++a) {
            const uint16_t* srow = sid + (ri[a] - r0) * CMAX - c0;
            const float wh = rw[a];
            for (int b = 0; b < nc; ++b) {
                const int id = srow[ci[b]];
                if (id != 0xffff) {
                    const float gw = wh * cw[b];
                    const float4 d0 = *reinterpret_cast<const float4*>(stab + id * 32 + cv * 8);
                    const float4 d1 = *reinterpret_cast<const float4*>(stab + id * 32 + cv * 8 + 4);
                    acc[0] += gw * d0.x; acc[1] += gw * d0.y; acc[2] += gw * d0.z; acc[3] += gw * d0.w;
                    acc[4] += gw * d1.x; acc[5] += gw * d1.y; acc[6] += gw * d1.z; acc[7] += gw * d1.w;
                }
            }
        }
        T* o = out + ((int64_t)hi * W + wi) * 32 + cv * 8;
        stv<4>(o, acc);
        stv<4>(o + 4, acc + 4);
    }
}
/* dx [N,H,W,32] = bilinear_bwd(dfeat), dfeat [N,Ho,Wo,32] looked up as above (align_corners as the forward resize of norm_add: 0) */
extern "C" int tcct_bilinear_bwd_fplgrad(const uint8_t* labels, const uint8_t* binmap, const float* dpro_over_n, const float* grad_out, float grad_scale,
                                         int ncls, void* dx, int N, int H, int W, int Ho, int Wo, int align_corners, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(H > 0 && W > 0 && Ho > 0 && Wo > 0 && ncls >= 1 && ncls <= 16, "bilinear_bwd_fplgrad: bad sizes");
    const float sh = align_corners ? (Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f) : (float)H / (float)Ho;
    const float sw = align_corners ? (Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f) : (float)W / (float)Wo;
    const float smin = fminf(sh, sw);
    const int KT = smin > 0.f ? (int)(2.f / smin) + 3 : BL_MAXC + 1;
    TCCT_CHECK(KT <= BL_MAXC, "bilinear_bwd_fplgrad: scale factor out of range");
    const int DW = 32;
    const int tilesW = (W + DW - 1) / DW, tilesH = (H + BT_DH - 1) / BT_DH;
    const int64_t blocks = (int64_t)N * tilesW * tilesH;
    TCCT_CHECK(blocks < 0x7fffffffLL, "bilinear_bwd_fplgrad: grid too large");
    // contributing region of a tile: its inputs span (BT_DH - 1) / sh resp. (DW - 1) / sw outputs plus one source pixel of reach on either side
    const int RMAX = (int)((BT_DH - 1) / sh + 2.f / sh) + 4, CMAX = ((int)((DW - 1) / sw + 2.f / sw) + 4 + 1) & ~1;
    const size_t lds = sizeof(int) * (((size_t)2 * (BT_DH + DW) * KT + BT_DH + DW + 3) & ~(size_t)3) + sizeof(float) * (size_t)ncls * FG_BINS * 32 +
                       sizeof(uint16_t) * (size_t)RMAX * CMAX;
    TCCT_CHECK(lds <= 150 * 1024, "bilinear_bwd_fplgrad: %zu B of LDS", lds);
    TCCT_DISPATCH(dtype, {
        static bool at_ = false;
        if (!at_) { (void)hipFuncSetAttribute((const void*)k_bilinear_bwd_fplgrad<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); at_ = true; }
        hipLaunchKernelGGL((k_bilinear_bwd_fplgrad<T>), dim3((unsigned)blocks), dim3(PB), lds, (hipStream_t)stream, labels, binmap, dpro_over_n, grad_out,
                           grad_scale, ncls, (T*)dx, N, H, W, Ho, Wo, sh, sw, align_corners, DW, KT, tilesW, tilesH, RMAX, CMAX); });
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------- norm_add (nets/tcct.py:937-942): mean of three L2-normalised maps at the first one's size
// out = (l2n(g0) + resize(l2n(g1)) + resize(l2n(g2))) / 3 in ONE pass over g0 / out: the inverse norms of the two coarser maps come from a
// small pre-pass (fp32 [N,h,w] each), the normalised coarse maps and both resized copies are never written.
template <typename T>
__global__ void k_invnorm(const T* __restrict__ x, float* __restrict__ inv, int64_t M, int C, float eps) {
    const int LP = C >> 2;
    const int64_t total = M * LP;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t rounds = (total + stride - 1) / stride;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t it = 0; it < rounds; ++it, i += stride) {
        const bool ok = i < total;
        const int64_t ii = ok ? i : 0;
        const f4 v = ld4(x + ii * 4);
        float ss = v.v[0] * v.v[0] + v.v[1] * v.v[1] + v.v[2] * v.v[2] + v.v[3] * v.v[3];
        for (int o = LP >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        if (ok && (ii % LP) == 0) inv[ii / LP] = 1.f / fmaxf(sqrtf(ss), eps);
    }
}
#define NA_ROWS 8
template <typename T>
__global__ void k_normadd_fwd(const T* __restrict__ g0, const T* __restrict__ g1, const T* __restrict__ g2, const float* __restrict__ inv1,
                              const float* __restrict__ inv2, T* __restrict__ out, int N, int H, int W, int C, int h1, int w1, int h2, int w2,
                              float eps) {
    const int LP = C >> 2;                                  // lanes per pixel (power of two <= 64): the pixel's norm is a lane-group reduce
    const unsigned lb = xcd_band(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);     // neighbouring bands on ONE XCD (common.h)
    const int bx = (int)(lb % gridDim.x), by = (int)(lb / gridDim.x);
    const int i = bx * blockDim.x + threadIdx.x;            // blockDim.x is a multiple of 64 -> lane groups stay inside a wave
    const bool ok = i < W * LP;
    const int wo = ok ? i / LP : 0, c = ok ? (i - wo * LP) * 4 : 0;
    const Lerp b1 = src_index(wo, (float)w1 / (float)W, w1, 0), b2 = src_index(wo, (float)w2 / (float)W, w2, 0);
    // a block walks a BAND of NA_ROWS consecutive output rows: the two coarse maps contribute 2 + 2 source rows to every output row, and with
    // neighbouring rows spread over blocks on different XCDs those re-reads went to HBM (PMC: 1 805 MB moved for 1 045 MB algorithmic)
    for (int row0 = by * NA_ROWS; row0 < N * H; row0 += gridDim.y * NA_ROWS)
    for (int row = row0; row < min(row0 + NA_ROWS, N * H); ++row) {
        const int n = row / H, ho = row - n * H;
        const f4 v = ld4(g0 + ((int64_t)row * W + wo) * C + c);
        float ss = v.v[0] * v.v[0] + v.v[1] * v.v[1] + v.v[2] * v.v[2] + v.v[3] * v.v[3];
        for (int o = LP >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        const float d0 = fmaxf(sqrtf(ss), eps);
        const Lerp a1 = src_index(ho, (float)h1 / (float)H, h1, 0), a2 = src_index(ho, (float)h2 / (float)H, h2, 0);
        f4 r;
        {
            const int64_t r0 = ((int64_t)n * h1 + a1.i0) * w1, r1 = ((int64_t)n * h1 + a1.i1) * w1;
            const f4 x00 = ld4(g1 + (r0 + b1.i0) * C + c), x01 = ld4(g1 + (r0 + b1.i1) * C + c);
            const f4 x10 = ld4(g1 + (r1 + b1.i0) * C + c), x11 = ld4(g1 + (r1 + b1.i1) * C + c);
            const float i00 = inv1[r0 + b1.i0], i01 = inv1[r0 + b1.i1], i10 = inv1[r1 + b1.i0], i11 = inv1[r1 + b1.i1];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r.v[k] = v.v[k] / d0 + (a1.l0 * (b1.l0 * (x00.v[k] * i00) + b1.l1 * (x01.v[k] * i01)) + a1.l1 * (b1.l0 * (x10.v[k] * i10) + b1.l1 * (x11.v[k] * i11)));
        }
        {
            const int64_t r0 = ((int64_t)n * h2 + a2.i0) * w2, r1 = ((int64_t)n * h2 + a2.i1) * w2;
            const f4 x00 = ld4(g2 + (r0 + b2.i0) * C + c), x01 = ld4(g2 + (r0 + b2.i1) * C + c);
            const f4 x10 = ld4(g2 + (r1 + b2.i0) * C + c), x11 = ld4(g2 + (r1 + b2.i1) * C + c);
            const float i00 = inv2[r0 + b2.i0], i01 = inv2[r0 + b2.i1], i10 = inv2[r1 + b2.i0], i11 = inv2[r1 + b2.i1];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r.v[k] = (r.v[k] + (a2.l0 * (b2.l0 * (x00.v[k] * i00) + b2.l1 * (x01.v[k] * i01)) + a2.l1 * (b2.l0 * (x10.v[k] * i10) + b2.l1 * (x11.v[k] * i11)))) * (1.f / 3.f);
        }
        if (ok) st4(out + ((int64_t)row * W + wo) * C + c, r);
    }
}
/* g0 [N,H,W,C], g1 [N,h1,w1,C], g2 [N,h2,w2,C] -> out [N,H,W,C]; inv1 / inv2: fp32 workspaces [N*h1*w1] / [N*h2*w2] (written here) */
// The network's case -- g1 at half, g2 at a quarter of g0's height, H a multiple of 8 -- in bands of 8 output rows: a band reads six rows of g1 and four
// of g2; each is interpolated HORIZONTALLY once (normalised on the fly) and kept in registers, the eight output rows only blend two cached rows per map.
// The row-by-row kernel above issued 17 loads per output pixel and lane (k_normadd_fwd: 0.55 ms for 1 045 MB, 0.22 of the HBM peak -- load-issue bound);
// here it is 6 per pixel.  Same arithmetic in the same association: the vertical weights still come from src_index(), only the row slots are
// compile-time (output row 8k + r blends cache rows ((r + 1) >> 1, +1) of g1 and ((r + 2) >> 2, +1) of g2; rows beyond the map are clamped at load time,
// where src_index() gives the clamped row a zero weight or the same row twice).
template <typename T>
__global__ void k_normadd_fwd_band(const T* __restrict__ g0, const T* __restrict__ g1, const T* __restrict__ g2, const float* __restrict__ inv1,
                                   const float* __restrict__ inv2, T* __restrict__ out, int N, int H, int W, int C, int h1, int w1, int h2, int w2,
                                   float eps) {
    const int LP = C >> 2;
    const unsigned lb = xcd_band(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int bx = (int)(lb % gridDim.x), by = (int)(lb / gridDim.x);
    const int i = bx * blockDim.x + threadIdx.x;            // blockDim.x is a multiple of 64 -> lane groups stay inside a wave
    const bool ok = i < W * LP;
    const int wo = ok ? i / LP : 0, c = ok ? (i - wo * LP) * 4 : 0;
    const Lerp b1 = src_index(wo, (float)w1 / (float)W, w1, 0), b2 = src_index(wo, (float)w2 / (float)W, w2, 0);
    const int bands = H >> 3;
    for (int band = by; band < N * bands; band += gridDim.y) {
        const int n = band / bands, kb = band - n * bands;
        f4 hx1[6], hx2[4];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int row = min(max(4 * kb - 1 + j, 0), h1 - 1);
            const int64_t r0 = ((int64_t)n * h1 + row) * w1;
            const f4 x0 = ld4(g1 + (r0 + b1.i0) * C + c), x1 = ld4(g1 + (r0 + b1.i1) * C + c);
            const float i0 = inv1[r0 + b1.i0], i1 = inv1[r0 + b1.i1];
#pragma unroll
            for (int k = 0; k < 4; ++k) hx1[j].v[k] = b1.l0 * (x0.v[k] * i0) + b1.l1 * (x1.v[k] * i1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = min(max(2 * kb - 1 + j, 0), h2 - 1);
            const int64_t r0 = ((int64_t)n * h2 + row) * w2;
            const f4 x0 = ld4(g2 + (r0 + b2.i0) * C + c), x1 = ld4(g2 + (r0 + b2.i1) * C + c);
            const float i0 = inv2[r0 + b2.i0], i1 = inv2[r0 + b2.i1];
#pragma unroll
            for (int k = 0; k < 4; ++k) hx2[j].v[k] = b2.l0 * (x0.v[k] * i0) + b2.l1 * (x1.v[k] * i1);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int ho = 8 * kb + r;
            const int64_t row = (int64_t)n * H + ho;
            const f4 v = ld4(g0 + (row * W + wo) * C + c);
            float ss = v.v[0] * v.v[0] + v.v[1] * v.v[1] + v.v[2] * v.v[2] + v.v[3] * v.v[3];
            ss = lane_group_sum(ss, LP);        // (another summation order than the xor butterfly of the row-by-row kernel: last-bit differences)
            const float id0 = 1.f / fmaxf(sqrtf(ss), eps);          // one division per pixel (the row-by-row kernel divides every channel: 0.5 ulp apart)
            const Lerp a1 = src_index(ho, (float)h1 / (float)H, h1, 0), a2 = src_index(ho, (float)h2 / (float)H, h2, 0);
            const f4 &p0 = hx1[(r + 1) >> 1], &p1 = hx1[((r + 1) >> 1) + 1], &q0 = hx2[(r + 2) >> 2], &q1 = hx2[((r + 2) >> 2) + 1];
            f4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float t = v.v[k] * id0 + (a1.l0 * p0.v[k] + a1.l1 * p1.v[k]);
                o.v[k] = (t + (a2.l0 * q0.v[k] + a2.l1 * q1.v[k])) * (1.f / 3.f);
            }
            if (ok) st4(out + (row * W + wo) * C + c, o);
        }
    }
}
extern "C" int tcct_normadd_fwd(const void* g0, const void* g1, const void* g2, float* inv1, float* inv2, void* out, int N, int H, int W,
                                int C, int h1, int w1, int h2, int w2, float eps, int dtype, tcct_stream_t stream) {
    const int LP = C / 4;
    TCCT_CHECK(C % 4 == 0 && LP >= 1 && LP <= 64 && (LP & (LP - 1)) == 0, "normadd_fwd: C=%d unsupported", C);
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1 && h1 >= 1 && w1 >= 1 && h2 >= 1 && w2 >= 1 && inv1 && inv2, "normadd_fwd: bad shapes / NULL workspace");
    hipStream_t st = (hipStream_t)stream;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_invnorm<T>, dim3(tcct_grid((int64_t)N * h1 * w1 * LP, PB, 1 << 16)), dim3(PB), 0, st, (const T*)g1, inv1, (int64_t)N * h1 * w1, C, eps));
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_invnorm<T>, dim3(tcct_grid((int64_t)N * h2 * w2 * LP, PB, 1 << 16)), dim3(PB), 0, st, (const T*)g2, inv2, (int64_t)N * h2 * w2, C, eps));
    static int banded = -1;         // compile-time A/B switch (0: the row-by-row kernel for every shape)
    if (banded < 0) banded = 1;
    if (banded && H == 2 * h1 && H == 4 * h2 && H % 8 == 0) {
        TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_normadd_fwd_band<T>, row_grid(W * LP, (int64_t)N * (H / 8), 8192), dim3(PB), 0, st, (const T*)g0, (const T*)g1, (const T*)g2, inv1, inv2, (T*)out, N, H, W, C, h1, w1, h2, w2, eps));
        TCCT_LAUNCH_OK();
    }
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_normadd_fwd<T>, row_grid(W * LP, ((int64_t)N * H + NA_ROWS - 1) / NA_ROWS, 8192), dim3(PB), 0, st, (const T*)g0, (const T*)g1, (const T*)g2, inv1, inv2, (T*)out, N, H, W, C, h1, w1, h2, w2, eps));
    TCCT_LAUNCH_OK();
}
