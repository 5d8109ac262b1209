// MHCABlock's token mixer with its LayerNorm in ONE pass each way (round 4).  Reference nets/tcct.py:457-465 (MHCABlock.forward) with att = MetaPool (:405-415, :449):
//     t1 = t + dp(pool(LN1(t)) - LN1(t)),      pool = AvgPool2d(3, 1, 1, count_include_pad=False) over the [tokens, channels] plane of each image
// Until round 3 this was LayerNorm (read t, write a) + mixer (read a, read t, write t1) forward and mixer^T (read dt1, write da) + LayerNorm backward
// (read t, read da, read dt1, write dt) backward: 11 tensor passes of 226 MB at stage 0 of the bench shape.  Here a group of LP lanes owns a strip of
// consecutive tokens and marches down it with the three normalised rows n-1, n, n+1 in registers: forward = read t, write t1; backward = read dt1, read t,
// write dt: 5 passes.  Lanes hold 8 channels (16-byte accesses: the 8-byte form of the older kernels runs at 0.54-0.70 of the 16-byte rate).  A marching
// wave is latency-bound per row unless several rows are in flight: rows are requested LNP_PF iterations ahead into a ring of PACKED registers (4 VGPRs per
// bf16 row) -- the first version kept two unpacked rows ahead and loaded the saved LayerNorm statistics right where it used them: 7 us per row backward.
// The backward kernel recomputes mean / rstd from the row it reads anyway (the same instructions on the same data as forward: identical values).
// Rounding: a = LN1(t) and da are rounded to the activation type in registers, where the two-kernel form stored them -- the values that enter the pooling
// sums and the LayerNorm backward are the stored ones of the old path (up to the summation order of the LayerNorm statistics).
#include "common.h"

#define LP_T 256
template <typename T> __device__ __forceinline__ float round_to(float v);
template <> __device__ __forceinline__ float round_to<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_to<bf16>(float v) { return __bfloat162float(__float2bfloat16(v)); }

struct f8 { float v[8]; };
__device__ __forceinline__ f8 ld8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    return f8{{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}};
}
__device__ __forceinline__ f8 ld8(const bf16* p) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    return f8{{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u),
               __uint_as_float(t.z << 16), __uint_as_float(t.z & 0xffff0000u), __uint_as_float(t.w << 16), __uint_as_float(t.w & 0xffff0000u)}};
}
__device__ __forceinline__ void st8(float* p, const f8& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(a.v[4], a.v[5], a.v[6], a.v[7]);
}
__device__ __forceinline__ void st8(bf16* p, const f8& a) {
    uint4 t;
    t.x = pack_bf16x2(a.v[0], a.v[1]); t.y = pack_bf16x2(a.v[2], a.v[3]); t.z = pack_bf16x2(a.v[4], a.v[5]); t.w = pack_bf16x2(a.v[6], a.v[7]);
    *reinterpret_cast<uint4*>(p) = t;
}
__device__ __forceinline__ f8 f8zero() { return f8{{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}}; }

// ten values of one row seen from a lane: left neighbour channel, the lane's 8 channels, right neighbour channel (zeros beyond the row / the image)
struct Row10 { float g[10]; };
template <int LP>
__device__ __forceinline__ Row10 widen(const f8& m, bool hasl, bool hasr) {
    Row10 o;
    const float ls = __shfl_up(m.v[7], 1, LP), rs = __shfl_down(m.v[0], 1, LP);
    o.g[0] = hasl ? ls : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) o.g[1 + k] = m.v[k];
    o.g[9] = hasr ? rs : 0.f;
    return o;
}

template <typename T> struct Raw8;
template <> struct Raw8<bf16> { uint4 v; };
template <> struct Raw8<float> { float4 a, b; };
__device__ __forceinline__ Raw8<bf16> ldraw(const bf16* p) { return Raw8<bf16>{*reinterpret_cast<const uint4*>(p)}; }
__device__ __forceinline__ Raw8<float> ldraw(const float* p) { return Raw8<float>{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)}; }
__device__ __forceinline__ f8 unpack(const Raw8<bf16>& r) {
    const uint4 t = r.v;
    return f8{{__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u),
               __uint_as_float(t.z << 16), __uint_as_float(t.z & 0xffff0000u), __uint_as_float(t.w << 16), __uint_as_float(t.w & 0xffff0000u)}};
}
__device__ __forceinline__ f8 unpack(const Raw8<float>& r) { return f8{{r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w}}; }
template <typename T> __device__ __forceinline__ Raw8<T> raw_zero();
template <> __device__ __forceinline__ Raw8<bf16> raw_zero<bf16>() { return Raw8<bf16>{make_uint4(0u, 0u, 0u, 0u)}; }
template <> __device__ __forceinline__ Raw8<float> raw_zero<float>() { return Raw8<float>{make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)}; }

#define LNP_STRIP 32
#define LNP_PF 4                // rows requested ahead (LNP_STRIP is a multiple)
#ifndef LNP_PF_FWD
#define LNP_PF_FWD 4
#endif

// mean and 1/std of one row held by the LP lanes of a group (inactive lanes hold zeros)
template <int LP>
__device__ __forceinline__ void row_stats(const f8& x, bool act, float invC, float eps, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += x.v[k];
    s = lane_group_sum(s, LP);
    mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) { const float d = act ? x.v[k] - mean : 0.f; q += d * d; }
    q = lane_group_sum(q, LP);
    rstd = rsqrtf(q * invC + eps);
}

// ------------------------------------------------------------------------------------------------ forward
// t, y: [B, N, C]; gamma, beta: [C]; scale: fp32 [B] or NULL (DropPath mask / keep)
// LN2: also write y2 = LayerNorm(y; gamma2, beta2, eps2) of the row just produced (MHCABlock.norm2, nets/tcct.py:466: the row is complete in the group's
// registers) with its mean / rstd [B*N][2] for the separate backward kernel -- the stand-alone LayerNorm pass re-read y to do the same
template <typename T, int LP, bool LN2>
__global__ void __launch_bounds__(LP_T)       // (134 VGPRs in bf16: three waves per SIMD; forcing four with a launch bound made the kernel 40 % SLOWER)
k_ln_metapool_fwd(const T* __restrict__ t, T* __restrict__ y, int N, int C, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                  const float* __restrict__ scale, T* __restrict__ y2, const float* __restrict__ gamma2, const float* __restrict__ beta2, float eps2,
                  float* __restrict__ mean_rstd2) {
    constexpr int GPB = LP_T / LP, S = LNP_STRIP, PF = LNP_PF_FWD;
    const int gl = threadIdx.x % LP, grp = threadIdx.x / LP;
    const int c0 = gl * 8;
    const bool act = c0 < C;                    // C < 8 LP: the last lanes of a group idle (C = 96 on 16 lanes)
    const bool hasl = act && c0 > 0, hasr = act && c0 + 8 < C;
    float gam[8], bet[8], cs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        gam[k] = act ? gamma[c0 + k] : 0.f; bet[k] = act ? beta[c0 + k] : 0.f;
        const int c = c0 + k;
        cs[k] = 1.f / (float)(1 + (c > 0) + (c < C - 1));
    }
    float gam2[8], bet2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { gam2[k] = (LN2 && act) ? gamma2[c0 + k] : 0.f; bet2[k] = (LN2 && act) ? beta2[c0 + k] : 0.f; }
    const int64_t img = (int64_t)blockIdx.y * N;
    const T* tb = t + img * C + c0;
    T* yb = y + img * C + c0;
    const float sc = scale ? scale[blockIdx.y] : 1.f;
    const float invC = 1.f / (float)C;
    const int n0 = (blockIdx.x * GPB + grp) * S;            // every lane of a group shares n0: the lane exchanges stay convergent
    if (n0 >= N) return;                                   // (whole groups only)
    auto fetch = [&](int nn) { return (act && nn >= 0 && nn < N) ? ldraw(tb + (int64_t)nn * C) : raw_zero<T>(); };
    auto normed = [&](const f8& x, int nn, f8& ctr) {        // LayerNorm of one row, rounded to T; zeros outside the image
        float mean, rstd;
        row_stats<LP>(x, act, invC, eps, mean, rstd);
        const bool in = nn >= 0 && nn < N;
#pragma unroll
        for (int k = 0; k < 8; ++k) ctr.v[k] = (in && act) ? round_to<T>((x.v[k] - mean) * rstd * gam[k] + bet[k]) : 0.f;
        return widen<LP>(ctr, hasl, hasr);
    };
    const Raw8<T> r0 = fetch(n0 - 1), r1 = fetch(n0);
    Raw8<T> ring[PF];                                       // ring[j]: row n0 + 1 + j, then the row PF further down each time it is consumed
#pragma unroll
    for (int j = 0; j < PF; ++j) ring[j] = fetch(n0 + 1 + j);
    f8 cprev, ccur, cnxt;
    f8 e = unpack(r1);                                      // the residual: the raw row n
    Row10 prev = normed(unpack(r0), n0 - 1, cprev), cur = normed(e, n0, ccur);
    for (int i0 = 0; i0 < S; i0 += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int n = n0 + i0 + j;
            const f8 xn = unpack(ring[j]);
            ring[j] = fetch(n + 1 + PF);
            const Row10 nxt = normed(xn, n + 1, cnxt);
            if (n < N) {
                const float rinv = 1.f / (float)(1 + (n > 0) + (n < N - 1));
                f8 o;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float s_ = (prev.g[k] + cur.g[k] + nxt.g[k]) + (prev.g[k + 1] + cur.g[k + 1] + nxt.g[k + 1]) + (prev.g[k + 2] + cur.g[k + 2] + nxt.g[k + 2]);
                    s_ *= rinv * cs[k];
                    o.v[k] = e.v[k] + sc * (s_ - ccur.v[k]);
                }
                if (act) st8(yb + (int64_t)n * C, o);
                if (LN2) {                                  // the second LayerNorm reads the STORED row: round first
                    f8 orr;
#pragma unroll
                    for (int k = 0; k < 8; ++k) orr.v[k] = act ? round_to<T>(o.v[k]) : 0.f;
                    float mean2, rstd2;
                    row_stats<LP>(orr, act, invC, eps2, mean2, rstd2);
                    if (gl == 0) { mean_rstd2[2 * (img + n)] = mean2; mean_rstd2[2 * (img + n) + 1] = rstd2; }
                    f8 b;
#pragma unroll
                    for (int k = 0; k < 8; ++k) b.v[k] = (orr.v[k] - mean2) * rstd2 * gam2[k] + bet2[k];
                    if (act) st8(y2 + (img + n) * C + c0, b);
                }
            }
            prev = cur; cur = nxt; ccur = cnxt; e = xn;
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// dt = dy + LN1^T(da),  da = scale[b] * (pool^T(dy) - dy) rounded to T;  dgamma / dbeta are ACCUMULATED (zero on entry)
template <typename T, int LP>
__global__ void __launch_bounds__(LP_T, sizeof(T) == 2 ? 3 : 2)
k_ln_metapool_bwd(const T* __restrict__ t, const T* __restrict__ dy, T* __restrict__ dt, int N, int C, const float* __restrict__ gamma, float eps,
                  const float* __restrict__ scale, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int GPB = LP_T / LP, S = LNP_STRIP, PF = LNP_PF;
    __shared__ float swv[(LP_T / 64) * 2 * 8 * LP];
    const int gl = threadIdx.x % LP, grp = threadIdx.x / LP;
    const int c0 = gl * 8;
    const bool act = c0 < C;
    const bool hasl = act && c0 > 0, hasr = act && c0 + 8 < C;
    float gam[8], cs[10];           // 1 / (valid columns) of SOURCE columns c0-1 .. c0+8
#pragma unroll
    for (int k = 0; k < 8; ++k) gam[k] = act ? gamma[c0 + k] : 0.f;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const int c = c0 - 1 + j;
        cs[j] = 1.f / (float)(1 + (c > 0) + (c < C - 1));
    }
    const int64_t img = (int64_t)blockIdx.y * N;
    const T* tb = t + img * C + c0;
    const T* db = dy + img * C + c0;
    T* ob = dt + img * C + c0;
    const float sc = scale ? scale[blockIdx.y] : 1.f;
    const float invC = 1.f / (float)C;
    const int n0 = (blockIdx.x * GPB + grp) * S;
    float ag[8], ab[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { ag[k] = 0.f; ab[k] = 0.f; }
    auto fetchd = [&](int nn) { return (act && nn >= 0 && nn < N) ? ldraw(db + (int64_t)nn * C) : raw_zero<T>(); };
    auto fetchx = [&](int nn) { return (act && nn >= 0 && nn < N) ? ldraw(tb + (int64_t)nn * C) : raw_zero<T>(); };
    auto weighted = [&](const f8& d, int nn) {               // the row's gradient divided by the window sizes of ITS OWN position (pool^T)
        Row10 o = widen<LP>(d, hasl, hasr);
        const float rw = 1.f / (float)(1 + (nn > 0) + (nn < N - 1));
#pragma unroll
        for (int j = 0; j < 10; ++j) o.g[j] *= rw * cs[j];
        return o;
    };
    if (n0 < N) {
        const Raw8<T> q0 = fetchd(n0 - 1), q1 = fetchd(n0);
        Raw8<T> dring[PF], xring[PF];                       // dring[j]: dy row n0 + 1 + j; xring[j]: t row n0 + j
#pragma unroll
        for (int j = 0; j < PF; ++j) { dring[j] = fetchd(n0 + 1 + j); xring[j] = fetchx(n0 + j); }
        f8 dcur = unpack(q1);
        Row10 prev = weighted(unpack(q0), n0 - 1), cur = weighted(dcur, n0);
        for (int i0 = 0; i0 < S; i0 += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int n = n0 + i0 + j;
                const f8 dn = unpack(dring[j]);
                dring[j] = fetchd(n + 1 + PF);
                const f8 xc = unpack(xring[j]);
                xring[j] = fetchx(n + PF);
                const Row10 nxt = weighted(dn, n + 1);
                float mean, rstd;
                row_stats<LP>(xc, act, invC, eps, mean, rstd);          // (group-uniform control flow: the lane exchanges run for rows past the image as well)
                if (n < N) {
                    float g[8], xh[8], s1 = 0.f, s2 = 0.f;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float s_ = (prev.g[k] + cur.g[k] + nxt.g[k]) + (prev.g[k + 1] + cur.g[k + 1] + nxt.g[k + 1]) + (prev.g[k + 2] + cur.g[k + 2] + nxt.g[k + 2]);
                        const float da = act ? round_to<T>(sc * (s_ - dcur.v[k])) : 0.f;
                        const float h = act ? (xc.v[k] - mean) * rstd : 0.f;
                        xh[k] = h;
                        ag[k] += da * h;
                        ab[k] += da;
                        g[k] = da * gam[k];
                        s1 += g[k]; s2 += g[k] * h;
                    }
                    s1 = lane_group_sum(s1, LP) * invC; s2 = lane_group_sum(s2, LP) * invC;
                    f8 o;
#pragma unroll
                    for (int k = 0; k < 8; ++k) o.v[k] = rstd * (g[k] - s1 - xh[k] * s2) + dcur.v[k];
                    if (act) st8(ob + (int64_t)n * C, o);
                }
                prev = cur; cur = nxt; dcur = dn;
            }
        }
    }
    // the 64 / LP groups of a wave hold the same channels: butterfly over the lane bits above LP, one LDS slot per wave, 2C atomics per block
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        float a = ag[k], b = ab[k];
        for (int o = LP; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (lane < LP) { swv[(wv * 2 + 0) * 8 * LP + c0 + k] = a; swv[(wv * 2 + 1) * 8 * LP + c0 + k] = b; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += LP_T) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < LP_T / 64; ++w2) { a += swv[(w2 * 2 + 0) * 8 * LP + c]; b += swv[(w2 * 2 + 1) * 8 * LP + c]; }
        atomicAdd(&dgamma[c], a); atomicAdd(&dbeta[c], b);
    }
}

static bool ln_metapool_shape_ok(int B, int64_t N, int C) { return C % 8 == 0 && C >= 16 && C <= 256 && B >= 1 && B <= 65535 && N >= 1 && N < (1LL << 30); }   // (round 6: 32-lane groups for 136..256 channels: MPViT stage 3 has 160)
/* y = t + scale[b] * (pool(a) - a), a = LayerNorm(t; gamma, beta, eps) rounded to the activation type: MHCABlock's first half (nets/tcct.py:457-465 with the
 * MetaPool mixer :405-415) in one pass.  t, y [B,N,C] (dtype 0 fp32 / 1 bf16), C a multiple of 8 in 16..256 (groups of 8, 16 or 32 lanes); scale fp32 [B] or NULL. */
static int ln_metapool_fwd_impl(const void* t, void* y, int B, int64_t N, int C, const float* gamma, const float* beta, float eps, const float* scale, void* y2,
                                const float* gamma2, const float* beta2, float eps2, float* mean_rstd2, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(ln_metapool_shape_ok(B, N, C), "ln_metapool_residual_fwd: B=%d N=%lld C=%d unsupported (C %% 8 == 0, 16 <= C <= 256)", B, (long long)N, C);
    TCCT_CHECK(t && y && gamma && beta, "ln_metapool_residual_fwd: NULL argument");
    const int64_t strips = (N + LNP_STRIP - 1) / LNP_STRIP;
#define LNPF(LP_, L2) hipLaunchKernelGGL((k_ln_metapool_fwd<T, LP_, L2>), dim3((unsigned)((strips + LP_T / LP_ - 1) / (LP_T / LP_)), (unsigned)B), dim3(LP_T), 0, \
                                         (hipStream_t)stream, (const T*)t, (T*)y, (int)N, C, gamma, beta, eps, scale, (T*)y2, gamma2, beta2, eps2, mean_rstd2)
    if (y2) { TCCT_DISPATCH(dtype, if (C <= 64) LNPF(8, true); else if (C <= 128) LNPF(16, true); else LNPF(32, true)); }
    else { TCCT_DISPATCH(dtype, if (C <= 64) LNPF(8, false); else if (C <= 128) LNPF(16, false); else LNPF(32, false)); }
#undef LNPF
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_ln_metapool_residual_fwd(const void* t, void* y, int B, int64_t N, int C, const float* gamma, const float* beta, float eps, const float* scale,
                                             int dtype, tcct_stream_t stream) {
    return ln_metapool_fwd_impl(t, y, B, N, C, gamma, beta, eps, scale, nullptr, nullptr, nullptr, 0.f, nullptr, dtype, stream);
}
/* ... and y2 = LayerNorm(y; gamma2, beta2, eps2) from the same pass (MHCABlock.norm2, nets/tcct.py:466), mean_rstd2 fp32 [B*N*2] for tcct_layernorm_bwd[_add] */
extern "C" int tcct_ln_metapool_residual_ln_fwd(const void* t, void* y, void* y2, int B, int64_t N, int C, const float* gamma, const float* beta, float eps,
                                                const float* scale, const float* gamma2, const float* beta2, float eps2, float* mean_rstd2, int dtype,
                                                tcct_stream_t stream) {
    TCCT_CHECK(y2 && gamma2 && beta2 && mean_rstd2, "ln_metapool_residual_ln_fwd: NULL argument");
    return ln_metapool_fwd_impl(t, y, B, N, C, gamma, beta, eps, scale, y2, gamma2, beta2, eps2, mean_rstd2, dtype, stream);
}
/* dt = dy + LN^T(da), da = scale[b] * (pool^T(dy) - dy); dgamma, dbeta [C] overwritten.  The LayerNorm statistics are recomputed from t. */
extern "C" int tcct_ln_metapool_residual_bwd(const void* t, const void* dy, void* dt, int B, int64_t N, int C, const float* gamma, float eps, const float* scale,
                                             float* dgamma, float* dbeta, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(ln_metapool_shape_ok(B, N, C), "ln_metapool_residual_bwd: B=%d N=%lld C=%d unsupported (C %% 8 == 0, 16 <= C <= 256)", B, (long long)N, C);
    TCCT_CHECK(t && dy && dt && gamma && dgamma && dbeta, "ln_metapool_residual_bwd: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && (hipMemsetAsync(dgamma, 0, sizeof(float) * C, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * C, st) != hipSuccess)) {
        tcct_set_error("ln_metapool_residual_bwd: memset failed"); return -2;
    }
    const int64_t strips = (N + LNP_STRIP - 1) / LNP_STRIP;
#define LNPB(LP_) hipLaunchKernelGGL((k_ln_metapool_bwd<T, LP_>), dim3((unsigned)((strips + LP_T / LP_ - 1) / (LP_T / LP_)), (unsigned)B), dim3(LP_T), 0, st, (const T*)t, \
                                     (const T*)dy, (T*)dt, (int)N, C, gamma, eps, scale, dgamma, dbeta)
    TCCT_DISPATCH(dtype, if (C <= 64) LNPB(8); else if (C <= 128) LNPB(16); else LNPB(32));
#undef LNPB
    TCCT_LAUNCH_OK();
}
