// MHCABlock's token mixer with its LayerNorm in ONE pass each way (round 4).  Reference nets/tcct.py:457-465 (MHCABlock.forward) with att = MetaPool (:405-415, :449):
//     t1 = t + dp(pool(LN1(t)) - LN1(t)),      pool = AvgPool2d(3, 1, 1, count_include_pad=False) over the [tokens, channels] plane of each image
// Until round 3 this was LayerNorm (read t, write a) + mixer (read a, read t, write t1) forward and mixer^T (read dt1, write da) + LayerNorm backward
// (read t, read da, read dt1, write dt) backward: 11 tensor passes of 226 MB at stage 0 of the bench shape.  Here a group of LP lanes owns a strip of
// consecutive tokens and marches down it with the three normalised rows n-1, n, n+1 in registers: forward = read t, write t1; backward = read dt1, read t,
// write dt: 5 passes.  Lanes hold 8 channels (16-byte accesses: the 8-byte form of the older kernels runs at 0.54-0.70 of the 16-byte rate) and keep two
// rows of loads in flight ahead of the row they work on.
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

// ------------------------------------------------------------------------------------------------ forward
// t, y: [B, N, C]; gamma, beta: [C]; scale: fp32 [B] or NULL (DropPath mask / keep); mean_rstd: fp32 [B*N][2] (written; read by the backward kernel)
template <typename T, int LP>
__global__ void __launch_bounds__(LP_T)
k_ln_metapool_fwd(const T* __restrict__ t, T* __restrict__ y, int N, int C, int S, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                  const float* __restrict__ scale, float* __restrict__ mean_rstd) {
    constexpr int GPB = LP_T / LP;
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
    const int64_t img = (int64_t)blockIdx.y * N;
    const T* tb = t + img * C + c0;
    T* yb = y + img * C + c0;
    const float sc = scale ? scale[blockIdx.y] : 1.f;
    const float invC = 1.f / (float)C;
    const int n0 = (blockIdx.x * GPB + grp) * S;            // every lane of a group shares n0: the lane exchanges stay convergent
    auto fetch = [&](int nn) { return (act && nn >= 0 && nn < N) ? ld8(tb + (int64_t)nn * C) : f8zero(); };
    auto normed = [&](const f8& x, int nn, f8& ctr) {        // LayerNorm of one row, rounded to T; zeros outside the image
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += x.v[k];
        s = lane_group_sum(s, LP);
        const float mean = s * invC;
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float d = act ? x.v[k] - mean : 0.f; q += d * d; }
        q = lane_group_sum(q, LP);
        const float rstd = rsqrtf(q * invC + eps);
        const bool in = nn >= 0 && nn < N;
        if (in && gl == 0 && nn >= n0 && nn < n0 + S) { mean_rstd[2 * (img + nn)] = mean; mean_rstd[2 * (img + nn) + 1] = rstd; }
#pragma unroll
        for (int k = 0; k < 8; ++k) ctr.v[k] = (in && act) ? round_to<T>((x.v[k] - mean) * rstd * gam[k] + bet[k]) : 0.f;
        return widen<LP>(ctr, hasl, hasr);
    };
    if (n0 >= N) return;                                   // (whole groups only: n0 is group-uniform)
    f8 x0 = fetch(n0 - 1), x1 = fetch(n0), x2 = fetch(n0 + 1), x3 = fetch(n0 + 2);
    f8 cprev, ccur, cnxt;
    Row10 prev = normed(x0, n0 - 1, cprev), cur = normed(x1, n0, ccur);
    f8 e = x1;                                              // the residual: the raw row n
    for (int i = 0; i < S; ++i) {
        const int n = n0 + i;
        const f8 x4 = fetch(n + 3);                         // two rows of loads in flight ahead of the row being normalised
        const Row10 nxt = normed(x2, n + 1, cnxt);
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
        }
        prev = cur; cur = nxt; ccur = cnxt;
        e = x2; x2 = x3; x3 = x4;
    }
}

// ------------------------------------------------------------------------------------------------ backward
// dt = dy + LN1^T(da),  da = scale[b] * (pool^T(dy) - dy) rounded to T;  dgamma / dbeta are ACCUMULATED (zero on entry)
template <typename T, int LP>
__global__ void __launch_bounds__(LP_T)
k_ln_metapool_bwd(const T* __restrict__ t, const T* __restrict__ dy, T* __restrict__ dt, int N, int C, int S, const float* __restrict__ gamma,
                  const float* __restrict__ scale, const float* __restrict__ mean_rstd, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int GPB = LP_T / LP;
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
    auto fetch = [&](int nn) { return (act && nn >= 0 && nn < N) ? ld8(db + (int64_t)nn * C) : f8zero(); };
    auto weighted = [&](const f8& d, int nn) {               // the row's gradient divided by the window sizes of ITS OWN position (pool^T)
        Row10 o = widen<LP>(d, hasl, hasr);
        const float rw = 1.f / (float)(1 + (nn > 0) + (nn < N - 1));
#pragma unroll
        for (int j = 0; j < 10; ++j) o.g[j] *= rw * cs[j];
        return o;
    };
    if (n0 < N) {
        f8 d0 = fetch(n0 - 1), d1 = fetch(n0), d2 = fetch(n0 + 1), d3 = fetch(n0 + 2);
        f8 xc = (act && n0 < N) ? ld8(tb + (int64_t)n0 * C) : f8zero(), xn = (act && n0 + 1 < N) ? ld8(tb + (int64_t)(n0 + 1) * C) : f8zero();
        Row10 prev = weighted(d0, n0 - 1), cur = weighted(d1, n0);
        f8 dcur = d1;
        for (int i = 0; i < S; ++i) {
            const int n = n0 + i;
            const f8 d4 = fetch(n + 3);
            const f8 xn2 = (act && n + 2 < N) ? ld8(tb + (int64_t)(n + 2) * C) : f8zero();
            const Row10 nxt = weighted(d2, n + 1);
            if (n < N) {
                const float mean = mean_rstd[2 * (img + n)], rstd = mean_rstd[2 * (img + n) + 1];
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
            prev = cur; cur = nxt; dcur = d2;
            d2 = d3; d3 = d4;
            xc = xn; xn = xn2;
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

static bool ln_metapool_shape_ok(int B, int64_t N, int C) { return C % 8 == 0 && C >= 16 && C <= 128 && B >= 1 && B <= 65535 && N >= 1 && N < (1LL << 30); }
#define LNP_STRIP 32
/* y = t + scale[b] * (pool(a) - a), a = LayerNorm(t; gamma, beta, eps) rounded to the activation type: MHCABlock's first half (nets/tcct.py:457-465 with the
 * MetaPool mixer :405-415) in one pass.  t, y [B,N,C] (dtype 0 fp32 / 1 bf16), C a multiple of 8 in 16..128; scale fp32 [B] or NULL; mean_rstd fp32 [B*N*2] out. */
extern "C" int tcct_ln_metapool_residual_fwd(const void* t, void* y, int B, int64_t N, int C, const float* gamma, const float* beta, float eps, const float* scale,
                                             float* mean_rstd, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(ln_metapool_shape_ok(B, N, C), "ln_metapool_residual_fwd: B=%d N=%lld C=%d unsupported (C %% 8 == 0, 16 <= C <= 128)", B, (long long)N, C);
    TCCT_CHECK(t && y && gamma && beta && mean_rstd, "ln_metapool_residual_fwd: NULL argument");
    const int64_t strips = (N + LNP_STRIP - 1) / LNP_STRIP;
#define LNPF(LP_) hipLaunchKernelGGL((k_ln_metapool_fwd<T, LP_>), dim3((unsigned)((strips + LP_T / LP_ - 1) / (LP_T / LP_)), (unsigned)B), dim3(LP_T), 0, (hipStream_t)stream, \
                                     (const T*)t, (T*)y, (int)N, C, LNP_STRIP, gamma, beta, eps, scale, mean_rstd)
    TCCT_DISPATCH(dtype, if (C <= 64) LNPF(8); else LNPF(16));
#undef LNPF
    TCCT_LAUNCH_OK();
}
/* dt = dy + LN^T(da), da = scale[b] * (pool^T(dy) - dy); dgamma, dbeta [C] overwritten */
extern "C" int tcct_ln_metapool_residual_bwd(const void* t, const void* dy, void* dt, int B, int64_t N, int C, const float* gamma, const float* scale,
                                             const float* mean_rstd, float* dgamma, float* dbeta, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(ln_metapool_shape_ok(B, N, C), "ln_metapool_residual_bwd: B=%d N=%lld C=%d unsupported (C %% 8 == 0, 16 <= C <= 128)", B, (long long)N, C);
    TCCT_CHECK(t && dy && dt && gamma && mean_rstd && dgamma && dbeta, "ln_metapool_residual_bwd: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && (hipMemsetAsync(dgamma, 0, sizeof(float) * C, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * C, st) != hipSuccess)) {
        tcct_set_error("ln_metapool_residual_bwd: memset failed"); return -2;
    }
    const int64_t strips = (N + LNP_STRIP - 1) / LNP_STRIP;
#define LNPB(LP_) hipLaunchKernelGGL((k_ln_metapool_bwd<T, LP_>), dim3((unsigned)((strips + LP_T / LP_ - 1) / (LP_T / LP_)), (unsigned)B), dim3(LP_T), 0, st, (const T*)t, \
                                     (const T*)dy, (T*)dt, (int)N, C, LNP_STRIP, gamma, scale, mean_rstd, dgamma, dbeta)
    TCCT_DISPATCH(dtype, if (C <= 64) LNPB(8); else LNPB(16));
#undef LNPB
    TCCT_LAUNCH_OK();
}
