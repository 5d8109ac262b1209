// Shared device helpers for the TCCT gfx950 kernels.  CDNA4 only: 64-wide wavefronts are assumed throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tcct_hip.h"

typedef __hip_bfloat16 bf16;

void tcct_set_error(const char* fmt, ...);
// launch census (tests only: "did the row-stream / chain kernel really run for this shape?"): host-side counters bumped next to the launch
enum { TCCT_CENSUS_CHAIN33 = 0, TCCT_CENSUS_WGRADK_STREAM = 1, TCCT_CENSUS_WGRAD33_STREAM = 2, TCCT_CENSUS_FWD33_STREAM = 3, TCCT_CENSUS_N = 8 };
void tcct_census_hit(int which);
int tcct_skip_zero_fill();      // 1: the caller guarantees accumulation outputs are already zero (tcct_set_outputs_prezeroed)

#define TCCT_CHECK(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            tcct_set_error(__VA_ARGS__);                        \
            return -1;                                          \
        }                                                       \
    } while (0)

#define TCCT_LAUNCH_OK()                                                        \
    do {                                                                        \
        hipError_t e__ = hipGetLastError();                                     \
        if (e__ != hipSuccess) {                                                \
            tcct_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return -2;                                                          \
        }                                                                       \
        return 0;                                                               \
    } while (0)

// dtype dispatch: F is a generic lambda-like macro body using type T
#define TCCT_DISPATCH(dtype, ...)                                               \
    do {                                                                        \
        if ((dtype) == TCCT_F32) { typedef float T; __VA_ARGS__; }              \
        else if ((dtype) == TCCT_BF16) { typedef bf16 T; __VA_ARGS__; }         \
        else { tcct_set_error("%s: bad dtype %d", __func__, (int)(dtype)); return -1; } \
    } while (0)

static inline int tcct_grid(int64_t work_items, int block, int cap = 256 * 16) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ---------------------------------------------------------------- scalar / vec4 loads (compute is fp32)
__device__ __forceinline__ float ldf(const float* p) { return *p; }
__device__ __forceinline__ float ldf(const bf16* p) { return __bfloat162float(*p); }
__device__ __forceinline__ void stf(float* p, float v) { *p = v; }
__device__ __forceinline__ void stf(bf16* p, float v) { *p = __float2bfloat16(v); }

struct f4 { float v[4]; };

__device__ __forceinline__ f4 ld4(const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    f4 r; r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w; return r;
}
__device__ __forceinline__ f4 ld4(const bf16* p) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    f4 r;
    r.v[0] = __uint_as_float(t.x << 16); r.v[1] = __uint_as_float(t.x & 0xffff0000u);
    r.v[2] = __uint_as_float(t.y << 16); r.v[3] = __uint_as_float(t.y & 0xffff0000u);
    return r;
}
__device__ __forceinline__ void st4(float* p, const f4& a) {
    *reinterpret_cast<float4*>(p) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
// two fp32 -> one dword of two bf16 (round to nearest even): ONE `v_cvt_pk_bf16_f32` on gfx950.  The earlier form (two __float2bfloat16 and
// shift / or by hand) compiled to two converts + lshl + and + or3 per dword: 5 VALU instructions where 1 does, in the epilogue of every
// bf16-storing kernel (the MFMA convolutions issue VALU and MFMA through the same port: 160 of ~600 epilogue instructions per tile).
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_hw v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ void st4(bf16* p, const f4& a) {
    uint2 t; t.x = pack_bf16x2(a.v[0], a.v[1]); t.y = pack_bf16x2(a.v[2], a.v[3]);
    *reinterpret_cast<uint2*>(p) = t;
}
__device__ __forceinline__ f4 f4zero() { f4 r; r.v[0] = r.v[1] = r.v[2] = r.v[3] = 0.f; return r; }

// VEC consecutive elements as fp32 (VEC = 1, 4 or 8).  bf16 with VEC = 8 moves 16 bytes per lane: the streaming kernels were written
// with 4-element vectors, i.e. 8-byte bf16 accesses (512 B per wave instruction), which holds a pure copy-like kernel at ~4.7 TB/s
// where 16-byte accesses reach ~5.5 (MI355X_MICROARCH.md: 8-B accesses run at 0.54-0.70 of the 16-B rate)
template <int VEC> __device__ __forceinline__ void ldv(const float* p, float* o) {
    if (VEC == 1) { o[0] = *p; return; }
#pragma unroll
    for (int j = 0; j < VEC / 4; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(p + 4 * j);
        o[4 * j] = t.x; o[4 * j + 1] = t.y; o[4 * j + 2] = t.z; o[4 * j + 3] = t.w;
    }
}
template <int VEC> __device__ __forceinline__ void ldv(const bf16* p, float* o) {
    if (VEC == 1) { o[0] = __bfloat162float(*p); return; }
    if (VEC == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);
        o[0] = __uint_as_float(t.x << 16); o[1] = __uint_as_float(t.x & 0xffff0000u);
        o[2] = __uint_as_float(t.y << 16); o[3] = __uint_as_float(t.y & 0xffff0000u);
        return;
    }
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[2 * j] = __uint_as_float(w[j] << 16); o[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
}
template <int VEC> __device__ __forceinline__ void stv(float* p, const float* v) {
    if (VEC == 1) { *p = v[0]; return; }
#pragma unroll
    for (int j = 0; j < VEC / 4; ++j) *reinterpret_cast<float4*>(p + 4 * j) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
}
template <int VEC> __device__ __forceinline__ void stv(bf16* p, const float* v) {
    if (VEC == 1) { *p = __float2bfloat16(v[0]); return; }
    if (VEC == 4) { uint2 t; t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]); *reinterpret_cast<uint2*>(p) = t; return; }
    uint4 t;
    t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]); t.z = pack_bf16x2(v[4], v[5]); t.w = pack_bf16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = t;
}

// ---------------------------------------------------------------- activations (kinds: TCCT_ACT_*)
// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 round-off level): 1 rcp + 1 exp + 6 FMAs instead of the ~40
// instruction libm erff -- the GELU junction kernels of the CNN blocks were VALU-bound on it (0.49 ms for a 1.36 GB reduction).
// (round 6: the reciprocal is v_rcp_f32 (1 ulp) -- `__frcp_rn` compiled to the 10-instruction IEEE division sequence (v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas,
// v_div_fixup + hazard nops), a quarter of the VALU instructions of the GELU junction kernels; the polynomial's own error is 1.5e-7)
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
// Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and phi(x) from ONE exponential: erf's A&S tail factor exp(-(x/sqrt2)^2) is exp(-x^2/2) = sqrt(2 pi) phi(x)
__device__ __forceinline__ void gauss_cdf_pdf(float x, float& cdf, float& pdf) {
    const float e = __expf(-0.5f * x * x);
    const float t = __builtin_amdgcn_rcpf(1.f + 0.23164189f * fabsf(x));            // 0.3275911 / sqrt(2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float h = 0.5f * poly * e;                                    // upper tail Q(|x|)
    cdf = x >= 0.f ? 1.f - h : h;
    pdf = 0.3989422804014327f * e;
}
__device__ __forceinline__ float act_fwd(int kind, float x) {
    switch (kind) {
        case TCCT_ACT_LRELU: return x > 0.f ? x : 0.01f * x;
        case TCCT_ACT_HSWISH: return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
        case TCCT_ACT_GELU: { float c, p; gauss_cdf_pdf(x, c, p); return x * c; }
        case TCCT_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
        case TCCT_ACT_ABS: return fabsf(x);
        default: return x;
    }
}
// derivative w.r.t. the pre-activation x
__device__ __forceinline__ float act_grad(int kind, float x) {
    switch (kind) {
        case TCCT_ACT_LRELU: return x > 0.f ? 1.f : 0.01f;
        case TCCT_ACT_HSWISH: return x < -3.f ? 0.f : (x <= 3.f ? (2.f * x + 3.f) * (1.f / 6.f) : 1.f);
        case TCCT_ACT_GELU: { float c, p; gauss_cdf_pdf(x, c, p); return c + x * p; }
        case TCCT_ACT_SIGMOID: { float s = 1.f / (1.f + __expf(-x)); return s * (1.f - s); }
        case TCCT_ACT_ABS: return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f);
        default: return 1.f;
    }
}

// v[i] = post(a[i] * pre(v[i]) + b[i]) for 4 values with BLOCK-UNIFORM activation kinds: the kind is resolved once per call into
// straight-line code (act_fwd's per-element switch costs a chain of scalar branches per value, which the MFMA epilogues that fold
// an eval-mode BatchNorm cannot hide: 0.13 -> 0.30 ms on a 64->64 pointwise GEMM before this)
template <int KIND> __device__ __forceinline__ float act_c(float x) {
    if (KIND == TCCT_ACT_LRELU) return x > 0.f ? x : 0.01f * x;
    if (KIND == TCCT_ACT_HSWISH) return x * fminf(fmaxf(x + 3.f, 0.f), 6.f) * (1.f / 6.f);
    if (KIND == TCCT_ACT_GELU) { float c, p; gauss_cdf_pdf(x, c, p); return x * c; }
    if (KIND == TCCT_ACT_SIGMOID) return 1.f / (1.f + __expf(-x));
    if (KIND == TCCT_ACT_ABS) return fabsf(x);
    return x;
}
template <int PRE, int POST> __device__ __forceinline__ void affine4_c(float* v, const float* a, const float* b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = act_c<POST>(a[i] * act_c<PRE>(v[i]) + b[i]);
}
__device__ __forceinline__ void affine4(float* v, const float* a, const float* b, int pre, int post) {
    if (pre == TCCT_ACT_NONE && post == TCCT_ACT_NONE) affine4_c<TCCT_ACT_NONE, TCCT_ACT_NONE>(v, a, b);
    else if (pre == TCCT_ACT_NONE && post == TCCT_ACT_HSWISH) affine4_c<TCCT_ACT_NONE, TCCT_ACT_HSWISH>(v, a, b);
    else if (pre == TCCT_ACT_NONE && post == TCCT_ACT_LRELU) affine4_c<TCCT_ACT_NONE, TCCT_ACT_LRELU>(v, a, b);
    else if (pre == TCCT_ACT_LRELU && post == TCCT_ACT_NONE) affine4_c<TCCT_ACT_LRELU, TCCT_ACT_NONE>(v, a, b);
    else if (pre == TCCT_ACT_NONE && post == TCCT_ACT_GELU) affine4_c<TCCT_ACT_NONE, TCCT_ACT_GELU>(v, a, b);
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = act_fwd(post, a[i] * act_fwd(pre, v[i]) + b[i]);
    }
}

// Lanes of ONE wave exchanging data through LDS (per-wave transposes in the MFMA epilogues): the hardware completes a wave's LDS
// operations in order, so no s_barrier is needed -- but the COMPILER must be told that other lanes wrote in between, or it may keep a
// lane's earlier read of the same address (seen after an edit of conv_mfma.hip: the second `ds_read_b128` of the epilogue was sunk
// into the exec-masked branch of the lanes that had just written, every other lane stored its stale first-round value).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // no instruction: wave-scope ordering only constrains the optimizer
}

// Workgroup b of a launch runs on XCD b % 8 (round-robin dispatch) and every XCD has its own L2: blocks that share input rows (resize, stencils)
// should be neighbours on ONE XCD.  xcd_band maps the hardware's linear block id to a logical one such that XCD x owns the contiguous band
// [start(x), start(x) + count(x)) of logical ids (a bijection for every n).
__device__ __forceinline__ unsigned xcd_band(unsigned b, unsigned n) {
    const unsigned q = n >> 3, r = n & 7u, x = b & 7u, k = b >> 3;
    return x * q + (x < r ? x : r) + k;
}

// ---------------------------------------------------------------- wave / block reductions (wave = 64)
// source index / weights of torch's bilinear resize (F.interpolate / nn.Upsample) for output index o
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

// Sum over each aligned group of 16 lanes, result in all 16: four DPP adds (xor 1 and xor 2 as quad permutes, then mirror inside each
// 8 and inside each 16 -- after the first two steps every lane of a quad holds the quad's sum, so a mirrored partner is as good as an
// xor partner).  The __shfl_xor form is a ds_bpermute per step: address arithmetic + an LDS round trip, ~4x the instructions.
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));    // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));    // row_mirror
    return v;
}
// sum over each aligned group of LP lanes (LP a power of two <= 64), result in all of them: DPP steps up to 16 lanes, ds_bpermute beyond
__device__ __forceinline__ float lane_group_sum(float v, int LP) {
    if (LP >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    if (LP >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    if (LP >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    if (LP >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    for (int o = 16; o < LP; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// sum over the whole block (blockDim.x multiple of 64, <= 1024); result valid in every thread
__device__ __forceinline__ float block_sum(float v, float* sm /* >= 16 floats */) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sm[i];
    return r;
}
