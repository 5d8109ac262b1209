// Train-mode BatchNorm2d (two-stage per-channel reductions, fp64 block combine) and LayerNorm over C.
// All tensors are [M, C] row-major (NHWC with M = N*H*W).  HBM-bound: every kernel streams x (and dy) once.
#include "common.h"
#include <cstdlib>

#define NB 256
// The per-channel reduction kernels run 1024-thread blocks on a grid of at most 512: every block ends with one fp64 atomic per
// channel sum and same-address atomics are serialised at L2, so the tail grows with the NUMBER of blocks (2048 blocks of 256:
// +40 us on every call, 4096: +80 us) while the streaming rate needs ~32 waves per CU -- big blocks give both.
#define NBR 1024
// elements per thread.  4 (8-byte bf16 accesses) is what the kernels were tuned with; the 8-element instantiations (16 bytes per lane)
// exist and are correct but measured NO better on the MI355X at level 0 (32 ch, 452 MB tensors, same box): bn_apply 0.193 vs 0.19 ms,
// bn_bwd_apply 0.278 vs 0.24, bn_bwd_reduce 0.222 vs 0.172 ms -- twice the per-channel constants in registers costs occupancy in the
// 1024-thread reduction blocks, and the 8-byte form already streams at 4.7-5.2 TB/s.  They were removed with the other round-3 A/B arms.
// Also tried (round 2, tools/probe/stream_probe.hip): ONE block-contiguous 8 KB chunk per block instead of persistent blocks.  As a bare
// y = a x + b kernel that reaches 6.0 TB/s on a 452 MB tensor against 5.3 TB/s for this file's layout (hipMemcpyDtoD: 4.9), and an eval-mode
// bn_apply built that way ran 0.159 instead of 0.188 ms at level 0.  In the training step it gained nothing (kernel time 30.21 vs 30.18 ms,
// step 27.7 vs 27.7 ms): in train mode every short-lived block has to derive its coefficients from the batch sums first (a prologue of
// dependent loads + fp64 arithmetic per 8 KB of tensor, ~as many instructions as the streaming itself), and the three-stream backward
// apply at 64 channels already streams at 5.8 TB/s with persistent blocks.  Not kept.
static inline int bn_vec(int C, int dtype) {
    static int v8 = -1;
    if (v8 < 0) v8 = 0;
    return (v8 && dtype == TCCT_BF16 && C % 8 == 0) ? 8 : ((C % 4 == 0) ? 4 : 1);
}

// block-level combine of two per-thread partial vectors (thread t = row r * CV + vector cv) into fp64 atomics on sums[0..C) / sums[C..2C)
template <int VEC>
__device__ __forceinline__ void bn_block_reduce2(float* sm, const float* s, const float* q, int t, int C, int CV, int R, double* sums) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int k = 0; k < VEC; ++k) sm[t * VEC + k] = pass ? q[k] : s[k];
        __syncthreads();
        if (t < C) {            // channel c = t lives at vector cv = t / VEC, element k = t % VEC of rows r = 0..R-1
            double a = 0.0;
            const int cvv = t / VEC, k = t % VEC;
            for (int rr = 0; rr < R; ++rr) a += (double)sm[(rr * CV + cvv) * VEC + k];
            atomicAdd(&sums[pass * C + t], a);
        }
    }
}

// ------------------------------------------------------------------ BN forward statistics: sum, sum of squares
template <typename T, int VEC, int PRE = -1>
__global__ void __launch_bounds__(NBR) k_bn_stats(const T* __restrict__ x, int64_t M, int C, int pre_act, double* __restrict__ sums) {
    __shared__ float sm[NBR * VEC];     // one 32 KB array, used for the sums and then for the sums of squares (VEC = 8)
    const int CV = C / VEC;            // vectors per row
    const int R = NBR / CV;             // rows per block pass
    const int t = threadIdx.x;
    const bool active = t < R * CV;
    const int cv = t % CV, r = t / CV;
    float s[VEC], q[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = q[k] = 0.f;
    if (active) {
        for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += (int64_t)gridDim.x * R) {
            float a[VEC];
            ldv<VEC>(x + m * C + cv * VEC, a);
#pragma unroll
            for (int k = 0; k < VEC; ++k) { float u = PRE == TCCT_ACT_NONE ? a[k] : act_fwd(pre_act, a[k]); s[k] += u; q[k] += u * u; }
        }
    }
    bn_block_reduce2<VEC>(sm, s, q, t, C, CV, R, sums);
}

extern "C" int tcct_bn_stats(const void* x, int64_t M, int C, int pre_act, double* sums, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= NB, "bn_stats: C=%d unsupported (1..%d)", C, NB);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st) != hipSuccess) { tcct_set_error("bn_stats: memset failed"); return -2; }
    int vec = bn_vec(C, dtype);
    int R = NBR / (C / vec);
    int grid = tcct_grid(M, R, 512);
    if (vec == 8) hipLaunchKernelGGL((k_bn_stats<bf16, 8>), dim3(grid), dim3(NBR), 0, st, (const bf16*)x, M, C, pre_act, sums);
    else if (vec == 4 && pre_act == TCCT_ACT_NONE) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_stats<T, 4, TCCT_ACT_NONE>), dim3(grid), dim3(NBR), 0, st, (const T*)x, M, C, pre_act, sums)); }     // (no per-element kind switch: the boundary loss's 4-channel fp32 BatchNorms ran at 0.7 TB/s)
    else if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_stats<T, 4>), dim3(grid), dim3(NBR), 0, st, (const T*)x, M, C, pre_act, sums)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_stats<T, 1>), dim3(grid), dim3(NBR), 0, st, (const T*)x, M, C, pre_act, sums)); }
    TCCT_LAUNCH_OK();
}

__global__ void k_bn_finalize(const double* __restrict__ sums, int64_t M, int C, const float* __restrict__ gamma,
                              const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                              float* running_var, int64_t* nbt, float* __restrict__ mean_rstd, float* __restrict__ ab) {
    int c = threadIdx.x;
    if (c < C) {
        double mean = sums[c] / (double)M;
        double var = sums[C + c] / (double)M - mean * mean;
        if (var < 0.0) var = 0.0;
        float rstd = (float)(1.0 / sqrt(var + (double)eps));
        mean_rstd[c] = (float)mean;
        mean_rstd[C + c] = rstd;
        float a = gamma[c] * rstd;
        ab[c] = a;
        ab[C + c] = beta[c] - (float)mean * a;
        if (running_mean) {
            double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
        }
    }
    if (c == 0 && nbt) *nbt += 1;
}
extern "C" int tcct_bn_finalize(const double* sums, int64_t M, int C, const float* gamma, const float* beta, float eps,
                                float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                float* mean_rstd, float* ab, tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= NB, "bn_finalize: C=%d unsupported", C);
    hipLaunchKernelGGL(k_bn_finalize, dim3(1), dim3(NB), 0, (hipStream_t)stream, sums, M, C, gamma, beta, eps, momentum,
                       running_mean, running_var, num_batches_tracked, mean_rstd, ab);
    TCCT_LAUNCH_OK();
}

__global__ void k_bn_eval_ab(int C, const float* gamma, const float* beta, float eps, const float* rm, const float* rv,
                             float* mean_rstd, float* ab) {
    int c = threadIdx.x;
    if (c < C) {
        float rstd = 1.f / sqrtf(rv[c] + eps);
        mean_rstd[c] = rm[c]; mean_rstd[C + c] = rstd;
        float a = gamma[c] * rstd;
        ab[c] = a; ab[C + c] = beta[c] - rm[c] * a;
    }
}
extern "C" int tcct_bn_eval_ab(int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                               const float* running_var, float* mean_rstd, float* ab, tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= NB, "bn_eval_ab: C=%d unsupported", C);
    hipLaunchKernelGGL(k_bn_eval_ab, dim3(1), dim3(NB), 0, (hipStream_t)stream, C, gamma, beta, eps, running_mean,
                       running_var, mean_rstd, ab);
    TCCT_LAUNCH_OK();
}

template <int KIND> __device__ __forceinline__ float act_grad_c(float x) {
    if (KIND == TCCT_ACT_LRELU) return x > 0.f ? 1.f : 0.01f;
    if (KIND == TCCT_ACT_HSWISH) return x < -3.f ? 0.f : (x <= 3.f ? (2.f * x + 3.f) * (1.f / 6.f) : 1.f);
    if (KIND == TCCT_ACT_NONE) return 1.f;
    return act_grad(KIND, x);
}
template <int K> __device__ __forceinline__ float actf(int rt, float x) { return K < 0 ? act_fwd(rt, x) : act_c<(K < 0 ? 0 : K)>(x); }
template <int K> __device__ __forceinline__ float actg(int rt, float x) { return K < 0 ? act_grad(rt, x) : act_grad_c<(K < 0 ? 0 : K)>(x); }
// Compile-time activation kinds for the combinations the networks use (round 6): act_fwd / act_grad resolve a RUN-TIME kind per element with a chain of scalar
// compare-and-branch pairs (five per call, two or three calls per element); the generic BatchNorm kernels ran at 2.9 (backward reduction) / 3.65 (apply) TB/s on that,
// the junction kernels had the same problem in round 3.  none + Hardswish (ViT), none + none, LeakyReLU + none (CNN block5), none + LeakyReLU (decoder); -1 = run-time kinds
// (everything else).
#define BN_KINDS(CALL) do { if (pre_act == TCCT_ACT_NONE && post_act == TCCT_ACT_HSWISH) { constexpr int PRE = TCCT_ACT_NONE, POST = TCCT_ACT_HSWISH; CALL; } \
                            else if (pre_act == TCCT_ACT_NONE && post_act == TCCT_ACT_NONE) { constexpr int PRE = TCCT_ACT_NONE, POST = TCCT_ACT_NONE; CALL; } \
                            else if (pre_act == TCCT_ACT_LRELU && post_act == TCCT_ACT_NONE) { constexpr int PRE = TCCT_ACT_LRELU, POST = TCCT_ACT_NONE; CALL; } \
                            else if (pre_act == TCCT_ACT_NONE && post_act == TCCT_ACT_LRELU) { constexpr int PRE = TCCT_ACT_NONE, POST = TCCT_ACT_LRELU; CALL; } \
                            else { constexpr int PRE = -1, POST = -1; CALL; } } while (0)
// ------------------------------------------------------------------ BN apply: y = post(a*pre(x)+b)
// train-mode coefficients of channel c from the batch sums (what k_bn_finalize computes), in double like it
__device__ __forceinline__ void bn_coeff(const double* __restrict__ sums, int64_t M, int C, int c, float gamma, float beta, float eps,
                                         float& mean_f, float& rstd, float& a, float& b, double& var_out) {
    const double mean = sums[c] / (double)M;
    double var = sums[C + c] / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    rstd = (float)(1.0 / sqrt(var + (double)eps));
    mean_f = (float)mean;
    a = gamma * rstd;
    b = beta - mean_f * a;
    var_out = var;
}
struct BnTrain {        // sums != NULL: train-mode apply with the statistics finalisation folded in (no separate k_bn_finalize launch)
    const double* sums; const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; int64_t* nbt; float* mean_rstd; float* ab_out;
};
// Train-mode coefficients of ALL channels, once per block (round 6): thread c < C derives channel c (bn_coeff: two fp64 divisions, an fp64 square root and an fp64
// reciprocal, ~100 fp64 instructions) into LDS; `publish` (one block of the launch) also writes mean / rstd / a / b for the backward kernels and moves the running
// statistics.  Every thread used to derive the coefficients of ITS 4-8 channels itself: ~800 fp64 instructions in front of a loop of ~1800 fp32 ones at level 0, and
// nearly all of a thread's work at levels 3-4, where it normalises one or two rows.  Same expressions, same bits.  Ends with a block barrier (C <= NB).
__device__ __forceinline__ void bn_block_coeffs(const BnTrain& tr, int64_t M, int C, bool publish, float* __restrict__ s_a, float* __restrict__ s_b) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float mean, rstd, a, b; double var;
        bn_coeff(tr.sums, M, C, c, tr.gamma[c], tr.beta[c], tr.eps, mean, rstd, a, b, var);
        s_a[c] = a; s_b[c] = b;
        if (publish) {
            tr.mean_rstd[c] = mean; tr.mean_rstd[C + c] = rstd; tr.ab_out[c] = a; tr.ab_out[C + c] = b;
            if (tr.running_mean) {
                const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
                tr.running_mean[c] = (1.f - tr.momentum) * tr.running_mean[c] + tr.momentum * mean;
                tr.running_var[c] = (1.f - tr.momentum) * tr.running_var[c] + tr.momentum * (float)unb;
            }
            if (c == 0 && tr.nbt) *tr.nbt += 1;
        }
    }
    __syncthreads();
}
template <typename T, int VEC, int PRE = -1, int POST = -1>
__global__ void k_bn_apply(const T* __restrict__ x, T* __restrict__ y, int64_t M, int C, const float* __restrict__ ab,
                           int pre_act, int post_act, const T* __restrict__ res, BnTrain tr) {
    // res != NULL: y = post(a*pre(x)+b) + res  (InvRes `x + conv2(f)` and the tran_vit + tran_cnn sum without a separate add pass)
    // thread = fixed channel vector (per-channel scale/shift live in registers), rows strided over the grid
    const int CV = C / VEC;
    const int R = NB / CV;
    const int t = threadIdx.x;
    __shared__ float s_a[NB], s_b[NB];
    if (tr.sums) bn_block_coeffs(tr, M, C, blockIdx.x == 0, s_a, s_b);        // (block-uniform; one block publishes the coefficients and moves the running stats)
    if (t >= R * CV) return;
    const int cv = t % CV, r = t / CV;
    float a_[VEC], b_[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        const int c = cv * VEC + k;
        if (tr.sums) { a_[k] = s_a[c]; b_[k] = s_b[c]; }
        else { a_[k] = ab[c]; b_[k] = ab[C + c]; }
    }
    const int64_t step = (int64_t)gridDim.x * R;
    for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += 2 * step) {
        const int64_t m2 = m + step;
        const int64_t o1 = m * C + cv * VEC, o2 = m2 * C + cv * VEC;
        const bool two = m2 < M;
        float v1[VEC], v2[VEC], e1[VEC], e2[VEC];
        ldv<VEC>(x + o1, v1);
        if (two) ldv<VEC>(x + o2, v2);
        if (res) { ldv<VEC>(res + o1, e1); if (two) ldv<VEC>(res + o2, e2); }
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            v1[k] = actf<POST>(post_act, a_[k] * actf<PRE>(pre_act, v1[k]) + b_[k]) + (res ? e1[k] : 0.f);
            if (two) v2[k] = actf<POST>(post_act, a_[k] * actf<PRE>(pre_act, v2[k]) + b_[k]) + (res ? e2[k] : 0.f);
        }
        stv<VEC>(y + o1, v1);
        if (two) stv<VEC>(y + o2, v2);
    }
}
static int bn_apply_impl(const void* x, const void* res, void* y, int64_t M, int C, const float* ab, int pre_act, int post_act,
                         int dtype, tcct_stream_t stream, BnTrain tr = BnTrain{nullptr, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr});
/* train-mode BatchNorm apply straight from the batch sums: tcct_bn_finalize + tcct_bn_apply[_add] in ONE launch.  sums [2C] fp64
 * (from tcct_bn_stats or a fused convolution epilogue); running_mean/var/num_batches_tracked are updated (nullable), mean_rstd [2C] and
 * ab [2C] are written for the backward kernels; res nullable (y = post(BN(pre(x))) + res). */
extern "C" int tcct_bn_apply_train(const void* x, const void* res, void* y, int64_t M, int C, const double* sums, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float* mean_rstd, float* ab, int pre_act, int post_act, int dtype,
                                   tcct_stream_t stream) {
    TCCT_CHECK(sums && gamma && beta && mean_rstd && ab, "bn_apply_train: NULL argument");
    return bn_apply_impl(x, res, y, M, C, nullptr, pre_act, post_act, dtype, stream,
                         BnTrain{sums, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, mean_rstd, ab});
}
extern "C" int tcct_bn_apply(const void* x, void* y, int64_t M, int C, const float* ab, int pre_act, int post_act,
                             int dtype, tcct_stream_t stream) {
    return bn_apply_impl(x, nullptr, y, M, C, ab, pre_act, post_act, dtype, stream);
}
/* y = post(a*pre(x)+b) + res: the normalisation pass with the residual / branch sum that follows it folded in */
extern "C" int tcct_bn_apply_add(const void* x, const void* res, void* y, int64_t M, int C, const float* ab, int pre_act, int post_act,
                                 int dtype, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr, "bn_apply_add: res is NULL");
    return bn_apply_impl(x, res, y, M, C, ab, pre_act, post_act, dtype, stream);
}
static int bn_apply_impl(const void* x, const void* res, void* y, int64_t M, int C, const float* ab, int pre_act, int post_act,
                         int dtype, tcct_stream_t stream, BnTrain tr) {
    TCCT_CHECK(C >= 1 && C <= NB, "bn_apply: C=%d unsupported", C);
    int vec = bn_vec(C, dtype);
    int R = NB / (C / vec);
    int grid = tcct_grid(M, 2 * R, 256 * 16);
    hipStream_t st = (hipStream_t)stream;
    if (vec == 8) hipLaunchKernelGGL((k_bn_apply<bf16, 8>), dim3(grid), dim3(NB), 0, st, (const bf16*)x, (bf16*)y, M, C, ab, pre_act, post_act, (const bf16*)res, tr);
    else if (vec == 4 && dtype == TCCT_BF16) { BN_KINDS(hipLaunchKernelGGL((k_bn_apply<bf16, 4, PRE, POST>), dim3(grid), dim3(NB), 0, st, (const bf16*)x, (bf16*)y, M, C, ab, pre_act, post_act, (const bf16*)res, tr)); }
    else if (vec == 4 && dtype == TCCT_F32 && pre_act == TCCT_ACT_NONE && post_act == TCCT_ACT_NONE) { hipLaunchKernelGGL((k_bn_apply<float, 4, TCCT_ACT_NONE, TCCT_ACT_NONE>), dim3(grid), dim3(NB), 0, st, (const float*)x, (float*)y, M, C, ab, pre_act, post_act, (const float*)res, tr); }
    else if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_apply<T, 4>), dim3(grid), dim3(NB), 0, st, (const T*)x, (T*)y, M, C, ab, pre_act, post_act, (const T*)res, tr)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_apply<T, 1>), dim3(grid), dim3(NB), 0, st, (const T*)x, (T*)y, M, C, ab, pre_act, post_act, (const T*)res, tr)); }
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------ BN backward: reductions then apply
template <typename T, int VEC, int PRE = -1, int POST = -1>
__global__ void __launch_bounds__(NBR) k_bn_bwd_reduce(const T* __restrict__ x, const T* __restrict__ dy, int64_t M, int C,
                                const float* __restrict__ mean_rstd, const float* __restrict__ ab, int pre_act,
                                int post_act, double* __restrict__ sums) {
    __shared__ float sm[NBR * VEC];
    const int CV = C / VEC;
    const int R = NBR / CV;
    const int t = threadIdx.x;
    const bool active = t < R * CV;
    const int cv = t % CV, r = t / CV;
    float s[VEC], q[VEC], mu[VEC], rs[VEC], a_[VEC], b_[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        s[k] = q[k] = 0.f;
        int c = active ? cv * VEC + k : 0;
        mu[k] = mean_rstd[c]; rs[k] = mean_rstd[C + c]; a_[k] = ab[c]; b_[k] = ab[C + c];
    }
    if (active) {
        for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += (int64_t)gridDim.x * R) {
            int64_t off = m * C + cv * VEC;
            float xv[VEC], gv[VEC];
            ldv<VEC>(x + off, xv);
            ldv<VEC>(dy + off, gv);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                float u = actf<PRE>(pre_act, xv[k]);
                float dz = gv[k] * actg<POST>(post_act, a_[k] * u + b_[k]);
                s[k] += dz; q[k] += dz * (u - mu[k]) * rs[k];
            }
        }
    }
    bn_block_reduce2<VEC>(sm, s, q, t, C, CV, R, sums);
}
// Two train-mode BatchNorms (no activation) whose outputs were ADDED (the encoder fusion BN(tran_vit(v)) + BN(tran_cnn(c)), reference nets/tcct.py:1016-1024)
// receive the same gradient dz: their backward sums in raw form, raw_i = {sum dz, sum dz y_i}, from ONE pass over dz, y1, y2 (round 4; two reduction passes read
// dz twice)
template <typename T, int VEC>
__global__ void __launch_bounds__(NBR) k_bn_bwd_reduce2_raw(const T* __restrict__ y1, const T* __restrict__ y2, const T* __restrict__ dz, int64_t M, int C,
                                                            double* __restrict__ raw1, double* __restrict__ raw2) {
    __shared__ float sm[NBR * VEC];
    const int CV = C / VEC;
    const int R = NBR / CV;
    const int t = threadIdx.x;
    const bool active = t < R * CV;
    const int cv = t % CV, r = t / CV;
    float s[VEC], q1[VEC], q2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) s[k] = q1[k] = q2[k] = 0.f;
    if (active) {
        for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += (int64_t)gridDim.x * R) {
            const int64_t off = m * C + cv * VEC;
            float a[VEC], b[VEC], g[VEC];
            ldv<VEC>(y1 + off, a); ldv<VEC>(y2 + off, b); ldv<VEC>(dz + off, g);
#pragma unroll
            for (int k = 0; k < VEC; ++k) { s[k] += g[k]; q1[k] += g[k] * a[k]; q2[k] += g[k] * b[k]; }
        }
    }
    bn_block_reduce2<VEC>(sm, s, q1, t, C, CV, R, raw1);
    __syncthreads();
    bn_block_reduce2<VEC>(sm, s, q2, t, C, CV, R, raw2);
}
extern "C" int tcct_bn_bwd_reduce2_raw(const void* y1, const void* y2, const void* dz, int64_t M, int C, double* raw1, double* raw2, int dtype,
                                       tcct_stream_t stream) {
    TCCT_CHECK(C >= 4 && C % 4 == 0 && C <= NB && y1 && y2 && dz && raw1 && raw2, "bn_bwd_reduce2_raw: C=%d (multiple of 4, <= 256) or NULL argument", C);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && (hipMemsetAsync(raw1, 0, sizeof(double) * 2 * C, st) != hipSuccess || hipMemsetAsync(raw2, 0, sizeof(double) * 2 * C, st) != hipSuccess)) {
        tcct_set_error("bn_bwd_reduce2_raw: memset failed"); return -2;
    }
    const int R = NBR / (C / 4);
    const int grid = tcct_grid(M, R, 512);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_bwd_reduce2_raw<T, 4>), dim3(grid), dim3(NBR), 0, st, (const T*)y1, (const T*)y2, (const T*)dz, M, C, raw1, raw2));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_bn_bwd_reduce(const void* x, const void* dy, int64_t M, int C, const float* mean_rstd, const float* ab,
                                  int pre_act, int post_act, double* sums, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= NB, "bn_bwd_reduce: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st) != hipSuccess) { tcct_set_error("bn_bwd_reduce: memset failed"); return -2; }
    int vec = bn_vec(C, dtype);
    int R = NBR / (C / vec);
    int grid = tcct_grid(M, R, 512);
    if (vec == 8) hipLaunchKernelGGL((k_bn_bwd_reduce<bf16, 8>), dim3(grid), dim3(NBR), 0, st, (const bf16*)x, (const bf16*)dy, M, C, mean_rstd, ab, pre_act, post_act, sums);
    else if (vec == 4 && dtype == TCCT_BF16) { BN_KINDS(hipLaunchKernelGGL((k_bn_bwd_reduce<bf16, 4, PRE, POST>), dim3(grid), dim3(NBR), 0, st, (const bf16*)x, (const bf16*)dy, M, C, mean_rstd, ab, pre_act, post_act, sums)); }
    else if (vec == 4 && dtype == TCCT_F32 && pre_act == TCCT_ACT_NONE && post_act == TCCT_ACT_NONE) { hipLaunchKernelGGL((k_bn_bwd_reduce<float, 4, TCCT_ACT_NONE, TCCT_ACT_NONE>), dim3(grid), dim3(NBR), 0, st, (const float*)x, (const float*)dy, M, C, mean_rstd, ab, pre_act, post_act, sums); }
    else if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_bwd_reduce<T, 4>), dim3(grid), dim3(NBR), 0, st, (const T*)x, (const T*)dy, M, C, mean_rstd, ab, pre_act, post_act, sums)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_bwd_reduce<T, 1>), dim3(grid), dim3(NBR), 0, st, (const T*)x, (const T*)dy, M, C, mean_rstd, ab, pre_act, post_act, sums)); }
    TCCT_LAUNCH_OK();
}

template <typename T, int VEC, int PRE = -1, int POST = -1>
__global__ void k_bn_bwd_apply(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int64_t M, int C,
                               const float* __restrict__ mean_rstd, const float* __restrict__ ab,
                               const double* __restrict__ sums, int pre_act, int post_act, float* __restrict__ dgamma,
                               float* __restrict__ dbeta) {
    const int CV = C / VEC;
    const int R = NB / CV;
    const int t = threadIdx.x;
    if (blockIdx.x == 0 && t < C) { dbeta[t] = (float)sums[t]; dgamma[t] = (float)sums[C + t]; }
    if (t >= R * CV) return;
    const int cv = t % CV, r = t / CV;
    const float invM = 1.f / (float)M;
    float a_[VEC], b_[VEC], mu[VEC], rs[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        int c = cv * VEC + k;
        a_[k] = ab[c]; b_[k] = ab[C + c]; mu[k] = mean_rstd[c]; rs[k] = mean_rstd[C + c];
        s1[k] = (float)sums[c] * invM; s2[k] = (float)sums[C + c] * invM;
    }
    const int64_t step = (int64_t)gridDim.x * R;
    for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += step) {
        const int64_t off = m * C + cv * VEC;
        float xv[VEC], gv[VEC], o[VEC];
        ldv<VEC>(x + off, xv);
        ldv<VEC>(dy + off, gv);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float u = actf<PRE>(pre_act, xv[k]);
            float dz = gv[k] * actg<POST>(post_act, a_[k] * u + b_[k]);
            float xh = (u - mu[k]) * rs[k];
            float du = a_[k] * (dz - s1[k] - xh * s2[k]);
            o[k] = du * actg<PRE>(pre_act, xv[k]);
        }
        stv<VEC>(dx + off, o);
    }
}
extern "C" int tcct_bn_bwd_apply(const void* x, const void* dy, void* dx, int64_t M, int C, const float* mean_rstd,
                                 const float* ab, const float* gamma, const double* sums, int pre_act, int post_act,
                                 float* dgamma, float* dbeta, int dtype, tcct_stream_t stream) {
    (void)gamma;
    TCCT_CHECK(C >= 1 && C <= NB, "bn_bwd_apply: C=%d unsupported", C);
    int vec = bn_vec(C, dtype);
    int R = NB / (C / vec);
    int grid = tcct_grid(M, R, 256 * 16);
    hipStream_t st = (hipStream_t)stream;
    if (vec == 8) hipLaunchKernelGGL((k_bn_bwd_apply<bf16, 8>), dim3(grid), dim3(NB), 0, st, (const bf16*)x, (const bf16*)dy, (bf16*)dx, M, C, mean_rstd, ab, sums, pre_act, post_act, dgamma, dbeta);
    else if (vec == 4 && dtype == TCCT_BF16) { BN_KINDS(hipLaunchKernelGGL((k_bn_bwd_apply<bf16, 4, PRE, POST>), dim3(grid), dim3(NB), 0, st, (const bf16*)x, (const bf16*)dy, (bf16*)dx, M, C, mean_rstd, ab, sums, pre_act, post_act, dgamma, dbeta)); }
    else if (vec == 4 && dtype == TCCT_F32 && pre_act == TCCT_ACT_NONE && post_act == TCCT_ACT_NONE) { hipLaunchKernelGGL((k_bn_bwd_apply<float, 4, TCCT_ACT_NONE, TCCT_ACT_NONE>), dim3(grid), dim3(NB), 0, st, (const float*)x, (const float*)dy, (float*)dx, M, C, mean_rstd, ab, sums, pre_act, post_act, dgamma, dbeta); }
    else if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_bwd_apply<T, 4>), dim3(grid), dim3(NB), 0, st, (const T*)x, (const T*)dy, (T*)dx, M, C, mean_rstd, ab, sums, pre_act, post_act, dgamma, dbeta)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_bwd_apply<T, 1>), dim3(grid), dim3(NB), 0, st, (const T*)x, (const T*)dy, (T*)dx, M, C, mean_rstd, ab, sums, pre_act, post_act, dgamma, dbeta)); }
    TCCT_LAUNCH_OK();
}

// per-channel constants for consumers that rebuild du = a (dz' - S1/M - xhat S2/M) on load: du = c1 dz' + c2 y + c3 (pre-activation none)
__global__ void k_bn_bwd_coef(const double* __restrict__ sums, int raw, int64_t M, int C, const float* __restrict__ mean_rstd,
                              const float* __restrict__ ab, float* __restrict__ coef, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = threadIdx.x;
    if (c >= C) return;
    const double mu = mean_rstd[c], rs = mean_rstd[C + c];
    const double S1 = sums[c];
    const double S2 = raw ? (sums[C + c] - mu * S1) * rs : sums[C + c];        // raw: sum dz' y  ->  sum dz' (y - mu) rstd
    dbeta[c] = (float)S1;
    dgamma[c] = (float)S2;
    const double a = ab[c], s1 = S1 / (double)M, s2 = S2 / (double)M;
    coef[c] = (float)a;
    coef[C + c] = (float)(-a * s2 * rs);
    coef[2 * C + c] = (float)(a * (s2 * rs * mu - s1));
    coef[3 * C + c] = ab[c];
    coef[4 * C + c] = ab[C + c];
}
extern "C" int tcct_bn_bwd_coef(const double* sums, int raw, int64_t M, int C, const float* mean_rstd, const float* ab, float* coef,
                                float* dgamma, float* dbeta, tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= NB && M > 0, "bn_bwd_coef: C=%d unsupported", C);
    hipLaunchKernelGGL(k_bn_bwd_coef, dim3(1), dim3(NB), 0, (hipStream_t)stream, sums, raw, M, C, mean_rstd, ab, coef, dgamma, dbeta);
    TCCT_LAUNCH_OK();
}

__global__ void k_bn_sums_from_raw(const double* __restrict__ raw, const float* __restrict__ mean_rstd, int C, double* __restrict__ out) {
    const int c = threadIdx.x;
    if (c < C) { out[c] = raw[c]; out[C + c] = (raw[C + c] - (double)mean_rstd[c] * raw[c]) * (double)mean_rstd[C + c]; }
}
extern "C" int tcct_bn_sums_from_raw(const double* raw, const float* mean_rstd, int C, double* sums, tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= NB, "bn_sums_from_raw: C=%d unsupported", C);
    hipLaunchKernelGGL(k_bn_sums_from_raw, dim3(1), dim3(NB), 0, (hipStream_t)stream, raw, mean_rstd, C, sums);
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------ train-mode BatchNorm + MaxPool2d(2) (CrossResNet levels 0-3)
// The last BatchNorm of an encoder level is followed by `self.pool` AND kept as the level's output (reference nets/tcct.py:876-884, block5 at
// :820-823).  One pass writes both: z = post(BN(pre(x))) (full size) and pooled = maxpool2(z), plus one byte per (window, 4 channels) with
// the positions of the maxima (2 bits per channel; ties and NaNs resolve like tcct_maxpool2_*: first maximum in (0,0),(0,1),(1,0),(1,1)
// order).  The backward pass never materialises the gradient of z: dz = dskip + scatter(dpool) is rebuilt from its two sources inside the
// BatchNorm reduction and inside the BatchNorm apply.  Against bn_apply + maxpool2_fwd | maxpool2_bwd_add + bn_bwd_reduce + bn_bwd_apply
// that is 2.25 instead of 3.25 and 5.5 instead of 8.25 tensor passes.  Thread = (window column, 4-channel vector) as in k_maxpool2; C/4
// divides the block size, so a thread's channels are fixed.  These kernels do ~2x the arithmetic per byte of the plain BatchNorm kernels:
// the activation kinds are template constants for the combination the network uses (PRE/POST = -1: run-time kinds; the first version,
// with act_fwd's per-element switch and the window maxima recomputed in both backward kernels, ran at 2.3 - 2.9 TB/s and was SLOWER than
// the three separate kernels: 0.94 vs 0.68 ms at level 0).
__device__ __forceinline__ float rnd_as(float v, const bf16*) { return __bfloat162float(__float2bfloat16(v)); }
__device__ __forceinline__ float rnd_as(float v, const float*) { return v; }

template <typename T, int PRE, int POST>
__global__ void k_bn_pool_fwd(const T* __restrict__ x, T* __restrict__ zout, T* __restrict__ pooled, unsigned char* __restrict__ amax, int N, int H,
                              int W, int C, int pre_act, int post_act, BnTrain tr) {
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t M = (int64_t)N * H * W;
    __shared__ float s_a[NB], s_b[NB];
    bn_block_coeffs(tr, M, C, blockIdx.x == 0 && blockIdx.y == 0, s_a, s_b);
    if (i >= Wo * C4) return;
    const int wo = i / C4, c0 = (i - wo * C4) * 4;
    float a_[4], b_[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a_[k] = s_a[c0 + k]; b_[k] = s_b[c0 + k]; }
    for (int row = blockIdx.y; row < N * Ho; row += gridDim.y) {
        const int n = row / Ho, ho = row - n * Ho;
        const int64_t b00 = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + c0;
        const int64_t offs[4] = {b00, b00 + C, b00 + (int64_t)W * C, b00 + (int64_t)W * C + C};
        f4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = ld4(x + offs[q]);
        f4 m;
        unsigned am = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float z = rnd_as(actf<POST>(post_act, a_[k] * actf<PRE>(pre_act, v[q].v[k]) + b_[k]), (const T*)nullptr);
                v[q].v[k] = z;
                if (q == 0) m.v[k] = z;
                else if (z > m.v[k] || z != z) { m.v[k] = z; am = (am & ~(3u << (2 * k))) | ((unsigned)q << (2 * k)); }
            }
            st4(zout + offs[q], v[q]);
        }
        const int64_t p = (int64_t)row * Wo + wo;
        st4(pooled + p * C + c0, m);
        amax[p * C4 + (c0 >> 2)] = (unsigned char)am;
    }
}
// sums[0..C) += sum dz', sums[C..2C) += sum dz' * xhat with dz' = (dskip + scatter(dpool)) * post'(.)   (what k_bn_bwd_reduce computes from a stored dz)
template <typename T, int PRE, int POST>
__global__ void __launch_bounds__(NBR) k_bn_pool_bwd_reduce(const T* __restrict__ x, const T* __restrict__ dpool, const T* __restrict__ dskip,
                                                            const unsigned char* __restrict__ amax, int N, int H, int W, int C,
                                                            const float* __restrict__ mean_rstd, const float* __restrict__ ab, int pre_act,
                                                            int post_act, double* __restrict__ sums) {
    __shared__ float sm[NBR * 4];
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const int R = NBR / C4;
    const int t = threadIdx.x;
    const bool active = t < R * C4;
    const int cv = t % C4, r = t / C4, c0 = cv * 4;
    float s[4], q2[4], mu[4], rs[4], a_[4], b_[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        s[k] = q2[k] = 0.f;
        const int c = active ? c0 + k : 0;
        mu[k] = mean_rstd[c]; rs[k] = mean_rstd[C + c]; a_[k] = ab[c]; b_[k] = ab[C + c];
    }
    if (active) {
        for (int row = blockIdx.x; row < N * Ho; row += gridDim.x) {
            const int n = row / Ho, ho = row - n * Ho;
            for (int wo = r; wo < Wo; wo += R) {
                const int64_t b00 = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + c0;
                const int64_t offs[4] = {b00, b00 + C, b00 + (int64_t)W * C, b00 + (int64_t)W * C + C};
                f4 v[4], e[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { v[q] = ld4(x + offs[q]); e[q] = dskip ? ld4(dskip + offs[q]) : f4zero(); }
                const int64_t p = (int64_t)row * Wo + wo;
                const f4 g = ld4(dpool + p * C + c0);
                const unsigned am = amax[p * C4 + cv];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float u = actf<PRE>(pre_act, v[q].v[k]);
                        float dz = e[q].v[k] + (((am >> (2 * k)) & 3u) == (unsigned)q ? g.v[k] : 0.f);
                        if (POST != TCCT_ACT_NONE) dz *= actg<POST>(post_act, a_[k] * u + b_[k]);
                        s[k] += dz; q2[k] += dz * (u - mu[k]) * rs[k];
                    }
            }
        }
    }
    bn_block_reduce2<4>(sm, s, q2, t, C, C4, R, sums);
}
template <typename T, int PRE, int POST>
__global__ void k_bn_pool_bwd_apply(const T* __restrict__ x, const T* __restrict__ dpool, const T* __restrict__ dskip,
                                    const unsigned char* __restrict__ amax, T* __restrict__ dx, int N, int H, int W, int C,
                                    const float* __restrict__ mean_rstd, const float* __restrict__ ab, const double* __restrict__ sums,
                                    int pre_act, int post_act, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int C4 = C >> 2, Ho = H >> 1, Wo = W >> 1;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < C) { dbeta[threadIdx.x] = (float)sums[threadIdx.x]; dgamma[threadIdx.x] = (float)sums[C + threadIdx.x]; }
    if (i >= Wo * C4) return;
    const int wo = i / C4, c0 = (i - wo * C4) * 4;
    const float invM = 1.f / (float)((int64_t)N * H * W);
    float a_[4], b_[4], mu[4], rs[4], s1[4], s2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + k;
        a_[k] = ab[c]; b_[k] = ab[C + c]; mu[k] = mean_rstd[c]; rs[k] = mean_rstd[C + c];
        s1[k] = (float)sums[c] * invM; s2[k] = (float)sums[C + c] * invM;
    }
    for (int row = blockIdx.y; row < N * Ho; row += gridDim.y) {
        const int n = row / Ho, ho = row - n * Ho;
        const int64_t b00 = (((int64_t)n * H + 2 * ho) * W + 2 * wo) * C + c0;
        const int64_t offs[4] = {b00, b00 + C, b00 + (int64_t)W * C, b00 + (int64_t)W * C + C};
        f4 v[4], e[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] = ld4(x + offs[q]); e[q] = dskip ? ld4(dskip + offs[q]) : f4zero(); }
        const int64_t p = (int64_t)row * Wo + wo;
        const f4 g = ld4(dpool + p * C + c0);
        const unsigned am = amax[p * C4 + (c0 >> 2)];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float u = actf<PRE>(pre_act, v[q].v[k]);
                float dz = e[q].v[k] + (((am >> (2 * k)) & 3u) == (unsigned)q ? g.v[k] : 0.f);
                if (POST != TCCT_ACT_NONE) dz *= actg<POST>(post_act, a_[k] * u + b_[k]);
                const float xh = (u - mu[k]) * rs[k];
                o.v[k] = a_[k] * (dz - s1[k] - xh * s2[k]);
                if (PRE != TCCT_ACT_NONE) o.v[k] *= actg<PRE>(pre_act, v[q].v[k]);
            }
            st4(dx + offs[q], o);
        }
    }
}
static bool bn_pool_ok(int N, int H, int W, int C) { return C % 4 == 0 && C >= 4 && C <= NB && NB % (C / 4) == 0 && H % 2 == 0 && W % 2 == 0 && H >= 2 && W >= 2 && N >= 1; }
static inline dim3 bn_pool_grid(int per_row, int64_t rows) {
    const int gx = (per_row + NB - 1) / NB;
    int64_t gy = ((1 << 14) + gx - 1) / gx;
    if (gy > rows) gy = rows;
    if (gy > 65535) gy = 65535;
    if (gy < 1) gy = 1;
    return dim3((unsigned)gx, (unsigned)gy);
}
// the network's combination (LeakyReLU in front, nothing behind) with compile-time kinds, anything else with run-time kinds
#define BN_POOL_KINDS(CALL) do { if (pre_act == TCCT_ACT_LRELU && post_act == TCCT_ACT_NONE) { constexpr int PRE = TCCT_ACT_LRELU, POST = TCCT_ACT_NONE; CALL; } \
                                 else { constexpr int PRE = -1, POST = -1; CALL; } } while (0)
/* z [N,H,W,C] = post(BN_train(pre(x))), pooled [N,H/2,W/2,C] = MaxPool2d(2)(z) and amax [N,H/2,W/2,C/4] (one byte per window and 4 channels:
 * 2-bit position of the maximum per channel, for tcct_bn_pool_bwd) in one pass; statistics finalisation as tcct_bn_apply_train (sums fp64
 * [2C] of pre(x); running stats / num_batches_tracked updated, mean_rstd [2C] and ab [2C] written for the backward kernels).
 * Needs even H, W and C/4 dividing 256 (C = 32, 64, 128, 256 ...).  Reference nets/tcct.py:820-823 (block5) + :876-884 (`self.pool`). */
extern "C" int tcct_bn_pool_fwd_train(const void* x, void* z, void* pooled, void* amax, int N, int H, int W, int C, const double* sums,
                                      const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                                      float* running_var, int64_t* num_batches_tracked, float* mean_rstd, float* ab, int pre_act, int post_act,
                                      int dtype, tcct_stream_t stream) {
    TCCT_CHECK(bn_pool_ok(N, H, W, C), "bn_pool_fwd_train: needs even H, W and C/4 dividing 256 (got %d,%d,%d)", H, W, C);
    TCCT_CHECK(sums && gamma && beta && mean_rstd && ab && amax, "bn_pool_fwd_train: NULL argument");
    BnTrain tr{sums, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, mean_rstd, ab};
    BN_POOL_KINDS(TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_pool_fwd<T, PRE, POST>), bn_pool_grid((W / 2) * (C / 4), (int64_t)N * (H / 2)), dim3(NB), 0,
                                                          (hipStream_t)stream, (const T*)x, (T*)z, (T*)pooled, (unsigned char*)amax, N, H, W, C, pre_act, post_act, tr)));
    TCCT_LAUNCH_OK();
}
/* backward of the pair: dx = BN_backward(dskip + maxpool2_backward(dpool)) without materialising that sum; dskip nullable (the full-size
 * output had no other consumer); amax from the forward call.  sums fp64 [2C] is scratch (overwritten); dgamma / dbeta fp32 [C] overwritten. */
extern "C" int tcct_bn_pool_bwd(const void* x, const void* dpool, const void* dskip, const void* amax, void* dx, int N, int H, int W, int C,
                                const float* mean_rstd, const float* ab, int pre_act, int post_act, double* sums, float* dgamma, float* dbeta,
                                int dtype, tcct_stream_t stream) {
    TCCT_CHECK(bn_pool_ok(N, H, W, C), "bn_pool_bwd: needs even H, W and C/4 dividing 256 (got %d,%d,%d)", H, W, C);
    TCCT_CHECK(dpool != nullptr && amax != nullptr, "bn_pool_bwd: dpool / amax is NULL");
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st) != hipSuccess) { tcct_set_error("bn_pool_bwd: memset failed"); return -2; }
    const int64_t rows = (int64_t)N * (H / 2);
    const int gr = rows < 512 ? (int)rows : 512;
    BN_POOL_KINDS(TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_pool_bwd_reduce<T, PRE, POST>), dim3(gr), dim3(NBR), 0, st, (const T*)x, (const T*)dpool,
                                                          (const T*)dskip, (const unsigned char*)amax, N, H, W, C, mean_rstd, ab, pre_act, post_act, sums)));
    BN_POOL_KINDS(TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn_pool_bwd_apply<T, PRE, POST>), bn_pool_grid((W / 2) * (C / 4), rows), dim3(NB), 0, st, (const T*)x,
                                                          (const T*)dpool, (const T*)dskip, (const unsigned char*)amax, (T*)dx, N, H, W, C, mean_rstd, ab, sums,
                                                          pre_act, post_act, dgamma, dbeta)));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------ LayerNorm over C (<= 192), 16 lanes per row
#define LN_MAXCH 3   // vec4 chunks per lane: C <= 16*4*3 = 192
// LN_TOK tokens per 16-lane group and iteration.  Measured at stage 0 (1 766 400 tokens x 64 channels, same box): 1 token 0.119 ms,
// 2 tokens 0.118 ms, 4 tokens 0.147 ms -- more loads in flight per lane do NOT help this kernel (3.8 TB/s either way); what did help
// both kernels was sizing the per-lane register arrays by the row width (NCH): k_ln_bwd 0.252 -> 0.20 ms at 64 channels.
#ifndef LN_TOK
#define LN_TOK 1
#endif
template <typename T, int NCH>          // NCH = vec4 chunks per lane in use: ceil(C / 64) (statically sized register arrays)
__global__ void k_ln_fwd(const T* __restrict__ x, T* __restrict__ y, int64_t M, int C, const float* __restrict__ gamma,
                         const float* __restrict__ beta, float eps, float* __restrict__ mean_rstd) {
    const int C4 = C >> 2;
    const int lane = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    float gam[NCH][4], bet[NCH][4];       // the lane's channels are fixed: scale / shift live in registers (8 loads per row before)
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int ch = lane + 16 * j;
            gam[j][k] = ch < C4 ? gamma[ch * 4 + k] : 0.f; bet[j][k] = ch < C4 ? beta[ch * 4 + k] : 0.f;
        }
    for (int64_t m0 = grp * LN_TOK; m0 < M; m0 += ngrp * LN_TOK) {     // all 16 lanes of a group share m0: no divergence inside the lane exchanges
        f4 v[LN_TOK][NCH];
#pragma unroll
        for (int t = 0; t < LN_TOK; ++t)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int ch = lane + 16 * j;
                if (ch < C4 && m0 + t < M) v[t][j] = ld4(x + (m0 + t) * C + ch * 4); else v[t][j] = f4zero();
            }
#pragma unroll
        for (int t = 0; t < LN_TOK; ++t) {
            const int64_t m = m0 + t;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) s += v[t][j].v[0] + v[t][j].v[1] + v[t][j].v[2] + v[t][j].v[3];
            s = row16_sum(s);
            float mean = s / (float)C, q = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                int ch = lane + 16 * j;
                if (ch < C4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { float d = v[t][j].v[k] - mean; q += d * d; }
                }
            }
            q = row16_sum(q);
            float rstd = rsqrtf(q / (float)C + eps);
            if (m < M) {
                if (lane == 0) { mean_rstd[2 * m] = mean; mean_rstd[2 * m + 1] = rstd; }
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    int ch = lane + 16 * j;
                    if (ch < C4) {
                        f4 r;
#pragma unroll
                        for (int k = 0; k < 4; ++k) r.v[k] = (v[t][j].v[k] - mean) * rstd * gam[j][k] + bet[j][k];
                        st4(y + m * C + ch * 4, r);
                    }
                }
            }
        }
    }
}
extern "C" int tcct_layernorm_fwd(const void* x, void* y, int64_t M, int C, const float* gamma, const float* beta,
                                  float eps, float* mean_rstd, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && C <= 64 * LN_MAXCH, "layernorm_fwd: C=%d unsupported", C);
    const int nch = (C + 63) / 64;
#define LNF(NC) hipLaunchKernelGGL((k_ln_fwd<T, NC>), dim3(tcct_grid((M + LN_TOK - 1) / LN_TOK * 16, NB)), dim3(NB), 0, (hipStream_t)stream, (const T*)x, (T*)y, M, C, gamma, beta, eps, mean_rstd)
    TCCT_DISPATCH(dtype, if (nch == 1) LNF(1); else if (nch == 2) LNF(2); else LNF(3));
#undef LNF
    TCCT_LAUNCH_OK();
}

template <typename T, int NCH>
__global__ void __launch_bounds__(NBR) k_ln_bwd(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int64_t M, int C,
                         const float* __restrict__ gamma, const float* __restrict__ mean_rstd,
                         float* __restrict__ dgamma, float* __restrict__ dbeta, const T* __restrict__ res) {
    // res != NULL: dx = LN_bwd(dy) + res -- the gradient of the residual path around the normalisation (x + f(LN(x)), tcct.py:461-468)
    __shared__ float swv[(NBR / 64) * 384];         // per wave: dgamma partials [192], dbeta partials [192]
    const int C4 = C >> 2;
    const int lane = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    f4 ag[NCH], abt[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { ag[j] = f4zero(); abt[j] = f4zero(); }
    constexpr int TB = NCH == 1 ? 2 : 1;      // tokens per iteration (wider rows: one, the 1024-thread blocks have 128 registers per lane)
    for (int64_t mb = grp * TB; mb < M; mb += ngrp * TB) {
      // two tokens per iteration: all loads of both tokens are issued before the first reduction (bytes in flight, see k_ln_fwd)
      f4 xin[TB][NCH], din[TB][NCH], rin[TB][NCH];
      float mrs[TB][2];
#pragma unroll
      for (int t = 0; t < TB; ++t) {
          const bool tok = mb + t < M;
          mrs[t][0] = tok ? mean_rstd[2 * (mb + t)] : 0.f; mrs[t][1] = tok ? mean_rstd[2 * (mb + t) + 1] : 0.f;
#pragma unroll
          for (int j = 0; j < NCH; ++j) {
              const int ch = lane + 16 * j;
              if (ch < C4 && tok) {
                  xin[t][j] = ld4(x + (mb + t) * C + ch * 4); din[t][j] = ld4(dy + (mb + t) * C + ch * 4);
                  rin[t][j] = res ? ld4(res + (mb + t) * C + ch * 4) : f4zero();
              } else { xin[t][j] = f4zero(); din[t][j] = f4zero(); rin[t][j] = f4zero(); }
          }
      }
#pragma unroll
      for (int t = 0; t < TB; ++t) {
        const int64_t m = mb + t;
        if (m >= M) break;
        float mean = mrs[t][0], rstd = mrs[t][1];
        f4 xh[NCH], g[NCH];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int ch = lane + 16 * j;
            if (ch < C4) {
                f4 xv = xin[t][j], d = din[t][j];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float h = (xv.v[k] - mean) * rstd;
                    xh[j].v[k] = h;
                    ag[j].v[k] += d.v[k] * h;
                    abt[j].v[k] += d.v[k];
                    float dg = d.v[k] * gamma[ch * 4 + k];
                    g[j].v[k] = dg;
                    s1 += dg; s2 += dg * h;
                }
            }
        }
        s1 = row16_sum(s1); s2 = row16_sum(s2);
        s1 /= (float)C; s2 /= (float)C;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int ch = lane + 16 * j;
            if (ch < C4) {
                f4 r;
#pragma unroll
                for (int k = 0; k < 4; ++k) r.v[k] = rstd * (g[j].v[k] - s1 - xh[j].v[k] * s2);
                if (res) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) r.v[k] += rin[t][j].v[k];
                }
                st4(dx + m * C + ch * 4, r);
            }
        }
      }
    }
    // the four 16-lane row groups of a wave hold the same channels: butterfly over lane bits 4,5, then one LDS slot per wave (no
    // LDS float atomics: they cost tens of microseconds per block), summed over the waves by the first 2C threads
    const int wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        int ch = lane + 16 * j;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float a = ag[j].v[k], b = abt[j].v[k];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if ((threadIdx.x & 63) < 16 && ch < C4) { swv[wv * 384 + ch * 4 + k] = a; swv[wv * 384 + 192 + ch * 4 + k] = b; }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int w2 = 0; w2 < nwv; ++w2) { a += swv[w2 * 384 + c]; b += swv[w2 * 384 + 192 + c]; }
        atomicAdd(&dgamma[c], a); atomicAdd(&dbeta[c], b);
    }
}
static int layernorm_bwd_impl(const void* x, const void* dy, void* dx, int64_t M, int C, const float* gamma, const float* mean_rstd,
                              float* dgamma, float* dbeta, int dtype, tcct_stream_t stream, const void* res);
extern "C" int tcct_layernorm_bwd(const void* x, const void* dy, void* dx, int64_t M, int C, const float* gamma,
                                  const float* mean_rstd, float* dgamma, float* dbeta, int dtype, tcct_stream_t stream) {
    return layernorm_bwd_impl(x, dy, dx, M, C, gamma, mean_rstd, dgamma, dbeta, dtype, stream, nullptr);
}
/* dx = LN_bwd(dy) + res: res [M,C] is the gradient of the residual path around the normalisation, added in the same pass */
extern "C" int tcct_layernorm_bwd_add(const void* x, const void* dy, const void* res, void* dx, int64_t M, int C, const float* gamma,
                                      const float* mean_rstd, float* dgamma, float* dbeta, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr, "layernorm_bwd_add: res is NULL");
    return layernorm_bwd_impl(x, dy, dx, M, C, gamma, mean_rstd, dgamma, dbeta, dtype, stream, res);
}
static int layernorm_bwd_impl(const void* x, const void* dy, void* dx, int64_t M, int C, const float* gamma, const float* mean_rstd,
                              float* dgamma, float* dbeta, int dtype, tcct_stream_t stream, const void* res) {
    TCCT_CHECK(C % 4 == 0 && C <= 64 * LN_MAXCH, "layernorm_bwd: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && (hipMemsetAsync(dgamma, 0, sizeof(float) * C, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * C, st) != hipSuccess)) {
        tcct_set_error("layernorm_bwd: memset failed"); return -2;
    }
    const int nch = (C + 63) / 64;
#define LNB(NC) hipLaunchKernelGGL((k_ln_bwd<T, NC>), dim3(tcct_grid((NC == 1 ? (M + 1) / 2 : M) * 16, NBR, 512)), dim3(NBR), 0, st, (const T*)x, (const T*)dy, (T*)dx, M, C, gamma, mean_rstd, dgamma, dbeta, (const T*)res)
    TCCT_DISPATCH(dtype, if (nch == 1) LNB(1); else if (nch == 2) LNB(2); else LNB(3));
#undef LNB
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------ fused CrossCNNBlock junction (reference nets/tcct.py:825-826)
//   y = act( BN_A(pre(xa)) + BN_B(pre(xb)) )       with act = GELU, pre = LeakyReLU, both BNs in train mode.
// Forward reads the two raw conv outputs once and writes y once (instead of 2 x bn_apply + add_act: 7 tensor passes -> 3);
// backward recomputes the junction from xa, xb: one reduction pass (4 per-channel sums) + one apply pass writing both input
// gradients (13 tensor passes -> 8).  Same fixed-channel-thread layout as the plain BN kernels.
// PRE / ACT: compile-time activation kinds for the network's combination (LeakyReLU in front of both BatchNorms, GELU behind the sum), -1 =
// run-time kinds.  act_fwd / act_grad resolve the kind per ELEMENT with a chain of scalar branches; with three input streams that made the
// backward reduction VALU-bound (3.3 TB/s at level 0)
template <typename T, int PRE, int ACT>
__global__ void k_bn2_add_act_fwd(const T* __restrict__ xa, const T* __restrict__ xb, T* __restrict__ y, int64_t M, int C,
                                  const float* __restrict__ abA, const float* __restrict__ abB, int pre_act, int act, BnTrain trA, BnTrain trB) {
    const int CV = C >> 2, R = NB / CV, t = threadIdx.x;
    __shared__ float s_aA[NB], s_bA[NB], s_aB[NB], s_bB[NB];
    if (trA.sums) {             // train mode: both statistics finalisations folded into this launch, once per block (bn_block_coeffs)
        bn_block_coeffs(trA, M, C, blockIdx.x == 0, s_aA, s_bA);
        bn_block_coeffs(trB, M, C, blockIdx.x == 0, s_aB, s_bB);
    }
    if (t >= R * CV) return;
    const int cv = t % CV, r = t / CV;
    float aA[4], bA[4], aB[4], bB[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = cv * 4 + k;
        if (trA.sums) { aA[k] = s_aA[c]; bA[k] = s_bA[c]; aB[k] = s_aB[c]; bB[k] = s_bB[c]; }
        else { aA[k] = abA[c]; bA[k] = abA[C + c]; aB[k] = abB[c]; bB[k] = abB[C + c]; }
    }
    const int64_t step = (int64_t)gridDim.x * R;
    for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += step) {
        const int64_t o = m * C + cv * 4;
        f4 va = ld4(xa + o), vb = ld4(xb + o), q;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            q.v[k] = actf<ACT>(act, aA[k] * actf<PRE>(pre_act, va.v[k]) + bA[k] + aB[k] * actf<PRE>(pre_act, vb.v[k]) + bB[k]);
        st4(y + o, q);
    }
}
#define JUNC_KINDS(CALL) do { if (pre_act == TCCT_ACT_LRELU && act == TCCT_ACT_GELU) { constexpr int PRE = TCCT_ACT_LRELU, ACT = TCCT_ACT_GELU; CALL; } \
                              else { constexpr int PRE = -1, ACT = -1; CALL; } } while (0)
static int bn2_fwd_impl(const void* xa, const void* xb, void* y, int64_t M, int C, const float* abA, const float* abB,
                        int pre_act, int act, int dtype, tcct_stream_t stream, BnTrain trA, BnTrain trB) {
    TCCT_CHECK(C % 4 == 0 && C >= 4 && C <= NB, "bn2_add_act_fwd: C=%d unsupported", C);
    int R = NB / (C / 4);
    JUNC_KINDS(TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn2_add_act_fwd<T, PRE, ACT>), dim3(tcct_grid(M, R, 256 * 16)), dim3(NB), 0, (hipStream_t)stream,
                                                       (const T*)xa, (const T*)xb, (T*)y, M, C, abA, abB, pre_act, act, trA, trB)));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_bn2_add_act_fwd(const void* xa, const void* xb, void* y, int64_t M, int C, const float* abA, const float* abB,
                                    int pre_act, int act, int dtype, tcct_stream_t stream) {
    const BnTrain none{nullptr, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr};
    return bn2_fwd_impl(xa, xb, y, M, C, abA, abB, pre_act, act, dtype, stream, none, none);
}
/* train-mode junction straight from the two sets of batch sums (2 x tcct_bn_finalize + tcct_bn2_add_act_fwd in one launch);
 * statsX = {running_mean, running_var} [2][C] pointers are passed separately; mean_rstdX / abX [2C] are written for the backward */
extern "C" int tcct_bn2_add_act_train(const void* xa, const void* xb, void* y, int64_t M, int C, const double* sumsA, const float* gammaA,
                                      const float* betaA, float* running_meanA, float* running_varA, int64_t* nbtA, float* mean_rstdA,
                                      float* abA, const double* sumsB, const float* gammaB, const float* betaB, float* running_meanB,
                                      float* running_varB, int64_t* nbtB, float* mean_rstdB, float* abB, float eps, float momentum,
                                      int pre_act, int act, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(sumsA && sumsB && gammaA && gammaB && betaA && betaB && mean_rstdA && mean_rstdB && abA && abB, "bn2_add_act_train: NULL argument");
    return bn2_fwd_impl(xa, xb, y, M, C, nullptr, nullptr, pre_act, act, dtype, stream,
                        BnTrain{sumsA, gammaA, betaA, eps, momentum, running_meanA, running_varA, nbtA, mean_rstdA, abA},
                        BnTrain{sumsB, gammaB, betaB, eps, momentum, running_meanB, running_varB, nbtB, mean_rstdB, abB});
}

// sums[4C] = { sum g, sum g*xhatA, (unused alias of sum g), sum g*xhatB } laid out as [C]:g, [C]:g*xhatA, [C]:g, [C]:g*xhatB
template <typename T, int PRE, int ACT>
__global__ void __launch_bounds__(NBR) k_bn2_add_act_bwd_reduce(const T* __restrict__ xa, const T* __restrict__ xb, const T* __restrict__ dy, int64_t M, int C,
                                         const float* __restrict__ mrA, const float* __restrict__ abA, const float* __restrict__ mrB,
                                         const float* __restrict__ abB, int pre_act, int act, double* __restrict__ sums) {
    __shared__ float sm[3 * NBR * 4];
    const int CV = C >> 2, R = NBR / CV, t = threadIdx.x;
    const bool active = t < R * CV;
    const int cv = t % CV, r = t / CV;
    float aA[4], bA[4], aB[4], bB[4], muA[4], rsA[4], muB[4], rsB[4], s0[4], s1[4], s2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c = active ? cv * 4 + k : 0;
        aA[k] = abA[c]; bA[k] = abA[C + c]; aB[k] = abB[c]; bB[k] = abB[C + c];
        muA[k] = mrA[c]; rsA[k] = mrA[C + c]; muB[k] = mrB[c]; rsB[k] = mrB[C + c];
        s0[k] = s1[k] = s2[k] = 0.f;
    }
    if (active) {
        // JR rows per iteration: three input streams and one resident 1024-thread block per CU need the extra loads in flight
        constexpr int JR = 2;          // (4 rows in flight: 0.44 ms against 0.41 at level 0)
        const int64_t step = (int64_t)gridDim.x * R;
        for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += JR * step) {
            f4 va[JR], vb[JR], g[JR];
#pragma unroll
            for (int j = 0; j < JR; ++j) {
                const int64_t mj = m + j * step;
                if (mj < M) { const int64_t o = mj * C + cv * 4; va[j] = ld4(xa + o); vb[j] = ld4(xb + o); g[j] = ld4(dy + o); }
            }
#pragma unroll
            for (int j = 0; j < JR; ++j) {
                if (m + j * step < M) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float ua = actf<PRE>(pre_act, va[j].v[k]), ub = actf<PRE>(pre_act, vb[j].v[k]);
                        float dz = g[j].v[k] * actg<ACT>(act, aA[k] * ua + bA[k] + aB[k] * ub + bB[k]);
                        s0[k] += dz; s1[k] += dz * (ua - muA[k]) * rsA[k]; s2[k] += dz * (ub - muB[k]) * rsB[k];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sm[t * 4 + k] = s0[k]; sm[(NBR + t) * 4 + k] = s1[k]; sm[(2 * NBR + t) * 4 + k] = s2[k]; }
    __syncthreads();
    if (t < C) {
        double a = 0.0, b = 0.0, c2 = 0.0;
        int cvv = t >> 2, k = t & 3;
        for (int rr = 0; rr < R; ++rr) {
            int tt = rr * CV + cvv;
            a += (double)sm[tt * 4 + k]; b += (double)sm[(NBR + tt) * 4 + k]; c2 += (double)sm[(2 * NBR + tt) * 4 + k];
        }
        atomicAdd(&sums[t], a); atomicAdd(&sums[C + t], b); atomicAdd(&sums[2 * C + t], a); atomicAdd(&sums[3 * C + t], c2);
    }
}
extern "C" int tcct_bn2_add_act_bwd_reduce(const void* xa, const void* xb, const void* dy, int64_t M, int C, const float* mean_rstdA,
                                           const float* abA, const float* mean_rstdB, const float* abB, int pre_act, int act,
                                           double* sums, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && C >= 4 && C <= NB, "bn2_add_act_bwd_reduce: C=%d unsupported", C);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(sums, 0, sizeof(double) * 4 * C, st) != hipSuccess) { tcct_set_error("bn2_add_act_bwd_reduce: memset failed"); return -2; }
    int R = NBR / (C / 4);
    JUNC_KINDS(TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn2_add_act_bwd_reduce<T, PRE, ACT>), dim3(tcct_grid(M, R, 512)), dim3(NBR), 0, st, (const T*)xa,
                                                       (const T*)xb, (const T*)dy, M, C, mean_rstdA, abA, mean_rstdB, abB, pre_act, act, sums)));
    TCCT_LAUNCH_OK();
}

template <typename T, int PRE, int ACT>
__global__ void k_bn2_add_act_bwd_apply(const T* __restrict__ xa, const T* __restrict__ xb, const T* __restrict__ dy, T* __restrict__ dxa,
                                        T* __restrict__ dxb, int64_t M, int C, const float* __restrict__ mrA, const float* __restrict__ abA,
                                        const float* __restrict__ mrB, const float* __restrict__ abB, const double* __restrict__ sums,
                                        int pre_act, int act, float* __restrict__ dgA, float* __restrict__ dbA, float* __restrict__ dgB,
                                        float* __restrict__ dbB) {
    const int CV = C >> 2, R = NB / CV, t = threadIdx.x;
    if (blockIdx.x == 0 && t < C) {
        dbA[t] = (float)sums[t]; dgA[t] = (float)sums[C + t]; dbB[t] = (float)sums[2 * C + t]; dgB[t] = (float)sums[3 * C + t];
    }
    if (t >= R * CV) return;
    const int cv = t % CV, r = t / CV;
    const float invM = 1.f / (float)M;
    float aA[4], bA[4], aB[4], bB[4], muA[4], rsA[4], muB[4], rsB[4], s0[4], s1[4], s2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int c = cv * 4 + k;
        aA[k] = abA[c]; bA[k] = abA[C + c]; aB[k] = abB[c]; bB[k] = abB[C + c];
        muA[k] = mrA[c]; rsA[k] = mrA[C + c]; muB[k] = mrB[c]; rsB[k] = mrB[C + c];
        s0[k] = (float)sums[c] * invM; s1[k] = (float)sums[C + c] * invM; s2[k] = (float)sums[3 * C + c] * invM;
    }
    for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += (int64_t)gridDim.x * R) {
        const int64_t o = m * C + cv * 4;
        f4 va = ld4(xa + o), vb = ld4(xb + o), g = ld4(dy + o), qa, qb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float ua = actf<PRE>(pre_act, va.v[k]), ub = actf<PRE>(pre_act, vb.v[k]);
            float dz = g.v[k] * actg<ACT>(act, aA[k] * ua + bA[k] + aB[k] * ub + bB[k]);
            qa.v[k] = aA[k] * (dz - s0[k] - (ua - muA[k]) * rsA[k] * s1[k]) * actg<PRE>(pre_act, va.v[k]);
            qb.v[k] = aB[k] * (dz - s0[k] - (ub - muB[k]) * rsB[k] * s2[k]) * actg<PRE>(pre_act, vb.v[k]);
        }
        st4(dxa + o, qa); st4(dxb + o, qb);
    }
}
extern "C" int tcct_bn2_add_act_bwd_apply(const void* xa, const void* xb, const void* dy, void* dxa, void* dxb, int64_t M, int C,
                                          const float* mean_rstdA, const float* abA, const float* mean_rstdB, const float* abB,
                                          const double* sums, int pre_act, int act, float* dgammaA, float* dbetaA, float* dgammaB,
                                          float* dbetaB, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C % 4 == 0 && C >= 4 && C <= NB, "bn2_add_act_bwd_apply: C=%d unsupported", C);
    int R = NB / (C / 4);
    JUNC_KINDS(TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_bn2_add_act_bwd_apply<T, PRE, ACT>), dim3(tcct_grid(M, R, 256 * 16)), dim3(NB), 0, (hipStream_t)stream,
                                                       (const T*)xa, (const T*)xb, (const T*)dy, (T*)dxa, (T*)dxb, M, C, mean_rstdA, abA, mean_rstdB,
                                                       abB, sums, pre_act, act, dgammaA, dbetaA, dgammaB, dbetaB)));
    TCCT_LAUNCH_OK();
}
