// Feature-polarization loss (nets/reg.py:86-105, nets/fcs.py:25-50,63-96, nets/fcp.py:72-75) and the fused
// clip_grad_norm_ + AdamW step (kite/loop_seg.py:128-130, kite/loopback.py:127).
//
// FPL pipeline (per step, all classes at once because the label classes partition the pixels):
//   key[p]  = (label[p] << 32) | ~bits(prob_label[p])          -> ascending sort == per class, descending prob
//   sort    : rocPRIM device radix sort (36 significant bits: 4 class bits + 32 probability bits) — the one library primitive used on the path
//   binmean : sorted position r of class c (segment offset off_c, n_c pixels, N_c = n_c/32) falls in bin r/N_c
//             (tail n_c%32 dropped); 8 lanes gather one 32-channel feature row; register run-length accumulation,
//             fp32 atomics per (class, bin) flush
//   backward: dfeat[p] = dpro[label[p]][bin[p]] / N_c  (coalesced, bin map written in the forward)
#include "common.h"
#include <cstring>
#include <rocprim/rocprim.hpp>

#define FB 256
#define FPL_BINS 32
#define FPL_MAXC 16     // classes (power of two): 5 for GOALS, 9 for the reference's Duke / HCMS models

__global__ void k_fpl_keys(const uint8_t* __restrict__ lab, const float* __restrict__ prob, int64_t M,
                           unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ counts) {
    __shared__ uint32_t sc[FPL_MAXC];
    if (threadIdx.x < FPL_MAXC) sc[threadIdx.x] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M; i += (int64_t)gridDim.x * blockDim.x) {
        uint32_t l = lab[i];
        uint32_t b = ~__float_as_uint(prob[i]);           // prob >= 0: uint order == float order; ~ => descending
        keys[i] = ((unsigned long long)l << 32) | b;
        vals[i] = (uint32_t)i;
        atomicAdd(&sc[l & (FPL_MAXC - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < FPL_MAXC && sc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], sc[threadIdx.x]);
}

extern "C" int64_t tcct_fpl_sort_workspace_bytes(int64_t M) {
    size_t bytes = 0;
    unsigned long long* k = nullptr;
    uint32_t* v = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)M, 0, 36, (hipStream_t)0, false);
    if (e != hipSuccess) return -1;
    return (int64_t)bytes;
}

/* keys/vals in -> sorted out.  counts[FPL_MAXC] (uint32) receives the per-class pixel counts. */
extern "C" int tcct_fpl_sort(const uint8_t* labels, const float* prob, int64_t M, uint64_t* keys_in, uint32_t* vals_in,
                             uint64_t* keys_out, uint32_t* vals_out, uint32_t* counts, void* workspace,
                             int64_t workspace_bytes, tcct_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    TCCT_CHECK(M > 0 && M < (1LL << 32), "fpl_sort: M out of range");
    if (hipMemsetAsync(counts, 0, sizeof(uint32_t) * FPL_MAXC, st) != hipSuccess) { tcct_set_error("fpl_sort: memset failed"); return -2; }
    hipLaunchKernelGGL(k_fpl_keys, dim3(tcct_grid(M, FB, 2048)), dim3(FB), 0, st, labels, prob, M, (unsigned long long*)keys_in, vals_in, counts);
    size_t bytes = (size_t)workspace_bytes;
    hipError_t e = rocprim::radix_sort_pairs(workspace, bytes, (unsigned long long*)keys_in, (unsigned long long*)keys_out,
                                             vals_in, vals_out, (size_t)M, 0, 36, st, false);
    if (e != hipSuccess) { tcct_set_error("fpl_sort: rocprim radix_sort_pairs failed: %s", hipGetErrorString(e)); return -2; }
    TCCT_LAUNCH_OK();
}

// 8 lanes per sorted position; each 8-lane group walks RUN consecutive positions.
#define FPL_RUN 64
template <typename T>
__global__ void k_fpl_binmean(const T* __restrict__ feat /*[M,32]*/, const unsigned long long* __restrict__ keys,
                              const uint32_t* __restrict__ vals, const uint32_t* __restrict__ counts, int64_t M, int C,
                              float* __restrict__ pro_sum /*[C][32][32]*/, uint8_t* __restrict__ binmap /*[M]*/) {
    __shared__ uint32_t off[FPL_MAXC + 1];
    if (threadIdx.x == 0) {
        uint32_t a = 0;
        for (int c = 0; c < FPL_MAXC; ++c) { off[c] = a; a += counts[c]; }
        off[FPL_MAXC] = a;
    }
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    const int64_t r0 = grp * FPL_RUN;
    if (r0 >= M) return;
    const int64_t r1 = min(M, r0 + FPL_RUN);
    f4 acc = f4zero();
    int cur = -1;     // class*32 + bin of the running accumulation
    for (int64_t r = r0; r < r1; ++r) {
        int c = (int)(keys[r] >> 32);
        uint32_t pix = vals[r];
        uint32_t nc = counts[c];
        uint32_t Nb = nc / FPL_BINS;
        uint32_t rr = (uint32_t)(r - off[c]);
        int b = Nb ? (int)(rr / Nb) : FPL_BINS;
        int id = (b < FPL_BINS && c < C) ? c * FPL_BINS + b : -1;
        if (sub == 0) binmap[pix] = id >= 0 ? (uint8_t)b : (uint8_t)255;
        if (id != cur) {
            if (cur >= 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(&pro_sum[cur * 32 + sub * 4 + k], acc.v[k]);
            }
            acc = f4zero();
            cur = id;
        }
        if (id >= 0) {
            f4 v = ld4(feat + (int64_t)pix * 32 + sub * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc.v[k] += v.v[k];
        }
    }
    if (cur >= 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&pro_sum[cur * 32 + sub * 4 + k], acc.v[k]);
    }
}

// pro = pro_sum / N_c ; loss = sum_c -(1/32) * mean_n(pro_c[n] . buf_c) + mse(pro_last, tgt_last) ; dpro (already / N_c)
__global__ void k_fpl_loss(const float* __restrict__ pro_sum, const uint32_t* __restrict__ counts, const float* __restrict__ buf /*[C][32]*/,
                           int C, float* __restrict__ pro /*[C][32][32]*/, float* __restrict__ loss,
                           float* __restrict__ dpro_over_n /*[C][32][32]*/) {
    __shared__ float sm[16];
    float part = 0.f;
    for (int i = threadIdx.x; i < C * 32 * 32; i += blockDim.x) {
        int c = i / 1024, j = i % 32;
        float Nb = (float)(counts[c] / FPL_BINS);
        float p = pro_sum[i] / Nb;            // Nb == 0 -> NaN/inf exactly like the reference's empty-bin mean
        pro[i] = p;
        float t = buf[c * 32 + j];
        float d = -t * (1.f / 1024.f);
        part += p * d;
        if (c == C - 1) { float e = p - t; part += e * e * (1.f / 1024.f); d += 2.f * e * (1.f / 1024.f); }
        dpro_over_n[i] = d / Nb;
    }
    part = block_sum(part, sm);
    if (threadIdx.x == 0) *loss = part;
}

extern "C" int tcct_fpl_forward(const void* feat, const uint64_t* keys_sorted, const uint32_t* vals_sorted,
                                const uint32_t* counts, int64_t M, int C, const float* buf_grad, float* pro_sum, float* pro,
                                float* loss, float* dpro_over_n, uint8_t* binmap, int dtype, tcct_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    TCCT_CHECK(C >= 1 && C <= FPL_MAXC, "fpl_forward: C=%d unsupported", C);
    if (hipMemsetAsync(pro_sum, 0, sizeof(float) * C * 32 * 32, st) != hipSuccess) { tcct_set_error("fpl_forward: memset failed"); return -2; }
    int64_t groups = (M + FPL_RUN - 1) / FPL_RUN;
    int64_t threads = groups * 8;
    int grid = (int)((threads + FB - 1) / FB);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_fpl_binmean<T>, dim3(grid), dim3(FB), 0, st, (const T*)feat, (const unsigned long long*)keys_sorted, vals_sorted, counts, M, C, pro_sum, binmap));
    hipLaunchKernelGGL(k_fpl_loss, dim3(1), dim3(FB), 0, st, pro_sum, counts, buf_grad, C, pro, loss, dpro_over_n);
    TCCT_LAUNCH_OK();
}

/* the loss part of tcct_fpl_forward alone, for bin sums that came out of tcct_fpl_select */
extern "C" int tcct_fpl_loss(const float* pro_sum, const uint32_t* counts, const float* buf_grad, int C, float* pro, float* loss, float* dpro_over_n,
                             tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= FPL_MAXC, "fpl_loss: C=%d unsupported", C);
    hipLaunchKernelGGL(k_fpl_loss, dim3(1), dim3(FB), 0, (hipStream_t)stream, pro_sum, counts, buf_grad, C, pro, loss, dpro_over_n);
    TCCT_LAUNCH_OK();
}

template <typename T>
__global__ void k_fpl_bwd(const uint8_t* __restrict__ lab, const uint8_t* __restrict__ binmap, const float* __restrict__ dpro_over_n,
                          const float* __restrict__ gout, float gscale, int64_t M, T* __restrict__ dfeat) {
    const float gs = gscale * (gout ? *gout : 1.f);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < M * 8; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = i >> 3;
        int sub = (int)(i & 7);
        int b = binmap[p];
        f4 o = f4zero();
        if (b < FPL_BINS) {
            const float* d = dpro_over_n + ((int)lab[p] * FPL_BINS + b) * 32 + sub * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) o.v[k] = gs * d[k];
        }
        st4(dfeat + i * 4, o);
    }
}
extern "C" int tcct_fpl_backward(const uint8_t* labels, const uint8_t* binmap, const float* dpro_over_n, const float* grad_out,
                                 float grad_scale, int64_t M, void* dfeat, int dtype, tcct_stream_t stream) {
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_fpl_bwd<T>, dim3(tcct_grid(M * 8, FB, 1 << 16)), dim3(FB), 0, (hipStream_t)stream, labels, binmap, dpro_over_n, grad_out, grad_scale, M, (T*)dfeat));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------ optimizer
__global__ void k_sumsq(const float* __restrict__ g, int64_t n, double* __restrict__ acc) {
    __shared__ float sm[16];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) s += g[i] * g[i];
    s = block_sum(s, sm);
    if (threadIdx.x == 0) atomicAdd(acc, (double)s);
}
/* acc[0] = sum g^2 over the flat gradient buffer (acc zeroed by the call) */
extern "C" int tcct_grad_sumsq(const float* g, int64_t n, double* acc, tcct_stream_t stream) {
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, sizeof(double), st) != hipSuccess) { tcct_set_error("grad_sumsq: memset failed"); return -2; }
    hipLaunchKernelGGL(k_sumsq, dim3(tcct_grid(n, FB, 1024)), dim3(FB), 0, st, g, n, acc);
    TCCT_LAUNCH_OK();
}

__global__ void k_clip_adamw(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, const double* __restrict__ sumsq, float max_norm, float gmul, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2, float* __restrict__ total_norm_out) {
    // torch.nn.utils.clip_grad_norm_: coef = min(1, max_norm / (total + 1e-6)); gmul pre-scales the raw gradient (1/world)
    const float total = sqrtf((float)(*sumsq)) * gmul;
    const float coef = fminf(max_norm / (total + 1e-6f), 1.f) * gmul;
    if (total_norm_out && blockIdx.x == 0 && threadIdx.x == 0) *total_norm_out = total;
    const float step = lr / bc1, isq2 = 1.f / sqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        float pi = p[i] * (1.f - lr * wd);
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) * isq2 + eps);
    }
}
// same update with the two per-step scalars in device memory, so that the launch has constant arguments and can be replayed from a
// hipGraph: state[0] = learning rate (written by the host between replays), state[1] = number of steps taken so far (as a float,
// exact up to 2^24; incremented by this kernel).  The increment is done by a second one-thread kernel so that every block of the
// update reads the same value.
__global__ void k_clip_adamw_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 int64_t n, const double* __restrict__ sumsq, float max_norm, float gmul, const float* __restrict__ state,
                                 float b1, float b2, float eps, float wd, float* __restrict__ total_norm_out) {
    const float lr = state[0], t = state[1] + 1.f;
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float total = sqrtf((float)(*sumsq)) * gmul;
    const float coef = fminf(max_norm / (total + 1e-6f), 1.f) * gmul;
    if (total_norm_out && blockIdx.x == 0 && threadIdx.x == 0) *total_norm_out = total;
    const float step = lr / bc1, isq2 = 1.f / sqrtf(bc2);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        float pi = p[i] * (1.f - lr * wd);
        float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi) * isq2 + eps);
    }
}
__global__ void k_step_inc(float* state) { state[1] += 1.f; }
extern "C" int tcct_clip_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm,
                                   float grad_mul, float* state, double beta1, double beta2, float eps, float weight_decay,
                                   float* total_norm_out, tcct_stream_t stream) {
    TCCT_CHECK(state != nullptr, "clip_adamw_dev: state is NULL");
    hipLaunchKernelGGL(k_clip_adamw_dev, dim3(tcct_grid(n, FB, 2048)), dim3(FB), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, max_norm,
                       grad_mul, state, (float)beta1, (float)beta2, eps, weight_decay, total_norm_out);
    hipLaunchKernelGGL(k_step_inc, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_clip_adamw(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm,
                               float grad_mul, float lr, double beta1, double beta2, float eps, float weight_decay, int step,
                               float* total_norm_out, tcct_stream_t stream) {
    TCCT_CHECK(step >= 1, "clip_adamw: step must be >= 1");
    float bc1 = (float)(1.0 - pow(beta1, (double)step)), bc2 = (float)(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(k_clip_adamw, dim3(tcct_grid(n, FB, 2048)), dim3(FB), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, max_norm,
                       grad_mul, lr, (float)beta1, (float)beta2, eps, weight_decay, bc1, bc2, total_norm_out);
    TCCT_LAUNCH_OK();
}
