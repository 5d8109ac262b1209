// Feature-polarization loss (nets/reg.py:86-105, nets/fcs.py:25-50,63-96, nets/fcp.py:72-75) and the fused
// clip_grad_norm_ + AdamW step (kite/loop_seg.py:128-130, kite/loopback.py:127).
//
// FPL pipeline (per step, all classes at once because the label classes partition the pixels):
//   bins    : fpl_select.hip -- radix multi-select of the 32 bin boundaries per class on the key (~prob, pixel index), i.e. the order a stable
//             descending sort by probability gives (nets/fcs.py:25-50), bin map + one-hot MFMA bin sums; NO sort and no library primitive
//             (rounds 1-2 used a rocPRIM radix sort here; round 4 removed it from the library -- the sorted binning lives on as the
//             test-side reference of tests/test_kernels_gpu.py::test_fpl_multiselect_equals_the_sorted_binning)
//   loss    : prototypes = bin sums / N_c, cosine term + MSE of the last class (this file)
//   backward: dfeat[p] = dpro[label[p]][bin[p]] / N_c  (coalesced, bin map written in the forward)
#include "common.h"
#include <cstring>

#define FB 256
#define FPL_BINS 32
#define FPL_MAXC 16     // classes (power of two): 5 for GOALS, 9 for the reference's Duke / HCMS models

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

/* prototypes, loss and d loss / d prototype from the bin sums of tcct_fpl_select */
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
