// First layers of both encoders with their train-mode BatchNorm as ONE store (round 4):
//     z = post(BN_train(conv3x3(x4, w) + bias))      3 -> 32 channels, pad 1, stride 1 (CrossResNet `cnn.0` -> `cnn.1`, reference nets/tcct.py:873)
//                                                     or stride 2 with Hardswish (MPViT `stem[0]` = Conv2d_BN, nets/tcct.py:55-97,674-681).
// The convolution's INPUT is the 4-channel image -- 1/8 of the bytes of its 32-channel output -- so the output y is cheaper to recompute than
// to store and re-read:
//   forward   pass 1 (k_c3_bn_fwd<0>): convolution on the matrix pipes, per-channel sum / sum of squares of y, NO store        reads  56 MB
//             pass 2 (k_c3_bn_fwd<1>): convolution again, a = gamma rstd, b = beta - mean a from the batch sums in the prologue,
//                                      z = post(a y + b) stored                                                                 reads  56, writes 452 MB
//             (before: conv + statistics writes y 452 MB, the normalisation pass reads 452 and writes 452 MB)
//   backward  pass 1 (k_c3_bn_bwd<0>): y recomputed per 128-pixel tile, dz' = dz post'(a y + b), sums of dz' and dz' y         reads 452 + 56 MB
//             pass 2 (k_c3_bn_bwd<1>): y recomputed, dy = c1 dz' + c2 y + c3 (per-channel constants from the two sums,
//                                      tcct_bn_bwd_coef) built IN REGISTERS in the MFMA operand layout, dW += dy^T patch, db    reads 452 + 56 MB
//             (before: reduction reads dz + y 904 MB, apply reads dz + y and writes dy 1 356 MB, weight gradient reads dy 452 MB)
// At the bench shape (bs 8, 800 x 1104) the CNN's first layer moves 1.5 GB instead of 4.1 GB per step and the 452 MB tensor y does not exist.
// y is never rounded: statistics and the normalisation see the fp32 accumulators (one rounding point, z, instead of two).
//
// Operand orientation of the backward kernel: D = patch . W^T (rows = pixels, columns = output channels), so a lane owns ONE channel
// (per-channel constants are lane scalars) and 16 pixels; two v_permlane32_swap per register pair turn that into the A-fragment layout
// (8 consecutive pixels per lane half) in which ds_read_b64_tr_b16 delivers dz and in which the weight-gradient MFMA wants dy^T: the rebuilt
// dy never touches LDS or HBM.
#include "common.h"
#include "c3_geom.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define C3B 256
#define C3_SW 112           // LDS row stride of the weights: 48 bf16 + 16 B (conflict-free ds_read_b128 over 32 rows)

// weights -> LDS as bf16 [32][k], k = 16*ky + 4*kx + ch (kx = 3 / ch = 3: zero), from w fp32 [32][3][3][3]
__device__ __forceinline__ void c3_stage_weights(unsigned char* sW, const float* __restrict__ w, int tid) {
    for (int i = tid; i < 32 * 48; i += C3B) {
        const int row = i / 48, k = i - row * 48, ky = k >> 4, kx = (k >> 2) & 3, ch = k & 3;
        *reinterpret_cast<bf16*>(sW + row * C3_SW + k * 2) = __float2bfloat16((kx < 3 && ch < 3) ? w[row * 27 + ch * 9 + ky * 3 + kx] : 0.f);
    }
}

template <int POST> __device__ __forceinline__ float post_grad(float t) {
    if (POST == TCCT_ACT_HSWISH) return t < -3.f ? 0.f : (t <= 3.f ? (2.f * t + 3.f) * (1.f / 6.f) : 1.f);
    if (POST == TCCT_ACT_LRELU) return t > 0.f ? 1.f : 0.01f;
    return 1.f;
}

struct C3BnTrain {      // what pass 2 of the forward needs to finalise the statistics (k_bn_apply's BnTrain)
    const double* sums; const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; int64_t* nbt; float* mean_rstd; float* ab_out;
};

// ------------------------------------------------------------------------------------------------------------------ forward
// MODE 0: statistics of y = conv + bias (sum, sum of squares -> stats fp64 [64], zero on entry), nothing stored
// MODE 1: z = post(a y + b) with a, b from the batch sums (every block derives them in its prologue; block 0 publishes mean / rstd / a / b and
//         moves the running statistics)
// Tile = 32 consecutive pixels of ONE output row per wave, six branch-free 8-byte gathers per lane issued two tiles ahead (k_pw_fwd<.., C3>).
template <int MODE, int POST>
__global__ void __launch_bounds__(C3B, 2)
k_c3_bn_fwd(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, bf16* __restrict__ z, int64_t M,
            double* __restrict__ stats, C3BnTrain tr, C3Geom g3) {
    __shared__ __attribute__((aligned(16))) unsigned char sW[32 * C3_SW];
    __shared__ __attribute__((aligned(16))) unsigned char sScr[4 * 2560];          // per-wave epilogue transpose (32 pixels x 80 B); MODE 0: reduction slots
    __shared__ float sBias[32], sA[32], sBb[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    c3_stage_weights(sW, w, tid);
    if (tid < 32) {
        sBias[tid] = bias ? bias[tid] : 0.f;
        if (MODE == 1) {
            const double mean = tr.sums[tid] / (double)M;
            double var = tr.sums[32 + tid] / (double)M - mean * mean;
            if (var < 0.0) var = 0.0;
            const float rstd = (float)(1.0 / sqrt(var + (double)tr.eps));
            const float a = tr.gamma[tid] * rstd, b = tr.beta[tid] - (float)mean * a;
            sA[tid] = a; sBb[tid] = b;
            if (blockIdx.x == 0) {
                tr.mean_rstd[tid] = (float)mean; tr.mean_rstd[32 + tid] = rstd; tr.ab_out[tid] = a; tr.ab_out[32 + tid] = b;
                if (tr.running_mean) {
                    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
                    tr.running_mean[tid] = (1.f - tr.momentum) * tr.running_mean[tid] + tr.momentum * (float)mean;
                    tr.running_var[tid] = (1.f - tr.momentum) * tr.running_var[tid] + tr.momentum * (float)unb;
                }
                if (tid == 0 && tr.nbt) *tr.nbt += 1;
            }
        } else if (MODE == 2) {             // eval mode: a, b from the running statistics (tcct_bn_eval_ab), handed over in tr.ab_out
            sA[tid] = tr.ab_out[tid]; sBb[tid] = tr.ab_out[32 + tid];
        }
    }
    __syncthreads();
    // the lane's 16 output channels are 8q + 4hh + k: bias (and a, b) of those channels in registers
    float bz[16], ca[MODE >= 1 ? 16 : 1], cb[MODE >= 1 ? 16 : 1];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bz[4 * q + k] = sBias[8 * q + 4 * hh + k];
            if (MODE >= 1) { ca[4 * q + k] = sA[8 * q + 4 * hh + k]; cb[4 * q + k] = sBb[8 * q + 4 * hh + k]; }
        }
    float ss[MODE == 0 ? 16 : 1], sq[MODE == 0 ? 16 : 1];
    if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) ss[i] = sq[i] = 0.f;
    }
    const int64_t mtiles = (M / g3.Wo) * g3.tpr;            // (B * Ho output rows) x (tiles per row)
    const __amdgpu_buffer_rsrc_t c3r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g3.bytes, 0x00020000);
    uint32_t c3q[12], c3p[12];
    auto c3_tile = [&](int64_t mt_, uint32_t& row, uint32_t& xb) {
        row = udiv_m((uint32_t)mt_, (uint32_t)g3.tpr, g3.m_tpr);
        xb = (uint32_t)mt_ - row * (uint32_t)g3.tpr;
    };
    auto c3_load = [&](uint32_t (&cq)[12], int64_t mt_) {       // no branches: a tile beyond the last one loads zeros
        const bool live = mt_ < mtiles;
        uint32_t row, xb;
        c3_tile(live ? mt_ : 0, row, xb);
        const uint32_t n_ = udiv_m(row, (uint32_t)g3.Ho, g3.m_ho), oy = row - n_ * (uint32_t)g3.Ho;
        const int ox = (int)xb * 32 + r;
        const int iy0 = (int)oy * g3.stride - 1, ix0 = ox * g3.stride - 1;
        const bool in = live & (ox < g3.Wo);
        const int pa = ix0 + 2 * hh, pb = ix0 + 1;
        const bool va = in & ((unsigned)pa < (unsigned)g3.W), vb = in & (hh == 0) & ((unsigned)pb < (unsigned)g3.W);
        const int base = ((int)(n_ * (uint32_t)g3.H) + iy0) * g3.W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool rv = (unsigned)(iy0 + ky) < (unsigned)g3.H;
            const uint32_t offa = (va & rv) ? (uint32_t)(base + ky * g3.W + pa) * 8u : 0x80000000u;
            const uint32_t offb = (vb & rv) ? (uint32_t)(base + ky * g3.W + pb) * 8u : 0x80000000u;
            const uint2 ta = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(c3r, offa, 0, 0));
            const uint2 tb = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(c3r, offb, 0, 0));
            cq[4 * ky] = ta.x; cq[4 * ky + 1] = ta.y; cq[4 * ky + 2] = tb.x; cq[4 * ky + 3] = tb.y;
        }
    };
    const int64_t mstep = (int64_t)gridDim.x * 4;
    auto do_tile = [&](int64_t mt, uint32_t (&cq)[12]) {
        uint32_t row, xb;
        c3_tile(mt < mtiles ? mt : 0, row, xb);
        const int64_t mbase = (int64_t)row * g3.Wo + xb * 32;
        const int64_t mlim = mt < mtiles ? (int64_t)(row + 1) * g3.Wo : 0;     // the unrolled loop may run one tile past the end: nothing valid in it
        f32x16 acc;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = bz[k];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bf16x8 bv = __builtin_bit_cast(bf16x8, make_uint4(cq[4 * i], cq[4 * i + 1], cq[4 * i + 2], cq[4 * i + 3]));
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(sW + r * C3_SW + (16 * i + 8 * hh) * 2);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        c3_load(cq, mt + 2 * mstep);            // two tiles ahead, into the registers the MFMAs have just read
        if (MODE == 0) {
            const float okf = (mbase + r < mlim) ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { const float v = acc[i] * okf; ss[i] += v; sq[i] += v * v; }
        } else {
            unsigned char* sc = sScr + wave * 2560;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = act_c<POST>(ca[4 * q + k] * acc[4 * q + k] + cb[4 * q + k]);
                uint2 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
                *reinterpret_cast<uint2*>(sc + r * 80 + (8 * q + 4 * hh) * 2) = o;
            }
            wave_lds_fence();
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int p = (lane >> 2) + 16 * h2, cch = lane & 3;
                const uint4 o = *reinterpret_cast<const uint4*>(sc + p * 80 + cch * 16);
                const int64_t mm = mbase + p;
                if (mm < mlim) *reinterpret_cast<uint4*>(z + mm * 32 + cch * 8) = o;
            }
            wave_lds_fence();
        }
    };
    const int64_t mt0 = (int64_t)blockIdx.x * 4 + wave;
    c3_load(c3q, mt0);
    __builtin_amdgcn_sched_barrier(0);
    c3_load(c3p, mt0 + mstep);
    __builtin_amdgcn_sched_barrier(0);
    for (int64_t mt = mt0; mt < mtiles; mt += 2 * mstep) {
        do_tile(mt, c3q);
        do_tile(mt + mstep, c3p);
    }
    if (MODE == 0) {
        // lanes with equal hh hold different pixels of the same 16 channels: sum over the 32 lanes of a half, then per-wave slots (no LDS atomics)
        float* red = reinterpret_cast<float*>(sScr);           // [4 waves][2][32]
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float a = lane_group_sum(ss[i], 32), b = lane_group_sum(sq[i], 32);
            if (r == 0) {
                const int c = 8 * (i >> 2) + 4 * hh + (i & 3);
                red[wave * 64 + c] = a; red[wave * 64 + 32 + c] = b;
            }
        }
        __syncthreads();
        if (tid < 64) {
            double a = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) a += (double)red[wv * 64 + tid];
            atomicAdd(&stats[tid], a);
        }
    }
}

static C3Geom c3_geom(int B, int H, int W, int stride) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    auto magic = [](uint32_t d) { return d == 1 ? 0xffffffffu : (uint32_t)((1ull << 32) / d); };
    const int tpr = (Wo + 31) / 32;
    return C3Geom{H, W, Ho, Wo, stride, (uint32_t)((int64_t)B * H * W * 8), tpr, magic((uint32_t)tpr), magic((uint32_t)Ho), magic((uint32_t)(Ho * Wo)),
                  magic((uint32_t)Wo)};
}

/* z bf16 [B,Ho,Wo,32] = post_act(BatchNorm_train(conv3x3(x4[..., :3], w, pad 1, stride) + bias)) in two launches, the convolution output never
 * stored (see the header of this file).  x4 bf16 [B,H,W,4]; w fp32 [32,3,3,3]; bias nullable; sums fp64 [64] (zero on entry; afterwards the
 * batch sums of y); gamma / beta / running_* / num_batches_tracked as tcct_bn_apply_train; mean_rstd [64], ab [64] written for the backward;
 * post_act: TCCT_ACT_NONE (cnn.1, nets/tcct.py:873) or TCCT_ACT_HSWISH (stem[0], :80-97). */
extern "C" int tcct_c3_bn_fwd_train(const void* x4, const float* w, const float* bias, void* z, int B, int H, int W, int stride, double* sums,
                                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                                    int64_t* num_batches_tracked, float* mean_rstd, float* ab, int post_act, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "c3_bn_fwd_train: stride %d", stride);
    TCCT_CHECK(post_act == TCCT_ACT_NONE || post_act == TCCT_ACT_HSWISH, "c3_bn_fwd_train: post_act %d (none or hswish)", post_act);
    TCCT_CHECK(sums && gamma && beta && mean_rstd && ab, "c3_bn_fwd_train: NULL argument");
    const C3Geom g = c3_geom(B, H, W, stride);
    const int64_t M = (int64_t)B * g.Ho * g.Wo, inb = (int64_t)B * H * W * 8;
    TCCT_CHECK(M > 0 && M * 64 < (1ll << 31) && inb < (1ll << 31), "c3_bn_fwd_train: image too large for 32-bit byte offsets (B=%d H=%d W=%d)", B, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(sums, 0, sizeof(double) * 64, st) != hipSuccess) { tcct_set_error("c3_bn_fwd_train: memset failed"); return -2; }
    const int64_t mtiles = (int64_t)B * g.Ho * g.tpr;
    int64_t gx = (mtiles + 3) / 4;
    if (gx > 256 * 4) gx = 256 * 4;
    // statistics pass: a gather-latency-bound loop (no stores to overlap with): four blocks per CU; the 64 fp64 atomics per block are noise at 1024 blocks
    // (512 blocks: 0.077 ms at the bench shape for a 56 MB read)
    int64_t gs = gx;
    C3BnTrain tr{sums, gamma, beta, eps, momentum, running_mean, running_var, num_batches_tracked, mean_rstd, ab};
    hipLaunchKernelGGL((k_c3_bn_fwd<0, 0>), dim3((unsigned)gs), dim3(C3B), 0, st, (const bf16*)x4, w, bias, (bf16*)nullptr, M, sums, tr, g);
    if (post_act == TCCT_ACT_HSWISH)
        hipLaunchKernelGGL((k_c3_bn_fwd<1, TCCT_ACT_HSWISH>), dim3((unsigned)gx), dim3(C3B), 0, st, (const bf16*)x4, w, bias, (bf16*)z, M, sums, tr, g);
    else
        hipLaunchKernelGGL((k_c3_bn_fwd<1, TCCT_ACT_NONE>), dim3((unsigned)gx), dim3(C3B), 0, st, (const bf16*)x4, w, bias, (bf16*)z, M, sums, tr, g);
    TCCT_LAUNCH_OK();
}

/* inference (KiteSeg.predict / val, reference kite/loop_seg.py:21-33,66-106): the same one-store form with the EVAL-mode BatchNorm -- z = post_act(a y + b), ab fp32 [64]
 * = {a[32], b[32]} from the running statistics (tcct_bn_eval_ab); one launch (the normalising pass of tcct_c3_bn_fwd_train), y never stored. */
extern "C" int tcct_c3_bn_fwd_eval(const void* x4, const float* w, const float* bias, void* z, int B, int H, int W, int stride, const float* ab, int post_act,
                                   tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "c3_bn_fwd_eval: stride %d", stride);
    TCCT_CHECK(post_act == TCCT_ACT_NONE || post_act == TCCT_ACT_HSWISH, "c3_bn_fwd_eval: post_act %d (none or hswish)", post_act);
    TCCT_CHECK(x4 && w && z && ab, "c3_bn_fwd_eval: NULL argument");
    const C3Geom g = c3_geom(B, H, W, stride);
    const int64_t M = (int64_t)B * g.Ho * g.Wo, inb = (int64_t)B * H * W * 8;
    TCCT_CHECK(M > 0 && M * 64 < (1ll << 31) && inb < (1ll << 31), "c3_bn_fwd_eval: image too large for 32-bit byte offsets (B=%d H=%d W=%d)", B, H, W);
    const int64_t mtiles = (int64_t)B * g.Ho * g.tpr;
    int64_t gx = (mtiles + 3) / 4;
    if (gx > 256 * 4) gx = 256 * 4;
    C3BnTrain tr{nullptr, nullptr, nullptr, 0.f, 0.f, nullptr, nullptr, nullptr, nullptr, const_cast<float*>(ab)};
    hipStream_t st = (hipStream_t)stream;
    if (post_act == TCCT_ACT_HSWISH)
        hipLaunchKernelGGL((k_c3_bn_fwd<2, TCCT_ACT_HSWISH>), dim3((unsigned)gx), dim3(C3B), 0, st, (const bf16*)x4, w, bias, (bf16*)z, M, (double*)nullptr, tr, g);
    else
        hipLaunchKernelGGL((k_c3_bn_fwd<2, TCCT_ACT_NONE>), dim3((unsigned)gx), dim3(C3B), 0, st, (const bf16*)x4, w, bias, (bf16*)z, M, (double*)nullptr, tr, g);
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------------------------ backward
#define C3_P 128            // pixels per staged tile (eight 16-pixel chunks: two per wave)
#define C3_SX 192           // patch rows: 48 (-> 64, zero padded) bf16 + 64 B: stride mod 256 = 192 keeps transposing reads on distinct banks
#define C3_SD 64            // dz rows
// MODE 0: red fp64 [64] += {sum dz', sum dz' y} (raw form, see tcct_bn_sums_from_raw / tcct_bn_bwd_coef(raw = 1))
// MODE 1: dw fp32 [32,3,3,3] += dy^T patch, dbias += sum dy with dy = c1 dz' + c2 y + c3 (coef = {c1[32], c2[32], c3[32], a[32], b[32]})
// MODE 2 (round 4): BOTH in one pass over dz.  dy = a (dz' - s1 - yh s2) is LINEAR in the per-pixel quantities (s1 = mean dz', s2 = mean dz' yh, yh = (y - mean) rstd
//         known from the forward), so dW = a (A1 - s2 A2 - s1 A3) with A1 = sum dz' patch, A2 = sum yh patch, A3 = sum patch: the kernel accumulates the three
//         sums (dw = workspace fp32: A1 [32][64], A2 [32][64], A3 [64]) and red fp64 [96] += {sum dz', sum dz' yh, sum yh}; k_c3_bn_bwd_fin combines them.  coef = mean_rstd [64].
// PF: tiles requested ahead (register rings px / pd, the tile loop unrolled PF times).  Round 6: with ONE tile ahead and the two blocks per CU that MODE 2's ~190 VGPRs
// allow, 28 KB per CU were in flight -- 2.2 TB/s at ~2.5 us of loaded HBM latency (Little's law), 0.276 ms for the CNN's first layer; MODE 2 runs with PF = 3.
template <int MODE, int POST, int PF = 1>
__global__ void __launch_bounds__(C3B, 2)
k_c3_bn_bwd(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, const bf16* __restrict__ dz, int64_t M,
            const float* __restrict__ ab, const float* __restrict__ coef, double* __restrict__ red, float* __restrict__ dw, float* __restrict__ dbias,
            C3Geom g3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sD = smem + C3_P * C3_SX;
    unsigned char* sW = sD + C3_P * C3_SD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    c3_stage_weights(sW, w, tid);
    for (int i = tid; i < C3_P * 2; i += C3B) *reinterpret_cast<uint4*>(sX + (i >> 1) * C3_SX + (6 + (i & 1)) * 16) = make_uint4(0, 0, 0, 0);  // patch elements 48..63
    // lane constants of output channel r
    const float bias_r = bias ? bias[r] : 0.f;
    const float a_r = ab[r], b_r = ab[32 + r];
    const float c1 = MODE == 1 ? coef[r] : 0.f, c2 = MODE == 1 ? coef[32 + r] : 0.f, c3 = MODE == 1 ? coef[64 + r] : 0.f;
    const float mu_r = MODE == 2 ? coef[r] : 0.f, rs_r = MODE == 2 ? coef[32 + r] : 0.f;
    __syncthreads();
    bf16x8 wf[3];           // the lane's weight fragments (B operand: k = 16 i + 8 hh .. + 7 of column co = r), resident
#pragma unroll
    for (int i = 0; i < 3; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sW + r * C3_SW + (16 * i + 8 * hh) * 2);
    f32x16 acc[2], acc2[MODE == 2 ? 2 : 1];
    float a3[2] = {0.f, 0.f};       // MODE 2: sum over pixels of patch element 32 kt + r (this lane half's pixels)
    if (MODE >= 1) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc[b][k] = 0.f; if (MODE == 2) acc2[b][k] = 0.f; }
    }
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;       // MODE 0: sum dz', sum dz' y of channel r;  MODE 1: s1 = sum dy (bias gradient);  MODE 2: sum dz', sum dz' yh, sum yh
    const __amdgpu_buffer_rsrc_t c3r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g3.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t c3d = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (uint32_t)(M * 64), 0x00020000);
    uint4 pxr[PF][3], pdr[PF][2];
    auto prefetch = [&](int64_t tile, uint4 (&px)[3], uint4 (&pd)[2]) {     // thread = (pixel tid % 128, half tid / 128): input pixels (ky, kx = 0,1) or (ky, kx = 2) + zeros; branch-free
        const int64_t m0 = tile * C3_P;
        const uint32_t mm = (uint32_t)(m0 + (tid & 127));
        const bool in = m0 + (tid & 127) < M;
        const uint32_t n_ = udiv_m(mm, (uint32_t)(g3.Ho * g3.Wo), g3.m_howo), rem = mm - n_ * (uint32_t)(g3.Ho * g3.Wo);
        const uint32_t oy = udiv_m(rem, (uint32_t)g3.Wo, g3.m_wo), ox = rem - oy * (uint32_t)g3.Wo;
        const int iy0 = (int)oy * g3.stride - 1, ix0 = (int)ox * g3.stride - 1;
        const int hi = tid >> 7;
        const int pa = ix0 + 2 * hi, pb = ix0 + 1;
        const bool va = in & ((unsigned)pa < (unsigned)g3.W), vb = in & (hi == 0) & ((unsigned)pb < (unsigned)g3.W);
        const int base = ((int)(n_ * (uint32_t)g3.H) + iy0) * g3.W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool rv = (unsigned)(iy0 + ky) < (unsigned)g3.H;
            const uint32_t offa = (va & rv) ? (uint32_t)(base + ky * g3.W + pa) * 8u : 0x80000000u;
            const uint32_t offb = (vb & rv) ? (uint32_t)(base + ky * g3.W + pb) * 8u : 0x80000000u;
            const uint2 ta = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(c3r, offa, 0, 0));
            const uint2 tb = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(c3r, offb, 0, 0));
            px[ky] = make_uint4(ta.x, ta.y, tb.x, tb.y);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)         // the 128 x 64 B tile of dz is one contiguous span; rows >= M read as zeros through the descriptor
            pd[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(c3d, (uint32_t)m0 * 64u + (uint32_t)(tid + j * C3B) * 16u, 0, 0));
    };
    // transposing-read bases (k_pw_wgrad): address = base + chunk * 16 rows (+ 4 rows for the second half)
    const int li = lane & 15, lq = li >> 2, lpp = li & 3, lg = lane >> 4;
    const int lrow = 8 * (lg >> 1) + lq, lcol = (16 * (lg & 1) + 4 * lpp) * 2;
    const unsigned char* lbX = sX + lrow * C3_SX + lcol;
    const unsigned char* lbD = sD + lrow * C3_SD + lcol;
    auto tr2 = [&](const unsigned char* p, int stride) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * stride));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    // the 32 rows of this wave's recompute GEMM: row i < 16 = pixel 16 wave + i (chunk `wave`), row i >= 16 = pixel 16 (wave + 4) + i - 16
    const int prow = r < 16 ? 16 * wave + r : 16 * (wave + 4) + (r - 16);
    const int64_t tiles = (M + C3_P - 1) / C3_P;
    int64_t tile0 = blockIdx.x;
#pragma unroll
    for (int u = 0; u < PF; ++u) prefetch(tile0 + (int64_t)u * gridDim.x, pxr[u], pdr[u]);       // branch-free loads: a tile past the end reads zeros
    for (; tile0 < tiles; tile0 += (int64_t)PF * gridDim.x) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int64_t tile = tile0 + (int64_t)u * gridDim.x;
        if (tile >= tiles) break;           // block-uniform
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<uint4*>(sX + (tid & 127) * C3_SX + (2 * j + (tid >> 7)) * 16) = pxr[u][j];
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int i = tid + j * C3B; *reinterpret_cast<uint4*>(sD + (i >> 2) * C3_SD + (i & 3) * 16) = pdr[u][j]; }
        __syncthreads();
        prefetch(tile + (int64_t)PF * gridDim.x, pxr[u], pdr[u]);
        // ---- y of the wave's 32 pixels: D[pixel][co] = patch . W^T + bias: lane = channel r, rows 8q + 4hh + k
        f32x16 y;
#pragma unroll
        for (int k = 0; k < 16; ++k) y[k] = bias_r;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bf16x8 pv = *reinterpret_cast<const bf16x8*>(sX + prow * C3_SX + (16 * i + 8 * hh) * 2);
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pv, wf[i], y, 0, 0, 0);
        }
        // rows 8q + 4hh + k -> A-fragment layout (chunk c = q / 2: rows 16c + 8hh + j): lower lanes give their q odd rows (8..11 / 24..27) and take
        // the upper lanes' q even rows (4..7 / 20..23)
        float yf[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // (scalars first: hipcc 7.2 compiles `__builtin_bit_cast(int, vector[i])` and `__builtin_bit_cast(float, sw[1])` to a load of
                // ELEMENT 0 of the vector -- the index is computed and dropped; found with tools/dbg_c3bn2.py, every row read row 0's value)
                const float ylo = y[8 * c + k], yhi = y[8 * c + 4 + k];
                // v_permlane32_swap: lo[lanes 32..63] <-> hi[lanes 0..31]
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ylo), __float_as_uint(yhi), false, false);
                const uint32_t s0 = sw[0], s1_ = sw[1];
                yf[c][k] = __uint_as_float(s0);
                yf[c][4 + k] = __uint_as_float(s1_);
            }
        const int64_t m0 = tile * C3_P;
        const bool tail = m0 + C3_P > M;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = wave + 4 * c;            // 16-pixel chunk of the tile
            const bf16x8 dzf = tr2(lbD + ch * 16 * C3_SD, C3_SD);          // pixels 16 ch + 8 hh + j of channel r
            const s16x8 dzs = __builtin_bit_cast(s16x8, dzf);
            float dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = __uint_as_float(((uint32_t)(uint16_t)dzs[j]) << 16);
                dv[j] = POST == TCCT_ACT_NONE ? d : d * post_grad<POST>(a_r * yf[c][j] + b_r);
            }
            if (MODE == 0) {
                // pixels beyond M were staged as zeros in dz: they add nothing
#pragma unroll
                for (int j = 0; j < 8; ++j) { s1 += dv[j]; s2 += dv[j] * yf[c][j]; }
            } else if (MODE == 2) {
                const int64_t p0 = m0 + 16 * ch + 8 * hh;
                uint32_t pk[4], ph[4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    // (pixels beyond M only exist in the last tile: a block-uniform flag keeps the 16 64-bit compares out of every other tile)
                    const float h0 = (!tail || p0 + j < M) ? (yf[c][j] - mu_r) * rs_r : 0.f, h1 = (!tail || p0 + j + 1 < M) ? (yf[c][j + 1] - mu_r) * rs_r : 0.f;
                    s1 += dv[j] + dv[j + 1]; s2 += dv[j] * h0 + dv[j + 1] * h1; s3 += h0 + h1;
                    if (POST != TCCT_ACT_NONE) pk[j >> 1] = pack_bf16x2(dv[j], dv[j + 1]);
                    ph[j >> 1] = pack_bf16x2(h0, h1);
                }
                const bf16x8 dzp = POST == TCCT_ACT_NONE ? dzf : __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));      // dz' = dz: the fragment as read
                const bf16x8 yhp = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const bf16x8 xf = tr2(lbX + ch * 16 * C3_SX + 64 * kt, C3_SX);      // (patch rows of pixels beyond M are zeros: nothing to mask in the products)
                    acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dzp, xf, acc[kt], 0, 0, 0);
                    acc2[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yhp, xf, acc2[kt], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) a3[kt] += (float)xf[j];
                }
            } else {
                const int64_t p0 = m0 + 16 * ch + 8 * hh;
                uint32_t pk[4];
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    float d0 = c1 * dv[j] + c2 * yf[c][j] + c3, d1 = c1 * dv[j + 1] + c2 * yf[c][j + 1] + c3;
                    if (p0 + j >= M) d0 = 0.f;          // c3 is not zero: pixels beyond the tensor must not contribute
                    if (p0 + j + 1 >= M) d1 = 0.f;
                    pk[j >> 1] = pack_bf16x2(d0, d1);   // rounded to bf16 like the tensor the separate apply pass used to write
                    s1 += __uint_as_float(pk[j >> 1] << 16) + __uint_as_float(pk[j >> 1] & 0xffff0000u);
                }
                const bf16x8 dyf = __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const bf16x8 xf = tr2(lbX + ch * 16 * C3_SX + 64 * kt, C3_SX);
                    acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dyf, xf, acc[kt], 0, 0, 0);
                }
            }
        }
    }
    }
    __syncthreads();
    float* redf = reinterpret_cast<float*>(smem);
    if (MODE == 0) {
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (lane < 32) { redf[wave * 64 + r] = s1; redf[wave * 64 + 32 + r] = s2; }
        __syncthreads();
        if (tid < 64) {
            double a = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) a += (double)redf[wv * 64 + tid];
            atomicAdd(&red[tid], a);
        }
        return;
    }
    if (MODE == 2) {
        // two [32 co][64 k] matrices + the 64 patch sums: the four waves take turns adding theirs into LDS, then one fp32 atomic per element and block into the workspace
        for (int turn = 0; turn < 4; ++turn) {
            if (wave == turn) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int co = (k & 3) + 8 * (k >> 2) + 4 * hh, o = co * 64 + kt * 32 + r;
                        redf[o] = turn == 0 ? acc[kt][k] : redf[o] + acc[kt][k];
                        redf[2048 + o] = turn == 0 ? acc2[kt][k] : redf[2048 + o] + acc2[kt][k];
                    }
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const float v = a3[kt] + __shfl_xor(a3[kt], 32, 64);
                    if (lane < 32) redf[4096 + kt * 32 + r] = turn == 0 ? v : redf[4096 + kt * 32 + r] + v;
                }
            }
            __syncthreads();
        }
        // only the 27 patch elements that are weights (k < 48, kx < 3, ch < 3) are read by k_c3_bn_bwd_fin: 1 755 atomics per block instead of 4 160 (round 6: with 512 blocks
        // the closing atomics were ~18 us of the launch at either first layer)
        for (int i = tid; i < 2 * 2048 + 64; i += C3B) {
            const int cl = i & 63;
            if (cl < 48 && ((cl >> 2) & 3) < 3 && (cl & 3) < 3) atomicAdd(&dw[i], redf[i]);
        }
        __syncthreads();
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); s3 += __shfl_xor(s3, 32, 64);
        if (lane < 32) { redf[wave * 96 + r] = s1; redf[wave * 96 + 32 + r] = s2; redf[wave * 96 + 64 + r] = s3; }
        __syncthreads();
        if (tid < 96) {
            double a = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) a += (double)redf[wv * 96 + tid];
            atomicAdd(&red[tid], a);
        }
        return;
    }
    // the four waves hold partial sums of the same [32 co][64 k] tile: they take turns adding them into LDS (no LDS float atomics)
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int co = (k & 3) + 8 * (k >> 2) + 4 * hh;
                    float* dst = &redf[co * 64 + kt * 32 + r];
                    *dst = turn == 0 ? acc[kt][k] : *dst + acc[kt][k];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < 32 * 64; i += C3B) {
        const int co = i >> 6, cl = i & 63;         // patch element 16*ky + 4*kx + ch -> w[co][ch][ky][kx]
        if (cl < 48 && ((cl >> 2) & 3) < 3 && (cl & 3) < 3) atomicAdd(&dw[co * 27 + (cl & 3) * 9 + (cl >> 4) * 3 + ((cl >> 2) & 3)], redf[i]);
    }
    if (dbias) {
        __syncthreads();
        s1 += __shfl_xor(s1, 32, 64);
        if (lane < 32) redf[wave * 32 + r] = s1;
        __syncthreads();
        if (tid < 32) atomicAdd(&dbias[tid], redf[tid] + redf[32 + tid] + redf[64 + tid] + redf[96 + tid]);
    }
}


// ---- MODE 2 with WAVE-PRIVATE tiles (round 6).  k_c3_bn_bwd<2> stages 128 pixels per block and pays two block barriers per tile with two blocks per CU resident: its
// eight waves per CU move in two lockstep groups and the VALU (286 instructions per tile and wave) idles about half of the time (profiles/r06_c3bwd_prefetch.txt).  Here a
// wave owns 32 consecutive pixels per iteration -- its own 6 KB patch tile and 2 KB of dz in LDS, ordered by the wave's in-order LDS queue, no barrier in the loop -- and the
// VALU work is cut: the patch sums A3 come out of a third MFMA with a fragment of ones (the matrix pipes were 30 % busy) instead of 64 conversions + adds per tile, the
// masks of pixels beyond M exist in the last tile only (a wave-uniform branch).  Same sums, same fragment layouts and MFMA order per pixel as the block-tile kernel; the order
// in which tiles are added differs (fp32 accumulators: ~1e-6 relative, as between two grid sizes of the old kernel).
template <int POST>
__global__ void __launch_bounds__(C3B, 2)
k_c3_bn_bwd_wave(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, const bf16* __restrict__ dz, int64_t M,
                 const float* __restrict__ ab, const float* __restrict__ coef, double* __restrict__ red, float* __restrict__ dw, C3Geom g3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    unsigned char* sW = smem + 4 * 32 * (C3_SX + C3_SD);
    unsigned char* sX = smem + wave * 32 * (C3_SX + C3_SD);
    unsigned char* sD = sX + 32 * C3_SX;
    c3_stage_weights(sW, w, tid);
    // patch elements 48..63 of the wave's 32 rows: zero once
    *reinterpret_cast<uint4*>(sX + r * C3_SX + (6 + hh) * 16) = make_uint4(0, 0, 0, 0);
    // the bias never enters the loop: y' = patch . W^T (accumulators start at the inline constant 0) and y = y' + bias is folded into the lane constants
    const float bias_r = bias ? bias[r] : 0.f;
    const float a_r = ab[r], b_r = ab[32 + r] + ab[r] * bias_r;
    const float rs_r = coef[32 + r], murs_r = (coef[r] - bias_r) * rs_r;
    __syncthreads();
    bf16x8 wf[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(sW + r * C3_SW + (16 * i + 8 * hh) * 2);
    bf16x8 ones;
#pragma unroll
    for (int k = 0; k < 8; ++k) ones[k] = (__bf16)1.0f;
    f32x16 acc[2], acc2[2], acc3[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc[b][k] = 0.f; acc2[b][k] = 0.f; acc3[b][k] = 0.f; }
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const __amdgpu_buffer_rsrc_t c3r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, g3.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t c3d = __builtin_amdgcn_make_buffer_rsrc((void*)dz, 0, (uint32_t)(M * 64), 0x00020000);
    uint4 px[3], pd[2];
    auto prefetch = [&](int64_t tile) {         // lane = (pixel lane % 32, half lane / 32): input pixels (ky, kx = 0, 1) or (ky, kx = 2) + zeros; branch-free
        const int64_t m0 = tile * 32;
        const uint32_t mm = (uint32_t)(m0 + r);
        const bool in = m0 + r < M;
        const uint32_t n_ = udiv_m(mm, (uint32_t)(g3.Ho * g3.Wo), g3.m_howo), rem = mm - n_ * (uint32_t)(g3.Ho * g3.Wo);
        const uint32_t oy = udiv_m(rem, (uint32_t)g3.Wo, g3.m_wo), ox = rem - oy * (uint32_t)g3.Wo;
        const int iy0 = (int)oy * g3.stride - 1, ix0 = (int)ox * g3.stride - 1;
        const int pa = ix0 + 2 * hh, pb = ix0 + 1;
        const bool va = in & ((unsigned)pa < (unsigned)g3.W), vb = in & (hh == 0) & ((unsigned)pb < (unsigned)g3.W);
        const int base = ((int)(n_ * (uint32_t)g3.H) + iy0) * g3.W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool rv = (unsigned)(iy0 + ky) < (unsigned)g3.H;
            const uint32_t offa = (va & rv) ? (uint32_t)(base + ky * g3.W + pa) * 8u : 0x80000000u;
            const uint32_t offb = (vb & rv) ? (uint32_t)(base + ky * g3.W + pb) * 8u : 0x80000000u;
            const uint2 ta = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(c3r, offa, 0, 0));
            const uint2 tb = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(c3r, offb, 0, 0));
            px[ky] = make_uint4(ta.x, ta.y, tb.x, tb.y);
        }
        const bool live = m0 < M;               // (a tile past the end: offsets may wrap, read nothing)
#pragma unroll
        for (int j = 0; j < 2; ++j)             // the 32 x 64 B tile of dz is one contiguous span; rows >= M read as zeros through the descriptor
            pd[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(c3d, live ? (uint32_t)m0 * 64u + (uint32_t)(lane + j * 64) * 16u : 0x80000000u, 0, 0));
    };
    const int li = lane & 15, lq = li >> 2, lpp = li & 3, lg = lane >> 4;
    const int lrow = 8 * (lg >> 1) + lq, lcol = (16 * (lg & 1) + 4 * lpp) * 2;
    const unsigned char* lbX = sX + lrow * C3_SX + lcol;
    const unsigned char* lbD = sD + lrow * C3_SD + lcol;
    auto tr2 = [&](const unsigned char* p, int stride) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * stride));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    const int64_t tiles = (M + 31) / 32;
    const int64_t step = (int64_t)gridDim.x * 4;
    int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    prefetch(tile);
    for (; tile < tiles; tile += step) {
        // the wave's LDS operations execute in order: these stores follow every read of the previous tile
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<uint4*>(sX + r * C3_SX + (2 * j + hh) * 16) = px[j];
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<uint4*>(sD + (lane + j * 64) * 16) = pd[j];
        wave_lds_fence();
        prefetch(tile + step);
        f32x16 y = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const bf16x8 pv = *reinterpret_cast<const bf16x8*>(sX + r * C3_SX + (16 * i + 8 * hh) * 2);
            y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pv, wf[i], y, 0, 0, 0);
        }
        float yf[2][8];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float ylo = y[8 * c + k], yhi = y[8 * c + 4 + k];       // (scalars first: see k_c3_bn_bwd)
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ylo), __float_as_uint(yhi), false, false);
                const uint32_t s0 = sw[0], s1_ = sw[1];
                yf[c][k] = __uint_as_float(s0);
                yf[c][4 + k] = __uint_as_float(s1_);
            }
        const int64_t m0 = tile * 32;
        const bool tail = __builtin_amdgcn_readfirstlane((int)(m0 + 32 > M)) != 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bf16x8 dzf = tr2(lbD + c * 16 * C3_SD, C3_SD);          // pixels 16 c + 8 hh + j of channel r
            const s16x8 dzs = __builtin_bit_cast(s16x8, dzf);
            float dv[8], hv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = __uint_as_float(((uint32_t)(uint16_t)dzs[j]) << 16);
                dv[j] = POST == TCCT_ACT_NONE ? d : d * post_grad<POST>(a_r * yf[c][j] + b_r);
                hv[j] = yf[c][j] * rs_r - murs_r;
            }
            if (tail) {         // wave-uniform: the last tile only
                const int64_t p0 = m0 + 16 * c + 8 * hh;
#pragma unroll
                for (int j = 0; j < 8; ++j) if (p0 + j >= M) hv[j] = 0.f;
            }
            uint32_t pk[4], ph[4];
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                s1 += dv[j] + dv[j + 1]; s2 += dv[j] * hv[j] + dv[j + 1] * hv[j + 1]; s3 += hv[j] + hv[j + 1];
                if (POST != TCCT_ACT_NONE) pk[j >> 1] = pack_bf16x2(dv[j], dv[j + 1]);
                ph[j >> 1] = pack_bf16x2(hv[j], hv[j + 1]);
            }
            const bf16x8 dzp = POST == TCCT_ACT_NONE ? dzf : __builtin_bit_cast(bf16x8, make_uint4(pk[0], pk[1], pk[2], pk[3]));
            const bf16x8 yhp = __builtin_bit_cast(bf16x8, make_uint4(ph[0], ph[1], ph[2], ph[3]));
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const bf16x8 xf = tr2(lbX + c * 16 * C3_SX + 64 * kt, C3_SX);      // (patch rows of pixels beyond M are zeros)
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dzp, xf, acc[kt], 0, 0, 0);
                acc2[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yhp, xf, acc2[kt], 0, 0, 0);
                acc3[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, xf, acc3[kt], 0, 0, 0);      // every row = sum over the 16 pixels of the patch column
            }
        }
    }
    __syncthreads();
    float* redf = reinterpret_cast<float*>(smem);
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int co = (k & 3) + 8 * (k >> 2) + 4 * hh, o = co * 64 + kt * 32 + r;
                    redf[o] = turn == 0 ? acc[kt][k] : redf[o] + acc[kt][k];
                    redf[2048 + o] = turn == 0 ? acc2[kt][k] : redf[2048 + o] + acc2[kt][k];
                }
            if (hh == 0) {      // row 0 of the ones product (k = 0, hh = 0): column kt * 32 + r
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) redf[4096 + kt * 32 + r] = turn == 0 ? acc3[kt][0] : redf[4096 + kt * 32 + r] + acc3[kt][0];
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < 2 * 2048 + 64; i += C3B) {
        const int cl = i & 63;
        if (cl < 48 && ((cl >> 2) & 3) < 3 && (cl & 3) < 3) atomicAdd(&dw[i], redf[i]);
    }
    __syncthreads();
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64); s3 += __shfl_xor(s3, 32, 64);
    if (lane < 32) { redf[wave * 96 + r] = s1; redf[wave * 96 + 32 + r] = s2; redf[wave * 96 + 64 + r] = s3; }
    __syncthreads();
    if (tid < 96) {
        double a = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) a += (double)redf[wv * 96 + tid];
        atomicAdd(&red[tid], a);
    }
}

static int g_c3_bwd_pf = 4;
static int c3_bwd_pf() { return g_c3_bwd_pf; }
/* kernel A/B (tools/c3bwd_bench.py) of the one-pass backward: 4 = wave-private 32-pixel tiles (k_c3_bn_bwd_wave); 1..3 = the block-tile kernel with that many tiles requested
 * ahead; 0 = the block-tile kernel, one tile ahead, on up to 1024 blocks (the round-5 launch); returns the previous value */
extern "C" int64_t tcct_c3_bn_bwd_prefetch(int tiles_ahead) {
    const int old = g_c3_bwd_pf;
    if (tiles_ahead >= 0 && tiles_ahead <= 4) g_c3_bwd_pf = tiles_ahead;
    return old;
}
static int c3_bn_bwd_launch(int mode, const void* x4, const float* w, const float* bias, const void* dz, int B, int H, int W, int stride,
                            const float* ab, const float* coef, double* red, float* dw, float* dbias, int post_act, hipStream_t st) {
    const C3Geom g = c3_geom(B, H, W, stride);
    const int64_t M = (int64_t)B * g.Ho * g.Wo;
    const size_t lds = (size_t)C3_P * (C3_SX + C3_SD) + 32 * C3_SW;
    const int64_t tiles = (M + C3_P - 1) / C3_P;
    // weight gradient: >= 24 tiles per block (every block ends with 864 same-address atomics), between one and four blocks per CU; the reduction ends
    // with 64 atomics per block: up to four blocks per CU whatever the size (stride 2 at the bench shape ran 287 blocks: 0.082 ms for a 127 MB read)
    const int per = mode == 0 ? 8 : 24;          // (mode 2 ends with 6144 + 96 atomics per block: as few blocks as mode 1)
    int gx = (int)(tiles / per < 256 ? 256 : (tiles / per > 1024 ? 1024 : tiles / per));
    if (gx > tiles) gx = (int)tiles;
#define C3L(MD, PA, PFD) hipLaunchKernelGGL((k_c3_bn_bwd<MD, PA, PFD>), dim3(gx), dim3(C3B), lds, st, (const bf16*)x4, w, bias, (const bf16*)dz, M, ab, coef, red, dw, dbias, g)
    if (mode == 0) { if (post_act == TCCT_ACT_HSWISH) C3L(0, TCCT_ACT_HSWISH, 1); else C3L(0, TCCT_ACT_NONE, 1); }
    else if (mode == 2 && c3_bwd_pf() == 4) {
        // wave-private tiles: 4 x 8 KB of tiles + the weights; one round of blocks (two per CU at ~220 VGPRs)
        const size_t ldsw = (size_t)4 * 32 * (C3_SX + C3_SD) + 32 * C3_SW;
        const int64_t t32 = (M + 31) / 32;
        int gw = (int)((t32 + 3) / 4 < 512 ? (t32 + 3) / 4 : 512);
        if (gw < 1) gw = 1;
        if (post_act == TCCT_ACT_HSWISH) hipLaunchKernelGGL((k_c3_bn_bwd_wave<TCCT_ACT_HSWISH>), dim3(gw), dim3(C3B), ldsw, st, (const bf16*)x4, w, bias, (const bf16*)dz, M, ab, coef, red, dw, g);
        else hipLaunchKernelGGL((k_c3_bn_bwd_wave<TCCT_ACT_NONE>), dim3(gw), dim3(C3B), ldsw, st, (const bf16*)x4, w, bias, (const bf16*)dz, M, ab, coef, red, dw, g);
    }
    else if (mode == 2) {
        const int pf = c3_bwd_pf();
        if (gx > 512 && pf > 0) gx = 512;   // two blocks per CU are resident (VGPRs): one round of blocks, half the closing atomics
        if (post_act == TCCT_ACT_HSWISH) { if (pf <= 1) C3L(2, TCCT_ACT_HSWISH, 1); else if (pf == 2) C3L(2, TCCT_ACT_HSWISH, 2); else C3L(2, TCCT_ACT_HSWISH, 3); }
        else { if (pf <= 1) C3L(2, TCCT_ACT_NONE, 1); else if (pf == 2) C3L(2, TCCT_ACT_NONE, 2); else C3L(2, TCCT_ACT_NONE, 3); }
    }
    else { if (post_act == TCCT_ACT_HSWISH) C3L(1, TCCT_ACT_HSWISH, 1); else C3L(1, TCCT_ACT_NONE, 1); }
#undef C3L
    return 0;
}

/* Backward of tcct_c3_bn_fwd_train, pass 1: raw[0..32) = sum dz', raw[32..64) = sum dz' y with dz' = dz post'(a y + b), y recomputed from the
 * image (raw fp64 [64], zero on entry; tcct_bn_bwd_coef(raw = 1) turns it into the constants of pass 2 and into dgamma / dbeta).
 * dz bf16 [B,Ho,Wo,32]; ab [64] as written by the forward. */
extern "C" int tcct_c3_bn_bwd_reduce(const void* x4, const float* w, const float* bias, const void* dz, int B, int H, int W, int stride,
                                     const float* ab, double* raw, int post_act, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "c3_bn_bwd_reduce: stride %d", stride);
    TCCT_CHECK(post_act == TCCT_ACT_NONE || post_act == TCCT_ACT_HSWISH, "c3_bn_bwd_reduce: post_act %d (none or hswish)", post_act);
    const C3Geom g = c3_geom(B, H, W, stride);
    const int64_t M = (int64_t)B * g.Ho * g.Wo, inb = (int64_t)B * H * W * 8;
    TCCT_CHECK(M > 0 && M * 64 < (1ll << 31) && inb < (1ll << 31), "c3_bn_bwd_reduce: image too large for 32-bit byte offsets (B=%d H=%d W=%d)", B, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(raw, 0, sizeof(double) * 64, st) != hipSuccess) { tcct_set_error("c3_bn_bwd_reduce: memset failed"); return -2; }
    c3_bn_bwd_launch(0, x4, w, bias, dz, B, H, W, stride, ab, nullptr, raw, nullptr, nullptr, post_act, st);
    TCCT_LAUNCH_OK();
}
/* pass 2: dw fp32 [32,3,3,3] and dbias fp32 [32] (nullable) of the convolution, both overwritten (left to the caller's zero pool when that is
 * active), from dy = c1 dz' + c2 y + c3 rebuilt in registers; coef fp32 [160] = {c1, c2, c3, a, b} from tcct_bn_bwd_coef. */
extern "C" int tcct_c3_bn_bwd_wgrad(const void* x4, const float* w, const float* bias, const void* dz, int B, int H, int W, int stride,
                                    const float* coef, float* dw, float* dbias, int post_act, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "c3_bn_bwd_wgrad: stride %d", stride);
    TCCT_CHECK(post_act == TCCT_ACT_NONE || post_act == TCCT_ACT_HSWISH, "c3_bn_bwd_wgrad: post_act %d (none or hswish)", post_act);
    const C3Geom g = c3_geom(B, H, W, stride);
    const int64_t M = (int64_t)B * g.Ho * g.Wo, inb = (int64_t)B * H * W * 8;
    TCCT_CHECK(M > 0 && M * 64 < (1ll << 31) && inb < (1ll << 31), "c3_bn_bwd_wgrad: image too large for 32-bit byte offsets (B=%d H=%d W=%d)", B, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * 32 * 27, st) != hipSuccess) { tcct_set_error("c3_bn_bwd_wgrad: memset failed"); return -2; }
    if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * 32, st) != hipSuccess) { tcct_set_error("c3_bn_bwd_wgrad: memset failed"); return -2; }
    c3_bn_bwd_launch(1, x4, w, bias, dz, B, H, W, stride, coef + 96, coef, nullptr, dw, dbias, post_act, st);
    TCCT_LAUNCH_OK();
}

// ---- single-pass backward (MODE 2): combination of the accumulated matrices
__global__ void k_c3_bn_bwd_fin(const float* __restrict__ wk /*[2][32][64] + [64]*/, const double* __restrict__ red /*[96]*/, int64_t M, const float* __restrict__ ab,
                                float* __restrict__ dw /*[32][3][3][3]*/, float* __restrict__ dbias, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sa[32], ss1[32], ss2[32];
    const int t = threadIdx.x;
    if (t < 32) {
        const double S1 = red[t], S2 = red[32 + t], S3 = red[64 + t];
        const double a = ab[t], s1 = S1 / (double)M, s2 = S2 / (double)M;
        sa[t] = (float)a; ss1[t] = (float)s1; ss2[t] = (float)s2;
        dbeta[t] = (float)S1; dgamma[t] = (float)S2;
        if (dbias) dbias[t] = (float)(a * (S1 - s2 * S3 - s1 * (double)M));      // (mathematically zero: the BatchNorm removes a bias; kept as the sum it is)
    }
    __syncthreads();
    for (int i = t; i < 32 * 27; i += blockDim.x) {
        const int co = i / 27, j = i - co * 27, ch = j / 9, ky = (j - ch * 9) / 3, kx = j - ch * 9 - ky * 3;
        const int cl = 16 * ky + 4 * kx + ch;       // patch element of w[co][ch][ky][kx]
        dw[i] = sa[co] * (wk[co * 64 + cl] - ss2[co] * wk[2048 + co * 64 + cl] - ss1[co] * wk[4096 + cl]);
    }
}
/* Backward of tcct_c3_bn_fwd_train in ONE pass over dz (+ a one-block combination): dw fp32 [32,3,3,3], dbias [32] (nullable), dgamma, dbeta [32] are
 * overwritten; work fp32 [4160] and sums fp64 [96] are scratch (zero on entry unless the outputs are pre-zeroed: cleared here otherwise).
 * mean_rstd [64] and ab [64] as written by the forward. */
extern "C" int tcct_c3_bn_bwd_onepass(const void* x4, const float* w, const float* bias, const void* dz, int B, int H, int W, int stride, const float* mean_rstd,
                                      const float* ab, float* work, double* sums, float* dw, float* dbias, float* dgamma, float* dbeta, int post_act,
                                      tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "c3_bn_bwd_onepass: stride %d", stride);
    TCCT_CHECK(post_act == TCCT_ACT_NONE || post_act == TCCT_ACT_HSWISH, "c3_bn_bwd_onepass: post_act %d (none or hswish)", post_act);
    TCCT_CHECK(mean_rstd && ab && work && sums && dw && dgamma && dbeta, "c3_bn_bwd_onepass: NULL argument");
    const C3Geom g = c3_geom(B, H, W, stride);
    const int64_t M = (int64_t)B * g.Ho * g.Wo, inb = (int64_t)B * H * W * 8;
    TCCT_CHECK(M > 0 && M * 64 < (1ll << 31) && inb < (1ll << 31), "c3_bn_bwd_onepass: image too large for 32-bit byte offsets (B=%d H=%d W=%d)", B, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && (hipMemsetAsync(work, 0, sizeof(float) * (2 * 2048 + 64), st) != hipSuccess || hipMemsetAsync(sums, 0, sizeof(double) * 96, st) != hipSuccess)) {
        tcct_set_error("c3_bn_bwd_onepass: memset failed"); return -2;
    }
    c3_bn_bwd_launch(2, x4, w, bias, dz, B, H, W, stride, ab, mean_rstd, sums, work, nullptr, post_act, st);
    hipLaunchKernelGGL(k_c3_bn_bwd_fin, dim3(1), dim3(256), 0, st, (const float*)work, (const double*)sums, M, ab, dw, dbias, dgamma, dbeta);
    TCCT_LAUNCH_OK();
}
