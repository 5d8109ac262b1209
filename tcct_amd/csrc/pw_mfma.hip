// MFMA pointwise (1x1) convolution / nn.Linear on bf16 NHWC rows: Y[M,N] = X[M,K] * Wt + bias   (reference
// nets/tcct.py:41-43,124,532-546,600,966-997).  K, N multiples of 32 (N may be < 32 for the 5-class aux heads).
//   D[co][pixel] per 32-pixel M-tile and 32-channel N-tile with v_mfma_f32_32x32x16_bf16; A = weights from LDS (staged once
//   per block as bf16 [N][K], row stride 2K+16 B => conflict-free ds_read_b128), B = activations straight from global
//   memory (every X element is used by exactly one wave, so an LDS round trip would be pure overhead): the K order inside
//   each 32-channel group is permuted so that a lane's two k-steps are 32 CONTIGUOUS bytes of its pixel row.
// wgrad: dW[co][ci] = sum_p dy[p][co] x[p][ci] with both tiles staged pixel-major in LDS and transposed on read by
//   ds_read_b64_tr_b16 (same scheme as conv_mfma.hip).
#include "common.h"
#include "c3_geom.h"
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4;

#define PWB 256

__device__ __forceinline__ void st_out4(bf16* p, float a, float b, float c, float d) {
    uint2 o; o.x = pack_bf16x2(a, b); o.y = pack_bf16x2(c, d);
    *reinterpret_cast<uint2*>(p) = o;
}
__device__ __forceinline__ void st_out4(float* p, float a, float b, float c, float d) {
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// NT = number of 32-channel output tiles handled per block (N chunk = NT*32 starting at blockIdx.y*NT*32)
__device__ __forceinline__ float rnd_out(float v, const bf16*) { return __bfloat162float(__float2bfloat16(v)); }
__device__ __forceinline__ float rnd_out(float v, const float*) { return v; }

// Concatenation-free operands (MHCA_stage.aggregate, reference nets/tcct.py:600-616: `cat([InvRes(x), Encoder(x)], 1)` -> 1x1 conv): the
// input rows may live in TWO tensors (channels [0,K1) in x, [K1,K) in x2) and, for the input-gradient GEMM, the output rows may go
// to two tensors (channels [0,N1) to y, [N1,N) to y2).  K1 / N1 are multiples of 32; x2 == NULL / y2 == NULL: ordinary operands.
// Direct 3-channel 3x3 convolution (C3 kernels below): x is the 4-channel NHWC image [B,H,W,4]; the "patch row" of an output pixel is
// gathered on the fly instead of being read from an im2col copy, in the order k = 16*ky + 4*slot + ch with slot 0, 1, 2 = kx 0, 1, 2 and
// slot 3 / channel 3 zero (K = 48): k-step ky of an MFMA is then, per lane half, two whole 8-byte pixels of ONE input row (kx 0,1 / kx 2,-)
// and needs no repacking, and the row validity of a load is one compare per ky
// (C3Geom, udiv_m: c3_geom.h)
// up (round 4, fp32 output with N <= 8): y += resize_x2(up) with align_corners, up fp32 [B, uH, uW, 8] at half the resolution (channels >= N zero) -- the level-0 head
// with the resize commuted behind the convolution adds up(Wa y) to Wb skip + c in the epilogue of the skip GEMM instead of in a pass of its own
struct PwUp { const float* p; int uH, uW, Ho, Wo; float sh, sw; uint32_t m_howo, m_wo; };
struct PwSplit { const bf16* x2; int K1; void* y2; int N1; const bf16* res; const float* rscale; int64_t per_sample; bf16* yplain; C3Geom c3; PwUp up; };
// res (inference-epilogue kernel only): y = res + rscale[m / per_sample] * (x W^T + bias), rscale nullable -- Mlp.fc2 with the
// residual add and the DropPath scale of MHCABlock folded in (reference nets/tcct.py:468)
// STATS: also accumulate per-channel sum / sum of squares of pre_act(y) (y as stored) into stats[0..N) / stats[N..2N): the
// train-mode BatchNorm statistics of the consumer (Conv2d_BN, DWConv2d_BN.pwconv, tran_*; reference nets/tcct.py:80,125,966-974)
// C3: the first layers of both encoders (cnn.0 / stem.0, reference nets/tcct.py:873,674-681), 3 -> 32 channels, 3x3, pad 1, stride 1 / 2:
// `x` is the 4-channel image, `w` the convolution weight as stored ([32][3][3][3] fp32), K = 48 (order above), an M-tile = 32 consecutive
// pixels of ONE output row.  Per tile a lane issues six branch-free 8-byte buffer loads (outside the image / beyond the last tile = out-of-
// range offset = zeros) two tiles ahead -- a load inside a branch makes the compiler wait for ALL outstanding loads at the top of every
// tile; epilogues (coalescing transpose, BatchNorm statistics, inference affine) are the pointwise kernel's.
template <int NT, typename Tout, bool STATS, bool AFF, bool C3 = false>
__global__ void __launch_bounds__(PWB, 2)
k_pw_fwd(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, Tout* __restrict__ y,
         int64_t M, int K, int N, int transposed, double* __restrict__ stats, int stat_pre, const float* __restrict__ aff,
         int aff_pre, int aff_post, PwSplit sp) {
    // aff != NULL or aff_pre/aff_post != 0 (inference only): y = post(a[c] * pre(x W + bias) + b[c]), aff = {a[N], b[N]}
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int SW = 2 * K + 16;                 // LDS row stride (bytes)
    const int scr_off = (NT * 32 * SW + 15) & ~15;   // per-wave epilogue transpose scratch (4 x 2560 B) behind the weights
    // bias of this block's NT*32 output channels behind the scratch: a global load of bias[] in the epilogue costs an s_waitcnt vmcnt(0) per
    // tile, i.e. it also waits for every activation load issued ahead
    float* sbias = reinterpret_cast<float*>(smem + scr_off + 4 * 2560);
    const int n_base = blockIdx.y * NT * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    for (int i = tid; i < NT * 32; i += PWB) sbias[i] = (bias && n_base + i < N) ? bias[n_base + i] : 0.f;
    // stage weights: rows n_base .. n_base+NT*32-1 (zero beyond N), bf16 [row][K]; eight elements per thread and iteration, read
    // along the contiguous direction of w (the element-wise loop with a division per element was most of a small-M launch)
    if (C3) {                   // k = 16*ky + 4*kx + ch  <-  w[n][ch][ky][kx]
        for (int i = tid; i < 32 * 48; i += PWB) {
            const int row = i / 48, k = i - row * 48, ky = k >> 4, kx = (k >> 2) & 3, ch = k & 3;
            *reinterpret_cast<bf16*>(smem + row * SW + k * 2) = __float2bfloat16((kx < 3 && ch < 3) ? w[row * 27 + ch * 9 + ky * 3 + kx] : 0.f);
        }
    } else if (!transposed) {
        const int K8 = K >> 3;
        for (int i = tid; i < NT * 32 * K8; i += PWB) {
            const int row = i / K8, c8 = i - row * K8, n = n_base + row;
            uint4 o = make_uint4(0, 0, 0, 0);
            if (n < N) {
                const float4 a = *reinterpret_cast<const float4*>(w + (int64_t)n * K + c8 * 8);
                const float4 b = *reinterpret_cast<const float4*>(w + (int64_t)n * K + c8 * 8 + 4);
                o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w); o.z = pack_bf16x2(b.x, b.y); o.w = pack_bf16x2(b.z, b.w);
            }
            *reinterpret_cast<uint4*>(smem + row * SW + c8 * 16) = o;
        }
    } else {
        const int R8 = NT * 4;                      // groups of 8 output rows
        for (int i = tid; i < K * R8; i += PWB) {
            const int k = i / R8, c8 = i - k * R8, n0 = n_base + c8 * 8;
            float v[8];
            if (n0 + 7 < N && (N & 3) == 0) {
                const float4 a = *reinterpret_cast<const float4*>(w + (int64_t)k * N + n0);
                const float4 b = *reinterpret_cast<const float4*>(w + (int64_t)k * N + n0 + 4);
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = n0 + j < N ? w[(int64_t)k * N + n0 + j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<bf16*>(smem + (c8 * 8 + j) * SW + k * 2) = __float2bfloat16(v[j]);
        }
    }
    __syncthreads();
    // statistics are taken after the epilogue transpose, where a lane owns 8 channels of each N-tile: 16 registers per tile
    // (32 before the transpose, which spilled for NT = 3); STATS implies bf16 output with N % 32 == 0
    float ss[STATS ? NT : 1][8], sq[STATS ? NT : 1][8];
#pragma unroll
    for (int a = 0; a < (STATS ? NT : 1); ++a)
#pragma unroll
        for (int k = 0; k < 8; ++k) ss[a][k] = sq[a][k] = 0.f;
    const int KT = K >> 5;
    const int64_t mtiles = C3 ? (M / sp.c3.Wo) * sp.c3.tpr : (M + 31) >> 5;        // C3: (B*Ho output rows) x (tiles per row)
    uint32_t c3q[12], c3p[12];          // the lane's six pixels (ky = 0..2; kx = 0,1 in the lower lane half, kx = 2 in the upper) of the next two tiles of this wave
    const __amdgpu_buffer_rsrc_t c3r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, C3 ? sp.c3.bytes : 0u, 0x00020000);
    // C3 tile -> output row (n*Ho + oy) and 32-pixel block inside the row
    auto c3_tile = [&](int64_t mt_, uint32_t& row, uint32_t& xb) {
        row = udiv_m((uint32_t)mt_, (uint32_t)sp.c3.tpr, sp.c3.m_tpr);
        xb = (uint32_t)mt_ - row * (uint32_t)sp.c3.tpr;
    };
    auto c3_load = [&](uint32_t (&cq)[12], int64_t mt_) {       // no branches (see above): a tile beyond the last one loads zeros
        const bool live = mt_ < mtiles;
        uint32_t row, xb;
        c3_tile(live ? mt_ : 0, row, xb);
        const uint32_t n_ = udiv_m(row, (uint32_t)sp.c3.Ho, sp.c3.m_ho), oy = row - n_ * (uint32_t)sp.c3.Ho;
        const int ox = (int)xb * 32 + r;
        const int iy0 = (int)oy * sp.c3.stride - 1, ix0 = ox * sp.c3.stride - 1;
        const bool in = live & (ox < sp.c3.Wo);
        const int pa = ix0 + 2 * hh, pb = ix0 + 1;
        const bool va = in & ((unsigned)pa < (unsigned)sp.c3.W), vb = in & (hh == 0) & ((unsigned)pb < (unsigned)sp.c3.W);
        const int base = ((int)(n_ * (uint32_t)sp.c3.H) + iy0) * sp.c3.W;          // first pixel of input row iy0 (only used where that row exists)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const bool rv = (unsigned)(iy0 + ky) < (unsigned)sp.c3.H;
            const uint32_t offa = (va & rv) ? (uint32_t)(base + ky * sp.c3.W + pa) * 8u : 0x80000000u;
            const uint32_t offb = (vb & rv) ? (uint32_t)(base + ky * sp.c3.W + pb) * 8u : 0x80000000u;
            const uint2 ta = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(c3r, offa, 0, 0));
            const uint2 tb = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(c3r, offb, 0, 0));
            cq[4 * ky] = ta.x; cq[4 * ky + 1] = ta.y; cq[4 * ky + 2] = tb.x; cq[4 * ky + 3] = tb.y;
        }
    };
    const int64_t mstep = (int64_t)gridDim.x * 4;
    auto do_tile = [&](int64_t mt, uint32_t (&cq)[12]) {
        int64_t mbase = mt * 32, mlim = M;           // first pixel of the tile, one past its last valid pixel
        if (C3) {
            uint32_t row, xb;
            c3_tile(mt, row, xb);
            mbase = (int64_t)row * sp.c3.Wo + xb * 32;
            mlim = mt < mtiles ? (int64_t)(row + 1) * sp.c3.Wo : 0;        // the unrolled loop below may run one tile past the end: nothing valid in it
        }
        const int64_t m = mbase + r;
        const bool ok = m < mlim;
        const int64_t mrow = ok ? m : 0;
        const int KA = sp.x2 ? sp.K1 : K;              // row length of the first source
        const bf16* xr = x + mrow * KA + 16 * hh;
        const bf16* xr2 = sp.x2 ? sp.x2 + mrow * (K - KA) + 16 * hh - KA : xr;    // indexed with the global channel offset 32t
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[nt][k] = 0.f;
        if (C3) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const bf16x8 bv = __builtin_bit_cast(bf16x8, make_uint4(cq[4 * i], cq[4 * i + 1], cq[4 * i + 2], cq[4 * i + 3]));
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(smem + r * SW + (16 * i + 8 * hh) * 2);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[0], 0, 0, 0);
            }
            // two tiles ahead (a gather is latency, not bandwidth), into the registers the MFMAs have just read: loading them into fresh
            // registers while these are live costs a copy at the loop's back edge, i.e. a wait for every load in flight
            __builtin_amdgcn_sched_barrier(0);
            c3_load(cq, mt + 2 * mstep);
        } else
        for (int t = 0; t < KT; ++t) {
            // channels 32t + 16hh + [0,16): two k-steps
            uint4 u0 = make_uint4(0, 0, 0, 0), u1 = u0;
            if (ok) {
                const bf16* src = 32 * t < KA ? xr : xr2;
                u0 = *reinterpret_cast<const uint4*>(src + 32 * t);
                u1 = *reinterpret_cast<const uint4*>(src + 32 * t + 8);
            }
            const bf16x8 b0 = __builtin_bit_cast(bf16x8, u0), b1 = __builtin_bit_cast(bf16x8, u1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const unsigned char* wr = smem + (nt * 32 + r) * SW + (32 * t + 16 * hh) * 2;
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(wr);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(wr + 16);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[nt], 0, 0, 0);
            }
        }
        // epilogue.  bf16 output with N % 32 == 0: each 32x32 tile goes through a per-wave LDS transpose (80-byte pixel rows) so that
        // every lane stores 16 contiguous bytes and a wave instruction covers 16 whole 64-byte pixel segments; the direct form
        // (four 8-byte stores per lane, 16 B of every 64-B line per instruction) held store-heavy shapes at 2.7 TB/s.
        constexpr bool kBf16 = sizeof(Tout) == 2;
        if (kBf16 && (N & 31) == 0) {
            unsigned char* sc = smem + scr_off + wave * 2560;          // 32 pixels x 80 B
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = n_base + nt * 32 + 8 * q + 4 * hh;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[nt][4 * q + k] + sbias[nt * 32 + 8 * q + 4 * hh + k];
                    if (AFF) {
                        float a4[4], b4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { a4[k] = aff ? aff[co + k] : 1.f; b4[k] = aff ? aff[N + co + k] : 0.f; }
                        affine4(v, a4, b4, aff_pre, aff_post);
                    }
                    uint2 o; o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(sc + r * 80 + (8 * q + 4 * hh) * 2) = o;
                }
                // same wave wrote and reads: LDS operations of one wave complete in order (the fence is for the compiler)
                wave_lds_fence();
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int p = (lane >> 2) + 16 * h2, cch = lane & 3;
                    const uint4 o = *reinterpret_cast<const uint4*>(sc + p * 80 + cch * 16);
                    const int64_t mm = mbase + p;
                    if (mm < mlim) {
                        const int n0 = n_base + nt * 32;
                        bf16* dst = (sp.y2 && n0 >= sp.N1) ? reinterpret_cast<bf16*>(sp.y2) + mm * (N - sp.N1) + (n0 - sp.N1)
                                                           : reinterpret_cast<bf16*>(y) + mm * (sp.y2 ? sp.N1 : N) + n0;
                        if (AFF && sp.res) {                    // y = res + s * y (values already rounded to bf16, like the op-by-op path)
                            if (sp.yplain) *reinterpret_cast<uint4*>(sp.yplain + mm * N + n0 + cch * 8) = o;     // the Linear output itself
                            const uint4 rv = *reinterpret_cast<const uint4*>(sp.res + mm * N + n0 + cch * 8);
                            const float sc_ = sp.rscale ? sp.rscale[mm / sp.per_sample] : 1.f;
                            const uint32_t ov[4] = {o.x, o.y, o.z, o.w}, rr[4] = {rv.x, rv.y, rv.z, rv.w};
                            uint32_t pk[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                pk[k] = pack_bf16x2(__uint_as_float(rr[k] << 16) + sc_ * __uint_as_float(ov[k] << 16),
                                                    __uint_as_float(rr[k] & 0xffff0000u) + sc_ * __uint_as_float(ov[k] & 0xffff0000u));
                            *reinterpret_cast<uint4*>(dst + cch * 8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                        } else
                        *reinterpret_cast<uint4*>(dst + cch * 8) = o;
                        if (STATS) {
                            const uint32_t wv[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                // (stat_pre == none, the usual case, without act_fwd's per-value chain of scalar compares and branches)
                                float u0 = __uint_as_float(wv[k] << 16), u1 = __uint_as_float(wv[k] & 0xffff0000u);
                                if (!C3 && stat_pre != TCCT_ACT_NONE) { u0 = act_fwd(stat_pre, u0); u1 = act_fwd(stat_pre, u1); }
                                ss[nt][2 * k] += u0; sq[nt][2 * k] += u0 * u0; ss[nt][2 * k + 1] += u1; sq[nt][2 * k + 1] += u1 * u1;
                            }
                        }
                    }
                }
                wave_lds_fence();       // the next N-tile's writes come after every lane's reads of this one
            }
        } else if (ok) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int co = n_base + nt * 32 + 8 * q + 4 * hh;
                    float v[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = acc[nt][4 * q + k] + sbias[nt * 32 + 8 * q + 4 * hh + k];
                    if (AFF) {
                        float a4[4], b4[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { a4[k] = (aff && co + k < N) ? aff[co + k] : 1.f; b4[k] = (aff && co + k < N) ? aff[N + co + k] : 0.f; }
                        affine4(v, a4, b4, aff_pre, aff_post);
                    }
                    if (sizeof(Tout) == 4 && sp.up.p && q == 0) {          // (block-uniform) + bilinear x2 of the low-resolution addend: the lane's pixel m, channels 4 hh + k
                        const uint32_t mm = (uint32_t)m, howo = (uint32_t)(sp.up.Ho * sp.up.Wo);
                        const uint32_t n_ = udiv_m(mm, howo, sp.up.m_howo), rem = mm - n_ * howo;
                        const uint32_t ho = udiv_m(rem, (uint32_t)sp.up.Wo, sp.up.m_wo), wo = rem - ho * (uint32_t)sp.up.Wo;
                        const Lerp a = src_index((int)ho, sp.up.sh, sp.up.uH, 1), b = src_index((int)wo, sp.up.sw, sp.up.uW, 1);
                        const float* r0 = sp.up.p + ((int64_t)n_ * sp.up.uH + a.i0) * sp.up.uW * 8 + 4 * hh;
                        const float* r1 = sp.up.p + ((int64_t)n_ * sp.up.uH + a.i1) * sp.up.uW * 8 + 4 * hh;
                        const float4 v00 = *reinterpret_cast<const float4*>(r0 + b.i0 * 8), v01 = *reinterpret_cast<const float4*>(r0 + b.i1 * 8);
                        const float4 v10 = *reinterpret_cast<const float4*>(r1 + b.i0 * 8), v11 = *reinterpret_cast<const float4*>(r1 + b.i1 * 8);
                        const float t0[4] = {v00.x, v00.y, v00.z, v00.w}, t1[4] = {v01.x, v01.y, v01.z, v01.w};
                        const float t2[4] = {v10.x, v10.y, v10.z, v10.w}, t3[4] = {v11.x, v11.y, v11.z, v11.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = (a.l0 * (b.l0 * t0[k] + b.l1 * t1[k]) + a.l1 * (b.l0 * t2[k] + b.l1 * t3[k])) + v[k];       // the order of tcct_bilinear_add_fwd
                    }
                    if (co + 3 < N && (N & 3) == 0) st_out4(y + m * N + co, v[0], v[1], v[2], v[3]);
                    else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (co + k < N) stf(y + m * N + co + k, v[k]);
                    }
                }
            }
        }
    };
    const int64_t mt0 = (int64_t)blockIdx.x * 4 + wave;
    if (C3) {
        c3_load(c3q, mt0);
        __builtin_amdgcn_sched_barrier(0);      // keep the order of first use: the loop's waits are the tighter of "entered" and "came round"
        c3_load(c3p, mt0 + mstep);
        __builtin_amdgcn_sched_barrier(0);
        for (int64_t mt = mt0; mt < mtiles; mt += 2 * mstep) {      // two register sets, no branch around the second tile (see c3_load)
            do_tile(mt, c3q);
            do_tile(mt + mstep, c3p);
        }
    } else {
        for (int64_t mt = mt0; mt < mtiles; mt += mstep) do_tile(mt, c3q);
    }
    if (STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);           // [4 waves][2][NT*32] (weights are no longer needed)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < 8; ++k) {       // lanes with equal (lane & 3) hold different pixels of channels 8*(lane&3)+k
                float a = ss[nt][k], b = sq[nt][k];
#pragma unroll
                for (int o = 32; o > 2; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (lane < 4) {                 // per-wave slots, summed below: no LDS atomics
                    red[wave * 2 * NT * 32 + nt * 32 + 8 * lane + k] = a;
                    red[wave * 2 * NT * 32 + NT * 32 + nt * 32 + 8 * lane + k] = b;
                }
            }
        __syncthreads();
        for (int i2 = tid; i2 < NT * 32; i2 += PWB) {
            const int co = n_base + i2;
            if (co < N) {
                double a = 0.0, b = 0.0;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) { a += (double)red[wv * 2 * NT * 32 + i2]; b += (double)red[wv * 2 * NT * 32 + NT * 32 + i2]; }
                atomicAdd(&stats[co], a); atomicAdd(&stats[N + co], b);
            }
        }
    }
}

/* x bf16 [M,K]; w fp32: [N,K] (transposed=0) or [K,N] (transposed=1: the input-gradient GEMM dx = dy * W);
 * y [M,N] bf16 or fp32. */
static int pw_fwd_impl(const void* x, const float* w, const float* bias, void* y, int64_t M, int K, int N, int transposed,
                       int out_dtype, double* stats, int stat_pre, tcct_stream_t stream, const float* aff = nullptr, int aff_pre = 0,
                       int aff_post = 0, PwSplit sp = PwSplit{nullptr, 0, nullptr, 0, nullptr, nullptr, 1, nullptr});
struct PwRes { const bf16* res; const float* rscale; int64_t per_sample; bf16* yplain; const float* xab; const float* aff; int aff_post; int has_aff; };
static bool pw_fwd2_ok(int64_t M, int K, int N, int K1, bool has_x2);
// shapes of the tile-staged forward WITH the inference epilogue: square 64 / 96 / 128 (plain or with the residual), the concatenated 64 + 64 -> 96
static bool pw_fwd2_aff_ok(int64_t M, int K, int N, bool has_x2, bool has_res) {
    if (M * (int64_t)(K > N ? K : N) * 2 >= (1LL << 31)) return false;
    if (has_x2) return K == 128 && N == 96 && !has_res;
    return K == N && (K == 64 || K == 96 || K == 128);
}
static int pw_fwd2_launch(const void* x, const void* x2, const float* w, const float* bias, void* y, int64_t M, int K, int N, double* stats,
                          int stat_pre, tcct_stream_t stream, PwRes pr, bool gelu_x = false);
struct PwRes;
static int pw_fwd2_route(const void* x, const void* x2, const float* w, const float* bias, void* y, int64_t M, int K, int N, double* stats,
                         int stat_pre, tcct_stream_t stream, const bf16* res, const float* rscale, int64_t per_sample, bf16* yplain);
static bool pw_fwd2_enabled() {         // compile-time A/B switch (off: the direct-from-global forward kernel for every shape)
    static int on = -1;
    if (on < 0) on = 1;
    return on == 1;
}
extern "C" int tcct_pw_fwd(const void* x, const float* w, const float* bias, void* y, int64_t M, int K, int N, int transposed,
                           int out_dtype, tcct_stream_t stream) {
    return pw_fwd_impl(x, w, bias, y, M, K, N, transposed, out_dtype, nullptr, 0, stream);
}
/* forward + fused BatchNorm statistics of the consumer (bf16 output, N <= 128): stats fp64 [2N], zero on entry */
extern "C" int tcct_pw_fwd_bnstats(const void* x, const float* w, const float* bias, void* y, int64_t M, int K, int N, double* stats,
                                   int pre_act, tcct_stream_t stream) {
    TCCT_CHECK(N % 32 == 0 && N <= 160, "pw_fwd_bnstats: N=%d unsupported (32, 64, 96, 128, 160)", N);
    return pw_fwd_impl(x, w, bias, y, M, K, N, 0, TCCT_BF16, stats, pre_act, stream);
}
/* y = [x1 | x2] W^T + bias without materialising the concatenation (x1 [M,K1], x2 [M,K-K1]); stats nullable (fused BN statistics) */
extern "C" int tcct_pw_fwd_cat2(const void* x1, const void* x2, int K1, const float* w, const float* bias, void* y, int64_t M, int K, int N,
                                double* stats, int pre_act, tcct_stream_t stream) {
    TCCT_CHECK(x2 != nullptr, "pw_fwd_cat2: x2 is NULL");
    if (stats) TCCT_CHECK(N % 32 == 0 && N <= 128, "pw_fwd_cat2: fused statistics need N in {32,64,96,128} (got %d)", N);
    return pw_fwd_impl(x1, w, bias, y, M, K, N, 0, TCCT_BF16, stats, pre_act, stream, nullptr, 0, 0, PwSplit{(const bf16*)x2, K1, nullptr, 0, nullptr, nullptr, 1, nullptr});
}
/* the same with fp32 output rows (N <= 32: the composed aux head over [up(y) | skip], csrc/decoder_tail.hip) */
extern "C" int tcct_pw_fwd_cat2_f32(const void* x1, const void* x2, int K1, const float* w, const float* bias, float* y, int64_t M, int K, int N,
                                    tcct_stream_t stream) {
    TCCT_CHECK(x2 != nullptr && N >= 1 && N <= 32, "pw_fwd_cat2_f32: x2 is NULL or N=%d > 32", N);
    return pw_fwd_impl(x1, w, bias, y, M, K, N, 0, TCCT_F32, nullptr, 0, stream, nullptr, 0, 0, PwSplit{(const bf16*)x2, K1, nullptr, 0, nullptr, nullptr, 1, nullptr});
}
/* input gradient of the same convolution: [dx1 | dx2] = dy W with w [Nout, K] (the weight as stored), dy [M, Nout]; dx1 [M,K1], dx2 [M,K-K1] */
/* y fp32 [M,N] = x W^T + bias + resize_x2(up)[..., :N] (bilinear, align_corners=True) for N <= 8: x bf16 [B*Ho*Wo, K] at the full resolution, up fp32 [B,uH,uW,8]
 * at half of it (Ho = 2 uH, Wo = 2 uW; channels >= N zero).  The level-0 aux head with the resize commuted behind the convolution (nets/tcct.py:908-914,1031-1041). */
extern "C" int tcct_pw_fwd_f32_upadd(const void* x, const float* w, const float* bias, float* y, int B, int uH, int uW, int K, int N, const float* up,
                                     tcct_stream_t stream) {
    TCCT_CHECK(N >= 1 && N <= 8 && up != nullptr && B >= 1 && uH >= 1 && uW >= 1, "pw_fwd_f32_upadd: N=%d (1..8), up, B, uH, uW", N);
    const int Ho = 2 * uH, Wo = 2 * uW;
    const int64_t M = (int64_t)B * Ho * Wo;
    TCCT_CHECK(M < (1ll << 31), "pw_fwd_f32_upadd: %lld pixels exceed the 32-bit index arithmetic", (long long)M);
    PwSplit sp{nullptr, 0, nullptr, 0, nullptr, nullptr, 1, nullptr};
    sp.up = PwUp{up, uH, uW, Ho, Wo, Ho > 1 ? (float)(uH - 1) / (float)(Ho - 1) : 0.f, Wo > 1 ? (float)(uW - 1) / (float)(Wo - 1) : 0.f,
                 (uint32_t)((1ull << 32) / (uint64_t)(Ho * Wo)), (uint32_t)((1ull << 32) / (uint64_t)Wo)};
    return pw_fwd_impl(x, w, bias, y, M, K, N, 0, TCCT_F32, nullptr, 0, stream, nullptr, 0, 0, sp);
}
extern "C" int tcct_pw_dgrad_split2(const void* dy, const float* w, void* dx1, void* dx2, int K1, int64_t M, int Nout, int K,
                                    tcct_stream_t stream) {
    TCCT_CHECK(dx2 != nullptr, "pw_dgrad_split2: dx2 is NULL");
    return pw_fwd_impl(dy, w, nullptr, dx1, M, Nout, K, 1, TCCT_BF16, nullptr, 0, stream, nullptr, 0, 0, PwSplit{nullptr, 0, dx2, K1, nullptr, nullptr, 1, nullptr});
}
/* y = res + scale[m / per_sample] * (x W^T + bias)  (bf16, N % 32 == 0; scale fp32 nullable): a Linear with the residual add and the
 * per-sample DropPath scale that follow it folded into the epilogue */
extern "C" int tcct_pw_fwd_residual(const void* x, const float* w, const float* bias, const void* res, const float* scale,
                                    int64_t per_sample, void* y, void* y_plain, int64_t M, int K, int N, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr && N % 32 == 0 && per_sample >= 1, "pw_fwd_residual: needs res, N %% 32 == 0, per_sample >= 1");
    return pw_fwd_impl(x, w, bias, y, M, K, N, 0, TCCT_BF16, nullptr, 0, stream, nullptr, 0, 0,
                       PwSplit{nullptr, 0, nullptr, 0, (const bf16*)res, scale, per_sample, (bf16*)y_plain});
}
/* input gradient with a second gradient folded in: dx_plain = dy W, dx_sum = dy W + res (w [Nout, K] as stored, dy [M, Nout], res and
 * both outputs [M, K] bf16; dx_plain nullable).  Decoder block backward: dx_plain continues into the bilinear resize, dx_sum is the skip tensor's gradient */
extern "C" int tcct_pw_dgrad_residual(const void* dy, const float* w, const void* res, void* dx_sum, void* dx_plain, int64_t M, int Nout,
                                      int K, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr && K % 32 == 0, "pw_dgrad_residual: needs res, K %% 32 == 0");
    return pw_fwd_impl(dy, w, nullptr, dx_sum, M, Nout, K, 1, TCCT_BF16, nullptr, 0, stream, nullptr, 0, 0,
                       PwSplit{nullptr, 0, nullptr, 0, (const bf16*)res, nullptr, 1, (bf16*)dx_plain});
}
/* inference: y = post_act(a[c] * pre_act(x W^T + bias[c]) + b[c]), ab = {a[N], b[N]} from tcct_bn_eval_ab (NULL: a = 1, b = 0):
 * eval-mode BatchNorm and the adjacent activation(s) folded into the GEMM epilogue (Conv2d_BN, Mlp.fc1 + GELU, tran_*) */
extern "C" int tcct_pw_fwd_affine(const void* x, const float* w, const float* bias, void* y, int64_t M, int K, int N, const float* ab,
                                  int pre_act, int post_act, int out_dtype, tcct_stream_t stream) {
    return pw_fwd_impl(x, w, bias, y, M, K, N, 0, out_dtype, nullptr, 0, stream, ab, pre_act, post_act);
}
/* inference: y = res + (a[c] * (x W^T + bias[c]) + b[c]) -- `x + BN_eval(conv2(f))` of the InvRes block (nets/tcct.py:563-572) as one GEMM, the normalised product rounded
 * to bf16 before the add like the op-by-op path; K = N in {64, 96, 128}; ab NULL: a = 1, b = 0 */
extern "C" int tcct_pw_fwd_affine_residual(const void* x, const float* w, const float* bias, const float* ab, const void* res, void* y, int64_t M, int K, int N,
                                           tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr && pw_fwd2_aff_ok(M, K, N, false, true), "pw_fwd_affine_residual: K=%d N=%d unsupported (64, 96 or 128 square) or res NULL", K, N);
    return pw_fwd2_launch(x, nullptr, w, bias, y, M, K, N, nullptr, 0, stream, PwRes{(const bf16*)res, nullptr, 1, nullptr, nullptr, ab, TCCT_ACT_NONE, 1});
}
/* inference: y = res + (a[c] * (hswish(a_prev[k] * y_prev + b_prev[k]) W^T + bias[c]) + b[c]): the InvRes tail `x + BN(conv2(hswish(BN(dw))))` (nets/tcct.py:563-572) as ONE
 * GEMM over the depthwise convolution's raw output -- `norm`'s eval-mode BatchNorm + Hardswish applied while the tile is staged, conv2's BatchNorm and the residual in the
 * epilogue; ab_prev = {a[K], b[K]}, ab = {a[N], b[N]}; K = N in {64, 96, 128} */
extern "C" int tcct_pw_fwd_xaff_affine_residual(const void* y_prev, const float* ab_prev, const float* w, const float* bias, const float* ab, const void* res, void* y,
                                                int64_t M, int K, int N, tcct_stream_t stream) {
    TCCT_CHECK(y_prev && ab_prev && res != nullptr && pw_fwd2_aff_ok(M, K, N, false, true), "pw_fwd_xaff_affine_residual: K=%d N=%d unsupported (64, 96 or 128 square) or NULL argument", K, N);
    return pw_fwd2_launch(y_prev, nullptr, w, bias, y, M, K, N, nullptr, 0, stream, PwRes{(const bf16*)res, nullptr, 1, nullptr, ab_prev, ab, TCCT_ACT_NONE, 1});
}
/* inference: y = post_act(a[c] * ([x1 | x2] W^T + bias[c]) + b[c]) over the never-materialised concatenation of two 64-channel tensors, N = 96: `aggregate` of MHCA stage 0
 * (nets/tcct.py:600-616) with its eval-mode BatchNorm + Hardswish in the epilogue */
extern "C" int tcct_pw_fwd_cat2_affine(const void* x1, const void* x2, const float* w, const float* bias, const float* ab, int post_act, void* y, int64_t M, int K, int N,
                                       tcct_stream_t stream) {
    TCCT_CHECK(x2 != nullptr && pw_fwd2_aff_ok(M, K, N, true, false), "pw_fwd_cat2_affine: K=%d N=%d unsupported (64 + 64 -> 96)", K, N);
    return pw_fwd2_launch(x1, x2, w, bias, y, M, K, N, nullptr, 0, stream, PwRes{nullptr, nullptr, 1, nullptr, nullptr, ab, post_act, 1});
}
static int pw_fwd_impl(const void* x, const float* w, const float* bias, void* y, int64_t M, int K, int N, int transposed,
                       int out_dtype, double* stats, int stat_pre, tcct_stream_t stream, const float* aff, int aff_pre, int aff_post, PwSplit sp) {
    if (!transposed && out_dtype == TCCT_BF16 && (aff || aff_post) && !aff_pre && !stats && !sp.y2 && !sp.yplain && !sp.rscale && pw_fwd2_enabled()
        && pw_fwd2_aff_ok(M, K, N, sp.x2 != nullptr, sp.res != nullptr) && (!sp.x2 || sp.K1 == 64))
        return pw_fwd2_launch(x, sp.x2, w, bias, y, M, K, N, nullptr, 0, stream, PwRes{sp.res, nullptr, 1, nullptr, nullptr, aff, aff_post, 1});
    if (!transposed && out_dtype == TCCT_BF16 && !aff && !aff_pre && !aff_post && !sp.y2 && pw_fwd2_enabled()
        && pw_fwd2_ok(M, K, N, sp.K1, sp.x2 != nullptr) && (!stats || N <= 128) && !(sp.res && (sp.x2 || stats)))
        return pw_fwd2_route(x, sp.x2, w, bias, y, M, K, N, stats, stat_pre, stream, sp.res, sp.rscale, sp.per_sample, sp.yplain);
    if (sp.x2) TCCT_CHECK(sp.K1 % 32 == 0 && sp.K1 > 0 && sp.K1 < K, "pw_fwd_cat2: K1=%d must be a multiple of 32 inside (0, K)", sp.K1);
    if (sp.y2) TCCT_CHECK(sp.N1 % 32 == 0 && sp.N1 > 0 && sp.N1 < N && N % 32 == 0 && out_dtype == TCCT_BF16, "pw_dgrad_split2: N1=%d must be a multiple of 32 inside (0, N), bf16 output", sp.N1);
    TCCT_CHECK(K % 32 == 0 && K >= 32 && K <= 512, "pw_fwd: K=%d must be a multiple of 32 (<=512)", K);
    TCCT_CHECK(N >= 1 && N <= 1024, "pw_fwd: N=%d", N);
    const int ntiles = (N + 31) / 32;
    // tiles per block: largest NT <= 5 dividing the work evenly enough and fitting 2 blocks/CU when possible
    int NT = ntiles <= 5 ? ntiles : (ntiles % 5 == 0 ? 5 : (ntiles % 4 == 0 ? 4 : (ntiles % 3 == 0 ? 3 : (ntiles % 2 == 0 ? 2 : 1))));
    const int64_t mblocks = ((M + 31) / 32 + 3) / 4;
    // small M (levels 3-4): fewer N-tiles per block so that the launch has >= 512 blocks and every block stages a fraction of the
    // weights (with NT = 5 a 27 600-pixel call ran 216 blocks, one per CU, each converting the whole 160 x K matrix first)
    while (NT > 1 && mblocks * ((ntiles + NT - 1) / NT) < 512) {
        int nn = NT - 1;
        while (nn > 1 && ntiles % nn != 0) --nn;
        NT = nn;
    }
    if (stats && NT > 4) NT = 1;            // the statistics epilogue is instantiated for 1-4 N-tiles per block: 160 outputs run as five single-tile columns of blocks
    const int gy = (ntiles + NT - 1) / NT;
    size_t lds = (((size_t)NT * 32 * (2 * K + 16) + 15) & ~(size_t)15) + 4 * 2560 + (size_t)NT * 32 * 4;
    TCCT_CHECK(lds <= 160 * 1024, "pw_fwd: weights need %zu B of LDS", lds);
    const int64_t mtiles = (M + 31) / 32;
    int64_t gx = (mtiles + 3) / 4;
    // persistent blocks: 4 per CU when LDS allows (measured sweep 2/3/4/6/8: the kernel is latency-bound with only 2 x 4 waves per
    // CU in flight -- 32->32 at level 0 went 0.204 -> 0.160 ms, 64->64 0.127 -> 0.119 ms; more than 4 gains nothing)
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    const int cap = 256 * per_cu;
    if (gx > cap) gx = cap;
    hipStream_t st = (hipStream_t)stream;
#define PW_L(NTV, TO)                                                                                                        \
    do {                                                                                                                     \
        static bool attr = false;                                                                                            \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_pw_fwd<NTV, TO, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_pw_fwd<NTV, TO, false, false>), dim3((unsigned)gx, gy), dim3(PWB), lds, st, (const bf16*)x, w, bias, (TO*)y, M, K, N, transposed, nullptr, 0, aff, aff_pre, aff_post, sp); \
    } while (0)
#define PW_D(TO)                                                                  \
    switch (NT) {                                                                 \
        case 1: PW_L(1, TO); break; case 2: PW_L(2, TO); break; case 3: PW_L(3, TO); break; \
        case 4: PW_L(4, TO); break; default: PW_L(5, TO); break;                  \
    }
    if (aff || aff_pre || aff_post || sp.res) {       // inference / residual epilogue: separate instantiations, the training kernels stay as they are
        TCCT_CHECK(out_dtype == TCCT_BF16 && !stats, "pw_fwd_affine: bf16 output only");
#define PW_A(NTV) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_fwd<NTV, bf16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_fwd<NTV, bf16, false, true>), dim3((unsigned)gx, gy), dim3(PWB), lds, st, (const bf16*)x, w, bias, (bf16*)y, M, K, N, transposed, nullptr, 0, aff, aff_pre, aff_post, sp); }
        if (NT == 1) PW_A(1) else if (NT == 2) PW_A(2) else if (NT == 3) PW_A(3) else if (NT == 4) PW_A(4) else PW_A(5)
#undef PW_A
    }
    else if (stats) {
#define PW_S(NTV) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_fwd<NTV, bf16, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_fwd<NTV, bf16, true, false>), dim3((unsigned)gx, gy), dim3(PWB), lds, st, (const bf16*)x, w, bias, (bf16*)y, M, K, N, transposed, stats, stat_pre, nullptr, 0, 0, sp); }
        if (NT == 1) PW_S(1) else if (NT == 2) PW_S(2) else if (NT == 3) PW_S(3) else PW_S(4)
#undef PW_S
    }
    else if (out_dtype == TCCT_BF16) { PW_D(bf16); }
    else if (out_dtype == TCCT_F32) { PW_D(float); }
    else { tcct_set_error("pw_fwd: bad out dtype"); return -1; }
#undef PW_D
#undef PW_L
    TCCT_LAUNCH_OK();
}

/* Direct 3 -> 32 channel 3x3 convolution (pad 1, stride 1 or 2) of the 4-channel bf16 NHWC image x4 [B,H,W,4] (channel 3 is padding):
 * y [B,Ho,Wo,32] bf16 = conv(x4[..., :3], w) + bias, w fp32 [32,3,3,3] as stored by nn.Conv2d (reference nets/tcct.py:873 `cnn.0`,
 * :674-681 `stem.0`).  stats (nullable, fp64 [64], zero on entry): sum / sum of squares of pre_act(y) for the train-mode BatchNorm that
 * follows; ab / pre_act / post_act (inference): y = post(a[c] * pre(conv + bias) + b[c]) as in tcct_pw_fwd_affine.  stats and ab exclude
 * each other.  Replaces tcct_im2col3x3_c3 + tcct_pw_fwd*: the 32-channel patch tensor is never written. */
extern "C" int tcct_c3_fwd(const void* x4, const float* w, const float* bias, void* y, int B, int H, int W, int stride, double* stats,
                           int stat_pre, const float* ab, int pre_act, int post_act, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "c3_fwd: stride %d", stride);
    TCCT_CHECK(!(stats && (ab || pre_act || post_act)), "c3_fwd: fused statistics and the inference epilogue exclude each other");
    TCCT_CHECK(!stats || stat_pre == 0, "c3_fwd: the fused statistics are those of the raw output (both first layers feed a BatchNorm directly)");
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int64_t M = (int64_t)B * Ho * Wo, inb = (int64_t)B * H * W * 8;
    TCCT_CHECK(M > 0 && M < (1ll << 31) - 64 && inb < (1ll << 31), "c3_fwd: image too large for 32-bit pixel offsets (B=%d H=%d W=%d)", B, H, W);
    const int tpr = (Wo + 31) / 32;
    auto magic = [](uint32_t d) { return d == 1 ? 0xffffffffu : (uint32_t)((1ull << 32) / d); };
    PwSplit sp{nullptr, 0, nullptr, 0, nullptr, nullptr, 1, nullptr,
               C3Geom{H, W, Ho, Wo, stride, (uint32_t)inb, tpr, magic((uint32_t)tpr), magic((uint32_t)Ho), magic((uint32_t)(Ho * Wo)), magic((uint32_t)Wo)}};
    const size_t lds = (((size_t)32 * (2 * 48 + 16) + 15) & ~(size_t)15) + 4 * 2560 + 32 * 4;
    const int64_t mtiles = (int64_t)B * Ho * tpr;
    int64_t gx = (mtiles + 3) / 4;
    if (gx > 256 * 4) gx = 256 * 4;        // (512 .. 2048 blocks: 0.148 - 0.157 ms at bs 8, 800x1104 -- the count does not matter)
    hipStream_t st = (hipStream_t)stream;
    if (stats)
        hipLaunchKernelGGL((k_pw_fwd<1, bf16, true, false, true>), dim3((unsigned)gx, 1), dim3(PWB), lds, st, (const bf16*)x4, w, bias, (bf16*)y, M, 48, 32, 0,
                           stats, stat_pre, nullptr, 0, 0, sp);
    else if (ab || pre_act || post_act)
        hipLaunchKernelGGL((k_pw_fwd<1, bf16, false, true, true>), dim3((unsigned)gx, 1), dim3(PWB), lds, st, (const bf16*)x4, w, bias, (bf16*)y, M, 48, 32, 0,
                           nullptr, 0, ab, pre_act, post_act, sp);
    else
        hipLaunchKernelGGL((k_pw_fwd<1, bf16, false, false, true>), dim3((unsigned)gx, 1), dim3(PWB), lds, st, (const bf16*)x4, w, bias, (bf16*)y, M, 48, 32, 0,
                           nullptr, 0, nullptr, 0, 0, sp);
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ weight gradient
__device__ __forceinline__ bf16x8 tr_load8g(const unsigned char* base, int stride, int P, int chan0, int lane) {
    // 16 consecutive LDS pixel rows P..P+15 (row stride `stride` bytes); returns pixels P+8*(lane>>5)+j (j=0..7) of
    // channel chan0 + (lane&31)
    const int i = lane & 15, q = i >> 2, pp = i & 3, g = lane >> 4;
    const int hh = g >> 1, cb = g & 1;
    const unsigned char* a0 = base + (P + 8 * hh + q) * stride + (chan0 + 16 * cb + 4 * pp) * 2;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * stride));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

#define PW_P 128    // pixels per staged tile (8 chunks of 16: two per wave)
#define PW_XS 4     // max x staging slots per thread (128 px * 8 chunks / 256)
#define PW_DS 10    // max dy staging slots per thread (128 px * 20 chunks / 256)
// Tiles are staged through registers one tile ahead (all global loads of tile i+1 are in flight while tile i is multiplied),
// fragments are read with immediate offsets from per-lane bases and the two chunks of a wave are double-buffered.
// C3 (NT = 1, KTB = 2): weight gradient of the direct 3-channel convolution (k_pw_fwd<..., C3>): x is the 4-channel image, the 128 x 48 patch
// tile (k = 16*ky + 4*kx + ch, see C3Geom) is gathered into LDS -- thread = (pixel, 16-byte chunk q: ky = q/2, kx = 0,1 (q even) or 2,- (q odd)),
// chunk index wave-uniform -- and dw is the stored [32][3][3][3] layout
template <int NT, int KTB, bool C3 = false>
__global__ void __launch_bounds__(PWB, 2)
k_pw_wgrad(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias, int64_t M,
           int K, int N, int SX, int SD, const bf16* __restrict__ x2, int K1, int64_t ldy, C3Geom g3 = C3Geom{0, 0, 0, 0, 0, 0u, 0, 0u, 0u, 0u, 0u}) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sD = smem + PW_P * SX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int ci_base = blockIdx.y * KTB * 32;
    // concatenation-free x: channels [0,K1) from x, [K1,K) from x2 (block-uniform choice: K1 is a multiple of KTB*32)
    const bf16* xsrc = (x2 && ci_base >= K1) ? x2 + (ci_base - K1) : x + ci_base;
    const int ldx = x2 ? (ci_base >= K1 ? K - K1 : K1) : K;
    const int xc = KTB * 4, dc = N >> 3;            // 16-byte chunks per staged pixel row
    f32x16 acc[NT][KTB];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < KTB; ++b)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[a][b][k] = 0.f;
    float bsum[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) bsum[a] = 0.f;
    // staging slots: slot j of the x image = (pixel xp[j], chunk xq[j]); packed as pixel<<8 | chunk, -1 = unused
    constexpr int DSN = C3 ? 2 : PW_DS;         // dy staging slots in use (C3: 128 px x 4 chunks / 256 threads)
    static_assert(!C3 || (NT == 1 && KTB == 2), "C3 weight gradient: 32 outputs, 48 (-> 64) patch elements");
    int xsl[PW_XS], dsl[DSN];
#pragma unroll
    for (int j = 0; j < PW_XS; ++j) {
        int i = tid + j * PWB; xsl[j] = i < PW_P * xc ? ((i / xc) << 8) | (i % xc) : -1;
        if (C3) xsl[j] = j < 3 ? ((tid & 127) << 8) | (2 * j + (tid >> 7)) : -1;       // pixel tid % 128, chunk 2j + tid / 128: input row ky = j
    }
    const __amdgpu_buffer_rsrc_t c3r = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, C3 ? g3.bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t c3d = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, C3 ? (uint32_t)(M * 64) : 0u, 0x00020000);
#pragma unroll
    for (int j = 0; j < DSN; ++j) { int i = tid + j * PWB; dsl[j] = i < PW_P * dc ? ((i / dc) << 8) | (i % dc) : -1; }
    uint4 px[PW_XS], pd[DSN];
    auto prefetch = [&](int64_t tile) {
        const int64_t m0 = tile * PW_P;
        if (C3) {       // px[ky] = input pixels (ky, kx = 0,1) for tid < 128, (ky, kx = 2) + zeros for tid >= 128, of output pixel m0 + tid % 128
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
        } else
#pragma unroll
        for (int j = 0; j < PW_XS; ++j) {
            px[j] = make_uint4(0, 0, 0, 0);
            if (xsl[j] >= 0 && m0 + (xsl[j] >> 8) < M)
                px[j] = *reinterpret_cast<const uint4*>(xsrc + (m0 + (xsl[j] >> 8)) * ldx + (xsl[j] & 255) * 8);
        }
#pragma unroll
        for (int j = 0; j < DSN; ++j) {
            if (C3) {           // the 128 x 64 B tile of dy is one contiguous span: no branch, rows >= M read as zeros through the descriptor
                pd[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(c3d, (uint32_t)m0 * 64u + (uint32_t)(tid + j * PWB) * 16u, 0, 0));
                continue;
            }
            pd[j] = make_uint4(0, 0, 0, 0);
            if (dsl[j] >= 0 && m0 + (dsl[j] >> 8) < M)
                pd[j] = *reinterpret_cast<const uint4*>(dy + (m0 + (dsl[j] >> 8)) * ldy + (dsl[j] & 255) * 8);
        }
    };
    // per-lane transposing-read bases (see conv_mfma.hip): address = base + pixel*stride (+ 4*stride for the second half)
    const int li = lane & 15, lq = li >> 2, lpp = li & 3, lg = lane >> 4;
    const int lrow = 8 * (lg >> 1) + lq, lcol = (16 * (lg & 1) + 4 * lpp) * 2;
    const unsigned char* lbX = sX + lrow * SX + lcol;
    const unsigned char* lbD = sD + lrow * SD + lcol;
    auto tr2 = [&](const unsigned char* p, int stride) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * stride));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    struct WF { bf16x8 a[NT], b[KTB]; };
    auto load_chunk = [&](WF& f, int ch) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) f.a[ct] = tr2(lbD + ch * 16 * SD + 64 * ct, SD);
#pragma unroll
        for (int kt = 0; kt < KTB; ++kt) f.b[kt] = tr2(lbX + ch * 16 * SX + 64 * kt, SX);
    };
    auto mma_chunk = [&](const WF& f) {
        if (blockIdx.y == 0) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int j = 0; j < 8; ++j) bsum[ct] += (float)f.a[ct][j];
        }
#pragma unroll
        for (int kt = 0; kt < KTB; ++kt)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[ct][kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[ct], f.b[kt], acc[ct][kt], 0, 0, 0);
    };
    const int64_t tiles = (M + PW_P - 1) / PW_P;
    int64_t tile = blockIdx.x;
    if (C3)         // patch elements 48..63 are never gathered: zero once
        for (int i = tid; i < PW_P * 2; i += PWB) *reinterpret_cast<uint4*>(sX + (i >> 1) * SX + (6 + (i & 1)) * 16) = make_uint4(0, 0, 0, 0);
    if (tile < tiles) prefetch(tile);
    for (; tile < tiles; tile += gridDim.x) {
        __syncthreads();
        if (C3) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if (xsl[j] >= 0) *reinterpret_cast<uint4*>(sX + (tid & 127) * SX + (xsl[j] & 255) * 16) = px[j];
        } else
#pragma unroll
        for (int j = 0; j < PW_XS; ++j)
            if (xsl[j] >= 0) *reinterpret_cast<uint4*>(sX + (xsl[j] >> 8) * SX + (xsl[j] & 255) * 16) = px[j];
#pragma unroll
        for (int j = 0; j < DSN; ++j)
            if (dsl[j] >= 0) *reinterpret_cast<uint4*>(sD + (dsl[j] >> 8) * SD + (dsl[j] & 255) * 16) = pd[j];
        __syncthreads();
        if (C3 || tile + gridDim.x < tiles) prefetch(tile + gridDim.x);        // C3: branch-free loads, a tile past the end reads zeros
        WF f0, f1;
        load_chunk(f0, wave);
        load_chunk(f1, wave + 4);
        __builtin_amdgcn_sched_barrier(0);
        mma_chunk(f0);
        mma_chunk(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the four waves hold partial sums of the same tiles: they take turns adding them into LDS (plain read-add-write, one barrier
    // per turn) -- LDS float atomics here cost ~20 us per block (see k_conv32_wgrad)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [NT*32][KTB*32]
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int ct = 0; ct < NT; ++ct)
#pragma unroll
                for (int kt = 0; kt < KTB; ++kt)
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int co = ct * 32 + (k & 3) + 8 * (k >> 2) + 4 * hh;
                        float* dst = &red[co * (KTB * 32) + kt * 32 + r];
                        *dst = turn == 0 ? acc[ct][kt][k] : *dst + acc[ct][kt][k];
                    }
        }
        __syncthreads();
    }
    for (int i = tid; i < NT * 32 * KTB * 32; i += PWB) {
        const int co = i / (KTB * 32), cl = i - co * (KTB * 32);
        if (C3) {       // patch element 16*ky + 4*kx + ch -> w[co][ch][ky][kx]
            if (cl < 48 && ((cl >> 2) & 3) < 3 && (cl & 3) < 3) atomicAdd(&dw[co * 27 + (cl & 3) * 9 + (cl >> 4) * 3 + ((cl >> 2) & 3)], red[i]);
        }
        else
        if (co < N && ci_base + cl < K) atomicAdd(&dw[(int64_t)co * K + ci_base + cl], red[i]);
    }
    if (dbias && blockIdx.y == 0) {
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
            float s = bsum[ct] + __shfl_xor(bsum[ct], 32, 64);
            if (lane < 32 && ct * 32 + r < N) atomicAdd(&dbias[ct * 32 + r], s);
        }
    }
}

/* x bf16 [M,K], dy bf16 [M,N] -> dw fp32 [N,K] and dbias fp32 [N] (nullable); both overwritten.  K, N multiples of 32, N <= 160 */
static int pw_wgrad_impl(const void* x, const void* x2, int K1, const void* dy, float* dw, float* dbias, int64_t M, int K, int N, tcct_stream_t stream,
                         int64_t ldy);
extern "C" int tcct_pw_wgrad(const void* x, const void* dy, float* dw, float* dbias, int64_t M, int K, int N, tcct_stream_t stream) {
    return pw_wgrad_impl(x, nullptr, 0, dy, dw, dbias, M, K, N, stream, 0);
}
/* weight gradient with x = [x1 | x2] given as two tensors (x1 [M,K1], x2 [M,K-K1]); dw [N,K] as for the concatenated input */
extern "C" int tcct_pw_wgrad_cat2(const void* x1, const void* x2, int K1, const void* dy, float* dw, float* dbias, int64_t M, int K, int N,
                                  tcct_stream_t stream) {
    TCCT_CHECK(x2 != nullptr && K1 % 32 == 0 && K1 > 0 && K1 < K, "pw_wgrad_cat2: K1=%d must be a multiple of 32 inside (0, K)", K1);
    return pw_wgrad_impl(x1, x2, K1, dy, dw, dbias, M, K, N, stream, 0);
}
/* weight gradient of an N-column slab of a wider GEMM: dy rows have stride ldy (elements, multiple of 8, dy points at the slab's first
 * column); dw / dbias point at the slab's first row.  Lets outputs wider than 160 (the qkv Linear of the factorised attention, N = 3C) run
 * as slabs on the same MFMA kernel. */
extern "C" int tcct_pw_wgrad_strided(const void* x, const void* dy, int64_t ldy, float* dw, float* dbias, int64_t M, int K, int N,
                                     tcct_stream_t stream) {
    return pw_wgrad_impl(x, nullptr, 0, dy, dw, dbias, M, K, N, stream, ldy);
}
static int pw_wgrad_impl(const void* x, const void* x2, int K1, const void* dy, float* dw, float* dbias, int64_t M, int K, int N, tcct_stream_t stream,
                         int64_t ldy) {
    if (ldy == 0) ldy = N;
    TCCT_CHECK(ldy >= N && ldy % 8 == 0, "pw_wgrad: ldy=%lld must be a multiple of 8 and >= N=%d", (long long)ldy, N);
    TCCT_CHECK(K % 32 == 0 && N % 32 == 0 && K >= 32 && N >= 32 && N <= 160 && K <= 1024, "pw_wgrad: unsupported K=%d N=%d", K, N);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)N * K, st) != hipSuccess) { tcct_set_error("pw_wgrad: memset failed"); return -2; }
    if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * N, st) != hipSuccess) { tcct_set_error("pw_wgrad: memset failed"); return -2; }
    const int NT = N / 32, ktiles = K / 32;
    const int KTB = (ktiles % 2 == 0 && NT <= 2) ? 2 : 1;       // <= 5 accumulators per wave next to the prefetch registers (round 6: two slabs at 96 / 128 outputs -- to re-read dy
                                                                // half as often: stage 1's `aggregate` reads its 113 MB dy six times -- spill 22 / 139 VGPRs at two blocks per CU: not offered)
    const int gy = ktiles / KTB;
    TCCT_CHECK(!x2 || K1 % (KTB * 32) == 0, "pw_wgrad_cat2: K1=%d must be a multiple of %d for this shape", K1, KTB * 32);
    // row strides: S mod 256 in {64,192} keeps the 4-pixel x 64-byte footprint of a transposing read on distinct banks
    const int SX = KTB == 1 ? 64 : 192;
    const int SD = 2 * N + ((NT & 1) ? 0 : 64);
    size_t lds = (size_t)PW_P * (SX + SD);
    size_t red = (size_t)NT * 32 * KTB * 32 * 4;
    if (red > lds) lds = red;
    const int64_t tiles = (M + PW_P - 1) / PW_P;
    (void)hipFuncSetAttribute;
    int per_cu = (int)((160 * 1024) / (lds + 512));
    if (per_cu > 2) per_cu = 2;                                   // __launch_bounds__(256, 2)
    if (per_cu < 1) per_cu = 1;
    int gx = 256 * per_cu / gy;
    if (gx < 64) gx = 64;
    // every block ends with NT x KTB x 1024 fp32 atomics on the same addresses: on small maps (<= 1024 tiles of 128 pixels, levels 3-4) fewer blocks walk more tiles each --
    // 160 -> 160 at level 4 ran 510 blocks for 216 tiles and 2.6 M atomics for a 10 MB problem (round 6)
    if (tiles <= 1024) { int small = 128 / gy; if (small < 16) small = 16; if (gx > small) gx = small; }
    if (gx > tiles) gx = (int)tiles;
    if (gx < 1) gx = 1;
#define WL(NTV, KV) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_wgrad<NTV, KV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } } hipLaunchKernelGGL((k_pw_wgrad<NTV, KV>), dim3(gx, gy), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)dy, dw, dbias, M, K, N, SX, SD, (const bf16*)x2, K1, ldy)
    if (KTB == 2) { switch (NT) { case 1: WL(1, 2); break; default: WL(2, 2); break; } }
    else { switch (NT) { case 1: WL(1, 1); break; case 2: WL(2, 1); break; case 3: WL(3, 1); break; case 4: WL(4, 1); break; default: WL(5, 1); break; } }
#undef WL
    TCCT_LAUNCH_OK();
}


/* weight / bias gradient of tcct_c3_fwd: dw fp32 [32,3,3,3] and dbias fp32 [32] (nullable), both overwritten (left to the caller's zero
 * pool when that is active); x4 the 4-channel bf16 image, dy bf16 [B,Ho,Wo,32]. */
extern "C" int tcct_c3_wgrad(const void* x4, const void* dy, float* dw, float* dbias, int B, int H, int W, int stride, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "c3_wgrad: stride %d", stride);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int64_t M = (int64_t)B * Ho * Wo, inb = (int64_t)B * H * W * 8;
    TCCT_CHECK(M > 0 && M * 64 < (1ll << 31) && inb < (1ll << 31), "c3_wgrad: image too large for 32-bit byte offsets (B=%d H=%d W=%d)", B, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * 32 * 27, st) != hipSuccess) { tcct_set_error("c3_wgrad: memset failed"); return -2; }
    if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * 32, st) != hipSuccess) { tcct_set_error("c3_wgrad: memset failed"); return -2; }
    const int SX = 192, SD = 64;
    const size_t lds = (size_t)PW_P * (SX + SD);
    const int64_t tiles = (M + PW_P - 1) / PW_P;
    // every block ends with 864 atomics on the same 864 addresses, which serialise: >= 48 tiles per block, between one and four blocks per CU
    // (bs 8, 800x1104: stride 1 0.238 / 0.162 / 0.145 ms with 256 / 512 / 1024 blocks; stride 2 0.059 / 0.059 / 0.083 ms)
    int gx = (int)(tiles / 48 < 256 ? 256 : (tiles / 48 > 1024 ? 1024 : tiles / 48));
    if (gx > tiles) gx = (int)tiles;
    auto magic = [](uint32_t d) { return d == 1 ? 0xffffffffu : (uint32_t)((1ull << 32) / d); };
    hipLaunchKernelGGL((k_pw_wgrad<1, 2, true>), dim3(gx, 1), dim3(PWB), lds, st, (const bf16*)x4, (const bf16*)dy, dw, dbias, M, 64, 32, SX, SD,
                       (const bf16*)nullptr, 0, (int64_t)32,
                       C3Geom{H, W, Ho, Wo, stride, (uint32_t)inb, (Wo + 31) / 32, magic((uint32_t)((Wo + 31) / 32)), magic((uint32_t)Ho), magic((uint32_t)(Ho * Wo)), magic((uint32_t)Wo)});
    TCCT_LAUNCH_OK();
}


// ------------------------------------------------------------------------------------------------ fused backward
// Input gradient AND weight gradient of a 1x1 convolution / Linear in ONE pass over dy (the separate kernels read dy twice -- the
// weight-gradient kernel up to K/32 times for wide shapes, its blockIdx.y slices -- and x once):
//     dx[M,K] = dy[M,N] W[N,K] (+ res),   dW[N,K] += dy^T x,   dbias[N] += sum_m dy.
// Per 128-pixel tile the x and dy rows are staged in LDS once (register prefetch one tile ahead, out-of-range rows read as zeros
// through buffer descriptors).  Weight gradient: the NT x KT output tiles are OWNED by waves (tile i -> wave i % 4, K = the 128
// pixels as eight transposing-read chunks), so no cross-wave reduction and every dy element is read from HBM exactly once.  Input
// gradient: wave w takes pixels 32w..32w+31, A = W^T from LDS (bf16 [K][N], conflict-free rows of 2N+16 bytes), B = the staged dy
// rows.  Same tile pipeline as k_conv32_mfma: the staging wait sits behind the MFMA phase, the dx stores of tile t are issued after
// the loads of tile t+2.  Algorithmic bytes: M (2K + N) 2 B against M (2K + 2N) 2 B (or more) for the two-kernel form.
#define PB_P 128

// GX (round 4): the operand tensor holds the PRE-activation y1 of Mlp.fc1 (reference nets/tcct.py:29-53) and the kernel applies GELU while it stages the
// tile -- h = bf16(gelu(y1)), bit-identical to the tensor the separate activation pass used to write (same gauss_cdf_pdf form, same rounding) -- so
// neither h nor, in the backward pass, dh ever exist in HBM: the activation pass (read y1, write h) and its backward (read dh, read y1, write dy1) are gone.
__device__ __forceinline__ u32x4 gelu8(u32x4 v) {
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = pack_bf16x2(act_c<TCCT_ACT_GELU>(__uint_as_float(v[k] << 16)), act_c<TCCT_ACT_GELU>(__uint_as_float(v[k] & 0xffff0000u)));
    return o;
}
__device__ __forceinline__ float gelu_grad1(float x) { float c, p; gauss_cdf_pdf(x, c, p); return c + x * p; }
// XAP >= 0 (round 4): the operand tensor holds the INPUT y_prev of a train-mode BatchNorm (+ activation XAP) whose output this convolution consumes as
// its only reader (InvRes.norm -> conv2, reference nets/tcct.py:563-572): z = act(a[c] y_prev + b[c]) is applied while the tile is staged -- rounded to
// bf16 exactly like the tensor the separate normalisation pass used to write -- so that pass (read y_prev, write z) and the tensor z do not exist.
template <int KIND> __device__ __forceinline__ u32x4 affine8(u32x4 v, const float* __restrict__ a, const float* __restrict__ b) {
    const float4 a0 = *reinterpret_cast<const float4*>(a), a1 = *reinterpret_cast<const float4*>(a + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(b), b1 = *reinterpret_cast<const float4*>(b + 4);
    const float aa[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    u32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        o[k] = pack_bf16x2(act_c<KIND>(aa[2 * k] * __uint_as_float(v[k] << 16) + bb[2 * k]),
                           act_c<KIND>(aa[2 * k + 1] * __uint_as_float(v[k] & 0xffff0000u) + bb[2 * k + 1]));
    return o;
}
// SPLIT: x = [x | x2] and dx = [dx | dx2] are two tensors of K/2 channels each (the aggregate convolution over a concatenation,
// MHCA_stage, reference nets/tcct.py:600-616): no concat / split passes; x2 / dx2 then travel in the res / dx_plain arguments.
// BNP >= 0: a train-mode BatchNorm sits behind this convolution (Conv2d_BN, DWConv2d_BN.pwconv, tran_*; reference nets/tcct.py:55-97,124-126,
// 966-974) and `dy` is the gradient of the BatchNorm OUTPUT z = post(a y + b): the gradient of the convolution output is rebuilt while the
// tile is staged, dy_conv = c1 dz' + c2 y + c3 with dz' = dz post'(a y + b) (the per-channel constants carry the two batch sums of the
// BatchNorm backward, tcct_bn_bwd_coef) -- the separate BatchNorm backward apply pass (read dz, read y, write dy) is gone, this kernel reads
// y in its place.  BNP = post activation kind (0 none, 2 hardswish).
// REDP >= 0: the INPUT x is itself the output of a train-mode BatchNorm z_prev = post_prev(ap y_prev + bp) whose only remaining gradient is
// this kernel's dx (+ res): the two batch sums of THAT BatchNorm's backward (sum dz', sum dz' y_prev per channel) are accumulated in the dx
// epilogue, so its separate reduction pass (read dz, read y_prev) is gone as well.  REDP = post_prev kind (0 / 2).
// coef == NULL: the per-channel constants are derived in the kernel's prologue from the two batch sums (what tcct_bn_bwd_coef computes; one launch
// fewer per BatchNorm), and block 0 writes dgamma / dbeta
struct PwBnBwd { const bf16* y; const float* coef; const bf16* yprev; const float* abprev; double* sums_prev;
                 const double* sums; int raw; const float* mean_rstd; const float* ab; float* dgamma; float* dbeta; };
template <int KIND> __device__ __forceinline__ float pw_act_grad(float u) {
    if (KIND == TCCT_ACT_HSWISH) return u < -3.f ? 0.f : (u <= 3.f ? (2.f * u + 3.f) * (1.f / 6.f) : 1.f);
    return 1.f;
}
// LNB (round 4): x is the OUTPUT of a LayerNorm over the K channels of each row (MHCABlock.norm2 -> Mlp.fc1, reference nets/tcct.py:466-468) that only this
// convolution reads.  The dx epilogue then runs that LayerNorm's backward on the spot -- dx as it would have been stored (bf16) is d; per row
// g = d gamma, s1 = mean_c g, s2 = mean_c g xh, out = rstd (g - s1 - xh s2) + res with xh = (t - mean) rstd rebuilt from the LayerNorm's INPUT t (bn.yprev) and
// its saved statistics (bn.mean_rstd [M][2]) -- and writes the gradient of t (+ res, the gradient that reaches t through the residual path); dgamma / dbeta
// (bn.dgamma / bn.dbeta, fp32 [K], zero on entry) are accumulated.  The separate LayerNorm backward pass (read t, read d, read res, write) is gone.
// DY2 (round 4): dy = dy + dy2 (bn.y carries dy2), rounded to bf16 as the tensor a separate add pass used to write -- the decoder block's two output gradients
// (MPUpBlock's own output and `x_i + y_i`, reference nets/tcct.py:908-914,1028-1031) are summed while the tile is staged
template <int NT, int KT, bool SPLIT = false, int BNP = -1, int REDP = -1, bool GX = false, int XAP = -1, bool LNB = false, bool DY2 = false>
__global__ void __launch_bounds__(PWB, 2)
k_pw_bwd(const bf16* __restrict__ x, const bf16* __restrict__ dy, const float* __restrict__ w, const bf16* __restrict__ res,
         bf16* __restrict__ dx, bf16* __restrict__ dx_plain, float* __restrict__ dw, float* __restrict__ dbias, int64_t M, PwBnBwd bn) {
    constexpr int K = 32 * KT, N = 32 * NT;
    constexpr bool BN = BNP >= 0, RED = REDP >= 0;
    constexpr int SX = 64 * KT + ((KT & 1) ? 0 : 64), SD = 64 * NT + ((NT & 1) ? 0 : 64);      // rows == 64 / 192 mod 256: conflict-free transposing reads
    constexpr int SW = 2 * N + 16;
    constexpr int XS = K / 16, DS = N / 16;                // 16-byte staging slots per thread (128 px x K/8 chunks / 256 threads)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sD = sX + PB_P * SX;
    unsigned char* sW = sD + PB_P * SD;
    unsigned char* sS = sW + ((K * SW + 15) & ~15);        // per-wave epilogue transpose scratch: 32 px x 80 B
    float* sC = reinterpret_cast<float*>(sS + 4 * 2560);   // BN: [5][N] = c1, c2, c3, a, b;  RED: [2][K] = ap, bp behind it
    float* sP = sC + (BN ? 5 * N : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    // W^T as bf16 [k][n]: w is [N][K] fp32; consecutive threads read consecutive k of one n (coalesced), transposed on the LDS side
    for (int i = tid; i < N * K; i += PWB) {
        const int n = i / K, k = i - n * K;
        *reinterpret_cast<bf16*>(sW + k * SW + n * 2) = __float2bfloat16(w[i]);
    }
    if (BN) {
        if (bn.coef) { for (int i = tid; i < 5 * N; i += PWB) sC[i] = bn.coef[i]; }
        else if (tid < N) {
            const int c = tid;
            const double mu = bn.mean_rstd[c], rs = bn.mean_rstd[N + c];
            const double S1 = bn.sums[c];
            const double S2 = bn.raw ? (bn.sums[N + c] - mu * S1) * rs : bn.sums[N + c];
            if (blockIdx.x == 0) { bn.dbeta[c] = (float)S1; bn.dgamma[c] = (float)S2; }
            const double a = bn.ab[c], s1 = S1 / (double)M, s2 = S2 / (double)M;
            sC[c] = (float)a; sC[N + c] = (float)(-a * s2 * rs); sC[2 * N + c] = (float)(a * (s2 * rs * mu - s1));
            sC[3 * N + c] = bn.ab[c]; sC[4 * N + c] = bn.ab[N + c];
        }
    }
    if (RED || XAP >= 0) for (int i = tid; i < 2 * (SPLIT ? K / 2 : K); i += PWB) sP[i] = bn.abprev[i];
    if (LNB) for (int i = tid; i < K; i += PWB) sP[i] = bn.abprev[i];          // gamma[K] of the LayerNorm in front
    constexpr int RKT = LNB ? KT : (RED ? (SPLIT ? KT / 2 : KT) : 1);   // 32-channel tiles of x that carry the BatchNorm in front (split: the first half)
    float rs[RKT][8], rq[RKT][8];                            // RED: per-lane partial sums of dz' and dz' y_prev (channels 8 (lane & 3) + k of tile kt)
#pragma unroll
    for (int a = 0; a < RKT; ++a)
#pragma unroll
        for (int k = 0; k < 8; ++k) rs[a][k] = rq[a][k] = 0.f;
    f32x16 accw[(NT * KT + 3) / 4];
#pragma unroll
    for (int a = 0; a < (NT * KT + 3) / 4; ++a)
#pragma unroll
        for (int k = 0; k < 16; ++k) accw[a][k] = 0.f;
    float bsum[(NT * KT + 3) / 4];                   // bias gradient: per owned tile with kt == 0
#pragma unroll
    for (int a = 0; a < (NT * KT + 3) / 4; ++a) bsum[a] = 0.f;
    const int64_t tiles = (M + PB_P - 1) / PB_P;
    constexpr int KS = SPLIT ? K / 2 : K;                   // channels per x / dx tensor
    const uint32_t xbytes = (uint32_t)(M * KS * 2), dbytes = (uint32_t)(M * N * 2);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)(res ? res : x), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)dx, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(dx_plain ? dx_plain : dx), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)((BN || DY2) ? bn.y : dy), 0, dbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ryp = __builtin_amdgcn_make_buffer_rsrc((void*)((RED || LNB) ? bn.yprev : x), 0, xbytes, 0x00020000);
    u32x4 px[XS], pd[DS], py[(BN || DY2) ? DS : 1];
    // a tile's rows are one contiguous span of memory: slot i of a thread = bytes [16 (tid + 256 i), +16) of it (fully coalesced);
    // rows beyond M fall outside the descriptor and read as zeros
    auto prefetch = [&](int64_t tile) {
        const uint32_t bx = (uint32_t)(tile * PB_P * KS * 2), bd = (uint32_t)(tile * PB_P * N * 2);
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            if (SPLIT && j >= XS / 2) px[j] = __builtin_amdgcn_raw_buffer_load_b128(rr, bx + (uint32_t)(tid + (j - XS / 2) * PWB) * 16u, 0, 0);      // x2
            else px[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, bx + (uint32_t)(tid + j * PWB) * 16u, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < DS; ++j) pd[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, bd + (uint32_t)(tid + j * PWB) * 16u, 0, 0);
        if (BN || DY2) {
#pragma unroll
            for (int j = 0; j < DS; ++j) py[j] = __builtin_amdgcn_raw_buffer_load_b128(ry, bd + (uint32_t)(tid + j * PWB) * 16u, 0, 0);
        }
    };
    auto stage_x1 = [&](int j, u32x4 xv_) {
        const int jj = (SPLIT && j >= XS / 2) ? j - XS / 2 : j;
        const int q = tid + jj * PWB, p = q / (KS / 8), c = q - p * (KS / 8);
        if (GX) xv_ = gelu8(xv_);
        if (XAP >= 0) xv_ = affine8<XAP < 0 ? 0 : XAP>(xv_, sP + c * 8, sP + KS + c * 8);       // x = act(a y_prev + b) (sP: a[K], b[K])
        *reinterpret_cast<u32x4*>(sX + p * SX + ((SPLIT && j >= XS / 2) ? KS * 2 : 0) + c * 16) = xv_;
    };
    auto stage_d = [&](int64_t tile) {      // tile: the tile whose registers are being staged (rows >= M of it must give dy = 0)
#pragma unroll
        for (int j = 0; j < DS; ++j) {
            const int q = tid + j * PWB, p = q / (N / 8), c = q - p * (N / 8);
            u32x4 v = pd[j];
            if (DY2) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    v[k] = pack_bf16x2(__uint_as_float(pd[j][k] << 16) + __uint_as_float(py[j][k] << 16),
                                       __uint_as_float(pd[j][k] & 0xffff0000u) + __uint_as_float(py[j][k] & 0xffff0000u));
            }
            if (BN) {
                // dy_conv = c1 dz post'(a y + b) + c2 y + c3, rounded to bf16 like the tensor the separate apply pass used to write
                const bool live = tile * PB_P + p < M;
                const float* cc = sC + c * 8;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 c1 = *reinterpret_cast<const float4*>(cc + 4 * h), c2 = *reinterpret_cast<const float4*>(cc + N + 4 * h);
                    const float4 c3 = *reinterpret_cast<const float4*>(cc + 2 * N + 4 * h);
                    float4 ca = c1, cb = c1;
                    if (BNP != TCCT_ACT_NONE) { ca = *reinterpret_cast<const float4*>(cc + 3 * N + 4 * h); cb = *reinterpret_cast<const float4*>(cc + 4 * N + 4 * h); }
                    const float k1[4] = {c1.x, c1.y, c1.z, c1.w}, k2[4] = {c2.x, c2.y, c2.z, c2.w}, k3[4] = {c3.x, c3.y, c3.z, c3.w};
                    const float ka[4] = {ca.x, ca.y, ca.z, ca.w}, kb[4] = {cb.x, cb.y, cb.z, cb.w};
#pragma unroll
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const uint32_t gz = pd[j][2 * h + e2], gy = py[j][2 * h + e2];
                        float z0 = __uint_as_float(gz << 16), z1 = __uint_as_float(gz & 0xffff0000u);
                        const float y0 = __uint_as_float(gy << 16), y1 = __uint_as_float(gy & 0xffff0000u);
                        if (BNP != TCCT_ACT_NONE) {
                            z0 *= pw_act_grad<BNP>(ka[2 * e2] * y0 + kb[2 * e2]);
                            z1 *= pw_act_grad<BNP>(ka[2 * e2 + 1] * y1 + kb[2 * e2 + 1]);
                        }
                        const float o0 = k1[2 * e2] * z0 + k2[2 * e2] * y0 + k3[2 * e2];
                        const float o1 = k1[2 * e2 + 1] * z1 + k2[2 * e2 + 1] * y1 + k3[2 * e2 + 1];
                        v[2 * h + e2] = live ? pack_bf16x2(o0, o1) : 0u;
                    }
                }
            }
            *reinterpret_cast<u32x4*>(sD + p * SD + c * 16) = v;
        }
    };
    auto stage = [&](int64_t tile) {
#pragma unroll
        for (int j = 0; j < XS; ++j) stage_x1(j, px[j]);
        stage_d(tile);
    };
    // transposing-read lane bases (see k_pw_wgrad)
    const int li = lane & 15, lq = li >> 2, lpp = li & 3, lg = lane >> 4;
    const int lrow = 8 * (lg >> 1) + lq, lcol = (16 * (lg & 1) + 4 * lpp) * 2;
    const unsigned char* lbX = sX + lrow * SX + lcol;
    const unsigned char* lbD = sD + lrow * SD + lcol;
    auto tr2 = [&](const unsigned char* p, int stride) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * stride));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    int64_t tile = blockIdx.x;
    if (tile < tiles) {
        prefetch(tile);
        __syncthreads();            // weights staged
        stage(tile);
        prefetch(tile + gridDim.x);                 // beyond the last tile: every row out of range, zeros, never used
    }
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
        u32x4 xq[GX ? KT : 1][2];       // GX: the raw x (= y1) values the dx epilogue multiplies gelu'(y1) from, requested before the MFMA phase (L2 / MALL hits:
        if (GX) {                       // the tile was staged from the same addresses two iterations ago)
            const int64_t m0q = tile * PB_P + 32 * wave;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int64_t mm = m0q + (lane >> 2) + 16 * h2;
                    xq[kt][h2] = __builtin_amdgcn_raw_buffer_load_b128(rx, mm < M ? (uint32_t)((mm * K + kt * 32 + (lane & 3) * 8) * 2) : 0x80000000u, 0, 0);
                }
        }
        u32x4 tq[LNB ? KT : 1][2];      // LNB: the LayerNorm input rows (and their statistics) the dx epilogue of this tile needs
        float tmean[2] = {0.f, 0.f}, trstd[2] = {0.f, 0.f};
        if (LNB) {
            const int64_t m0q = tile * PB_P + 32 * wave;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int64_t mm = m0q + (lane >> 2) + 16 * h2;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt)
                    tq[kt][h2] = __builtin_amdgcn_raw_buffer_load_b128(ryp, mm < M ? (uint32_t)((mm * K + kt * 32 + (lane & 3) * 8) * 2) : 0x80000000u, 0, 0);
                if (mm < M) { const float2 mr = *reinterpret_cast<const float2*>(bn.mean_rstd + 2 * mm); tmean[h2] = mr.x; trstd[h2] = mr.y; }
            }
        }
        u32x4 yq[RKT][2];               // RED: the y_prev values the dx epilogue of this tile needs, requested before the MFMA phase
        if (RED) {
            const int64_t m0q = tile * PB_P + 32 * wave;
#pragma unroll
            for (int kt = 0; kt < RKT; ++kt)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int64_t mm = m0q + (lane >> 2) + 16 * h2;
                    yq[kt][h2] = __builtin_amdgcn_raw_buffer_load_b128(ryp, mm < M ? (uint32_t)((mm * KS + kt * 32 + (lane & 3) * 8) * 2) : 0x80000000u, 0, 0);
                }
        }
        // ---- weight gradient: owned tiles, K-dim = the 128 staged pixels
#pragma unroll
        for (int a = 0; a < (NT * KT + 3) / 4; ++a) {
            const int ti = wave + 4 * a;
            if (ti < NT * KT) {                       // wave-uniform
                const int ct = ti / KT, kt = ti - ct * KT;
#pragma unroll
                for (int ch = 0; ch < 8; ++ch) {
                    const bf16x8 fa = tr2(lbD + ch * 16 * SD + 64 * ct, SD);
                    const bf16x8 fb = tr2(lbX + ch * 16 * SX + 64 * kt, SX);
                    if (kt == 0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) bsum[a] += (float)fa[j];
                    }
                    accw[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, accw[a], 0, 0, 0);
                }
            }
        }
        // ---- input gradient of this wave's 32 pixels: D[k][pixel] = sum_n W^T[k][n] dy[pixel][n]
        f32x16 accd[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int k = 0; k < 16; ++k) accd[kt][k] = 0.f;
        const unsigned char* bB = sD + (32 * wave + r) * SD + hh * 16;
#pragma unroll
        for (int st = 0; st < 2 * NT; ++st) {
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(bB + st * 32);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(sW + (kt * 32 + r) * SW + st * 32 + hh * 16);
                accd[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, accd[kt], 0, 0, 0);
            }
        }
        __syncthreads();                              // every wave has read the staged tile
        if ((GX || XAP >= 0) && !SPLIT) {
            // on-load transform of the x tile (k_pw_fwd2): slot by slot -- copy the arrived registers, re-issue the slot's load for tile t + 2, then
            // transform and store -- so that ~1000 cycles of VALU do not sit between this tile's last use and the next loads
            const bool do_stage = tile + (int64_t)gridDim.x < tiles;
            const int64_t tp = tile + 2 * (int64_t)gridDim.x;
            const uint32_t bx = (uint32_t)(tp * PB_P * KS * 2), bd = (uint32_t)(tp * PB_P * N * 2);
#pragma unroll
            for (int j = 0; j < XS; ++j) {
                const u32x4 xv_ = px[j];
                px[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, bx + (uint32_t)(tid + j * PWB) * 16u, 0, 0);
                if (do_stage) stage_x1(j, xv_);
            }
            if (do_stage) stage_d(tile + (int64_t)gridDim.x);
#pragma unroll
            for (int j = 0; j < DS; ++j) pd[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, bd + (uint32_t)(tid + j * PWB) * 16u, 0, 0);
            if (BN) {
#pragma unroll
                for (int j = 0; j < DS; ++j) py[j] = __builtin_amdgcn_raw_buffer_load_b128(ry, bd + (uint32_t)(tid + j * PWB) * 16u, 0, 0);
            }
        } else {
            if (tile + (int64_t)gridDim.x < tiles) stage(tile + (int64_t)gridDim.x);
            prefetch(tile + 2 * (int64_t)gridDim.x);
        }
        // ---- dx epilogue: per-wave LDS transpose -> 16-byte stores (16 whole 64-byte pixel segments per wave instruction)
        unsigned char* sc = sS + wave * 2560;
        const int64_t m0 = tile * PB_P + 32 * wave;
        if (LNB) {
            u32x4 ov[KT][2];            // d = dx as stored, transposed: lane = (row p, 8 channels) of every 32-channel tile
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 o;
                    o.x = pack_bf16x2(accd[kt][4 * q], accd[kt][4 * q + 1]);
                    o.y = pack_bf16x2(accd[kt][4 * q + 2], accd[kt][4 * q + 3]);
                    *reinterpret_cast<uint2*>(sc + r * 80 + (8 * q + 4 * hh) * 2) = o;
                }
                wave_lds_fence();
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) ov[kt][h2] = *reinterpret_cast<const u32x4*>(sc + ((lane >> 2) + 16 * h2) * 80 + (lane & 3) * 16);
                wave_lds_fence();
            }
            const int cch = lane & 3;
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int64_t mm = m0 + (lane >> 2) + 16 * h2;
                const float mean = tmean[h2], rstd = trstd[h2];
                float g[KT][8], xh[KT][8], s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const u32x4 tv = tq[kt][h2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float d0 = __uint_as_float(ov[kt][h2][k] << 16), d1 = __uint_as_float(ov[kt][h2][k] & 0xffff0000u);
                        const float h0 = (__uint_as_float(tv[k] << 16) - mean) * rstd, h1 = (__uint_as_float(tv[k] & 0xffff0000u) - mean) * rstd;
                        const float g0 = d0 * sP[kt * 32 + cch * 8 + 2 * k], g1 = d1 * sP[kt * 32 + cch * 8 + 2 * k + 1];
                        g[kt][2 * k] = g0; g[kt][2 * k + 1] = g1; xh[kt][2 * k] = h0; xh[kt][2 * k + 1] = h1;
                        s1 += g0 + g1; s2 += g0 * h0 + g1 * h1;
                        rs[kt][2 * k] += d0 * h0; rs[kt][2 * k + 1] += d1 * h1;         // dgamma
                        rq[kt][2 * k] += d0; rq[kt][2 * k + 1] += d1;                   // dbeta
                    }
                }
                s1 += __shfl_xor(s1, 1, 64); s2 += __shfl_xor(s2, 1, 64);               // the four lanes of a row
                s1 += __shfl_xor(s1, 2, 64); s2 += __shfl_xor(s2, 2, 64);
                s1 *= 1.f / (float)K; s2 *= 1.f / (float)K;
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    const uint32_t off = mm < M ? (uint32_t)((mm * K + kt * 32 + cch * 8) * 2) : 0x80000000u;
                    const u32x4 rv = __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0);
                    u32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k] = pack_bf16x2(rstd * (g[kt][2 * k] - s1 - xh[kt][2 * k] * s2) + __uint_as_float(rv[k] << 16),
                                           rstd * (g[kt][2 * k + 1] - s1 - xh[kt][2 * k + 1] * s2) + __uint_as_float(rv[k] & 0xffff0000u));
                    __builtin_amdgcn_raw_buffer_store_b128(o, ro, off, 0, 0);
                }
            }
        } else
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 o;
                o.x = pack_bf16x2(accd[kt][4 * q], accd[kt][4 * q + 1]);
                o.y = pack_bf16x2(accd[kt][4 * q + 2], accd[kt][4 * q + 3]);
                *reinterpret_cast<uint2*>(sc + r * 80 + (8 * q + 4 * hh) * 2) = o;
            }
            wave_lds_fence();
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int p = (lane >> 2) + 16 * h2, cch = lane & 3;
                u32x4 o = *reinterpret_cast<const u32x4*>(sc + p * 80 + cch * 16);
                const int64_t mm = m0 + p;
                auto red_acc = [&](const u32x4& ov, int kch) {     // kch: first channel of the lane's 8 inside x / y_prev rows
                    const u32x4 yv = yq[kt < RKT ? kt : 0][h2];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float d0 = __uint_as_float(ov[k] << 16), d1 = __uint_as_float(ov[k] & 0xffff0000u);
                        const float y0 = __uint_as_float(yv[k] << 16), y1 = __uint_as_float(yv[k] & 0xffff0000u);
                        if (REDP != TCCT_ACT_NONE) {
                            d0 *= pw_act_grad<REDP < 0 ? 0 : REDP>(sP[kch + 2 * k] * y0 + sP[KS + kch + 2 * k]);
                            d1 *= pw_act_grad<REDP < 0 ? 0 : REDP>(sP[kch + 2 * k + 1] * y1 + sP[KS + kch + 2 * k + 1]);
                        }
                        rs[kt < RKT ? kt : 0][2 * k] += d0; rq[kt < RKT ? kt : 0][2 * k] += d0 * y0;
                        rs[kt < RKT ? kt : 0][2 * k + 1] += d1; rq[kt < RKT ? kt : 0][2 * k + 1] += d1 * y1;
                    }
                };
                if (SPLIT) {        // channels [0, K/2) -> dx, [K/2, K) -> dx2 (= dx_plain argument), rows of K/2 channels each
                    const uint32_t off2 = mm < M ? (uint32_t)((mm * KS + (kt * 32) % KS + cch * 8) * 2) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(o, kt * 32 < KS ? ro : rp, off2, 0, 0);
                    if (RED && kt < RKT) red_acc(o, kt * 32 + cch * 8);       // the BatchNorm sits in front of the FIRST half (x, not x2)
                    continue;
                }
                const uint32_t off = mm < M ? (uint32_t)((mm * K + kt * 32 + cch * 8) * 2) : 0x80000000u;
                if (GX) {           // dy1 = bf16(dh) gelu'(y1): what the separate activation backward computed from the stored dh
                    const u32x4 xv = xq[GX ? kt : 0][h2];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k] = pack_bf16x2(__uint_as_float(o[k] << 16) * gelu_grad1(__uint_as_float(xv[k] << 16)),
                                           __uint_as_float(o[k] & 0xffff0000u) * gelu_grad1(__uint_as_float(xv[k] & 0xffff0000u)));
                }
                if (res) {          // block-uniform: dx = dy W + res (the gradient that reaches x through its other consumers)
                    if (dx_plain) __builtin_amdgcn_raw_buffer_store_b128(o, rp, off, 0, 0);      // dy W itself (decoder tail: continues into the resize)
                    const u32x4 rv = __builtin_amdgcn_raw_buffer_load_b128(rr, off, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k] = pack_bf16x2(__uint_as_float(o[k] << 16) + __uint_as_float(rv[k] << 16),
                                           __uint_as_float(o[k] & 0xffff0000u) + __uint_as_float(rv[k] & 0xffff0000u));
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, ro, off, 0, 0);
                if (RED) red_acc(o, kt * 32 + cch * 8);
            }
            wave_lds_fence();
        }
        __syncthreads();                              // the next tile's image is complete
    }
    if (RED) {
        // lanes with equal (lane & 3) hold different pixels of the same 8 channels: butterfly over lane bits 2..5, per-wave LDS slots, fp64 atomics
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);           // [4 waves][2][K] (the staged tiles are dead)
        constexpr int KR = SPLIT ? K / 2 : K;
#pragma unroll
        for (int kt = 0; kt < RKT; ++kt) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float a = rs[kt][k], b = rq[kt][k];
#pragma unroll
                for (int o = 32; o > 2; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (lane < 4) {
                    red[wave * 2 * KR + kt * 32 + 8 * lane + k] = a;
                    red[wave * 2 * KR + KR + kt * 32 + 8 * lane + k] = b;
                }
            }
        }
        __syncthreads();
        for (int i2 = tid; i2 < KR; i2 += PWB) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { a += (double)red[wv * 2 * KR + i2]; b += (double)red[wv * 2 * KR + KR + i2]; }
            atomicAdd(&bn.sums_prev[i2], a); atomicAdd(&bn.sums_prev[KR + i2], b);
        }
        __syncthreads();
    }
    if (LNB) {
        // dgamma / dbeta: lanes with equal (lane & 3) hold different rows of the same 8 channels (see RED above)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);           // [4 waves][2][K]
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float a = rs[kt][k], b = rq[kt][k];
#pragma unroll
                for (int o = 32; o > 2; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (lane < 4) {
                    red[wave * 2 * K + kt * 32 + 8 * lane + k] = a;
                    red[wave * 2 * K + K + kt * 32 + 8 * lane + k] = b;
                }
            }
        }
        __syncthreads();
        for (int i2 = tid; i2 < K; i2 += PWB) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { a += red[wv * 2 * K + i2]; b += red[wv * 2 * K + K + i2]; }
            atomicAdd(&bn.dgamma[i2], a); atomicAdd(&bn.dbeta[i2], b);
        }
        __syncthreads();
    }
    // ---- weight-gradient tiles straight from the owning wave's registers: lanes r = 0..31 of a register are 128 contiguous bytes
#pragma unroll
    for (int a = 0; a < (NT * KT + 3) / 4; ++a) {
        const int ti = wave + 4 * a;
        if (ti < NT * KT) {
            const int ct = ti / KT, kt = ti - ct * KT;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int co = ct * 32 + (k & 3) + 8 * (k >> 2) + 4 * hh;
                atomicAdd(&dw[(int64_t)co * K + kt * 32 + r], accw[a][k]);
            }
            if (dbias && kt == 0) {
                const float sb = bsum[a] + __shfl_xor(bsum[a], 32, 64);
                if (lane < 32) atomicAdd(&dbias[ct * 32 + r], sb);
            }
        }
    }
}

/* Fused backward of y = x W^T + b for bf16 rows: dx [M,K] (= dy W, + res when res != NULL), dw [N,K] fp32 and dbias [N] fp32 (nullable)
 * are ACCUMULATED into after being cleared here (or by the caller: tcct_set_outputs_prezeroed).  K, N in {32, 64, 96, 128}. */
static int pw_bwd_impl(const void* x, const void* dy, const float* w, const void* res, void* dx, void* dx_plain, float* dw, float* dbias,
                       int64_t M, int K, int N, tcct_stream_t stream, bool split = false, int bnp = -1, int redp = -1,
                       PwBnBwd bn = PwBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr}, bool gelu_x = false,
                       int xap = -1, bool lnb = false, bool dy2 = false);
/* the same over a concatenation: x = [x1 | x2], dx = [dx1 | dx2], each [M, K/2] (K = 128): backward of tcct_pw_fwd_cat2 */
extern "C" int tcct_pw_bwd_cat2(const void* x1, const void* x2, const void* dy, const float* w, void* dx1, void* dx2, float* dw, int64_t M,
                                int K, int N, tcct_stream_t stream) {
    TCCT_CHECK(K == 128 && x2 != nullptr && dx2 != nullptr, "pw_bwd_cat2: two halves of 64 channels only (K=%d)", K);
    return pw_bwd_impl(x1, dy, w, x2, dx1, dx2, dw, nullptr, M, K, N, stream, true);
}
/* the concatenated form with a bias gradient and two halves of 32 OR 64 channels (K = 64 / 128): the decoder's composed tail, tcct_tail_compose */
extern "C" int tcct_pw_bwd_cat2_bias(const void* x1, const void* x2, const void* dy, const float* w, void* dx1, void* dx2, float* dw, float* dbias,
                                     int64_t M, int K, int N, tcct_stream_t stream) {
    TCCT_CHECK((K == 128 || (K == 64 && N == 32)) && x2 != nullptr && dx2 != nullptr, "pw_bwd_cat2_bias: K=%d N=%d (128 -> any, or 64 -> 32)", K, N);
    return pw_bwd_impl(x1, dy, w, x2, dx1, dx2, dw, dbias, M, K, N, stream, true);
}
extern "C" int tcct_pw_bwd(const void* x, const void* dy, const float* w, const void* res, void* dx, float* dw, float* dbias, int64_t M,
                           int K, int N, tcct_stream_t stream) {
    return pw_bwd_impl(x, dy, w, res, dx, nullptr, dw, dbias, M, K, N, stream);
}
/* the same with BOTH input gradients written: dx_sum = dy W + res and dx_plain = dy W (decoder block tail: the skip tensor's gradient and
 * the gradient that continues into the bilinear resize, reference nets/tcct.py:908-914,1028-1031) */
extern "C" int tcct_pw_bwd_residual2(const void* x, const void* dy, const float* w, const void* res, void* dx_sum, void* dx_plain, float* dw,
                                     float* dbias, int64_t M, int K, int N, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr && dx_plain != nullptr && dx_plain != dx_sum, "pw_bwd_residual2: needs res and two distinct outputs");
    return pw_bwd_impl(x, dy, w, res, dx_sum, dx_plain, dw, dbias, M, K, N, stream);
}
/* Shapes / activation kinds for which tcct_pw_bwd_bn has a kernel (the host mirror asks before building its autograd node):
 *   post (BatchNorm behind the convolution) and red_post (BatchNorm in front of it, -1: no reduction epilogue): TCCT_ACT_NONE / TCCT_ACT_HSWISH;
 *   K = N in {64, 96} (the reduction epilogue only at 64); N = 32 with K in {32, 96, 128} and the concatenated K = 128 -> N = 96 form without it. */
extern "C" int64_t tcct_pw_bwd_bn_supported(int K, int N, int post, int red_post, int split) {
    const bool p_ok = post == TCCT_ACT_NONE || post == TCCT_ACT_HSWISH;
    const bool r_ok = red_post == -1 || red_post == TCCT_ACT_NONE || red_post == TCCT_ACT_HSWISH;
    if (!p_ok || !r_ok) return 0;
    // Register budget (256 VGPRs at two blocks per CU): the reduction epilogue keeps 16 partial sums + 8 prefetched registers per 32 input
    // channels, which only fits at K = N = 64 (the concatenated 128 -> 96 form spills 51 VGPRs with it, 96 / 128 square 94 / 215); the plain
    // form spills 37 VGPRs at 128 x 128 and measured SLOWER than the separate kernels there (0.080 vs 0.045 + 0.02 ms at level 3): not offered.
    if (split) return K == 128 && N == 96 && post == TCCT_ACT_HSWISH && red_post == -1;
    if (K == N && (K == 64 || K == 96)) return red_post == -1 || K == 64;
    if (N == 32 && (K == 32 || K == 96 || K == 128)) return red_post == -1 && post == TCCT_ACT_NONE;
    return 0;
}
/* Backward of  z = post(BN_train(x W^T + bias)) [+ residual]  given dz (the gradient of z), in ONE pass over dz and y:
 *   dy_conv is rebuilt from (dz, y, coef) while the tiles are staged (coef [5][N] from tcct_bn_bwd_coef), then dx = dy_conv W (+ res),
 *   dw += dy_conv^T x, dbias += sum dy_conv as in tcct_pw_bwd.  x2 / dx2 non-NULL: the concatenated form of tcct_pw_bwd_cat2.
 *   red_post >= 0: x = post_prev(BN_prev(y_prev)) and dx (+ res) is the complete gradient of x -- sums_prev [2][K or K/2] (fp64, zero on entry)
 *   receive {sum dz', sum dz' y_prev} of BN_prev's backward (raw form: tcct_bn_bwd_coef(raw = 1) converts), ab_prev = BN_prev's {a[K], b[K]}. */
static int pw_bwd_bn_impl(const void* x, const void* x2, const void* dz, PwBnBwd bn, int post, const float* w, const void* res, void* dx, void* dx2,
                          float* dw, float* dbias, int64_t M, int K, int N, int red_post, tcct_stream_t stream);
extern "C" int tcct_pw_bwd_bn(const void* x, const void* x2, const void* dz, const void* y, const float* coef, int post, const float* w,
                              const void* res, void* dx, void* dx2, float* dw, float* dbias, int64_t M, int K, int N, const void* y_prev,
                              const float* ab_prev, int red_post, double* sums_prev, tcct_stream_t stream) {
    TCCT_CHECK(coef != nullptr, "pw_bwd_bn: coef is NULL");
    return pw_bwd_bn_impl(x, x2, dz, PwBnBwd{(const bf16*)y, coef, (const bf16*)y_prev, ab_prev, sums_prev, nullptr, 0, nullptr, nullptr, nullptr, nullptr},
                          post, w, res, dx, dx2, dw, dbias, M, K, N, red_post, stream);
}
/* the same with the constants derived in the kernel from the BatchNorm's batch sums (sums [2][N] fp64 from tcct_bn_bwd_reduce, or raw = 1: from a
 * reduction epilogue), mean_rstd / ab as saved by the forward; dgamma / dbeta [N] are written.  One launch instead of tcct_bn_bwd_coef + tcct_pw_bwd_bn. */
extern "C" int tcct_pw_bwd_bn_sums(const void* x, const void* x2, const void* dz, const void* y, const double* sums, int raw, const float* mean_rstd,
                                   const float* ab, float* dgamma, float* dbeta, int post, const float* w, const void* res, void* dx, void* dx2,
                                   float* dw, float* dbias, int64_t M, int K, int N, const void* y_prev, const float* ab_prev, int red_post,
                                   double* sums_prev, tcct_stream_t stream) {
    TCCT_CHECK(sums && mean_rstd && ab && dgamma && dbeta, "pw_bwd_bn_sums: NULL argument");
    return pw_bwd_bn_impl(x, x2, dz, PwBnBwd{(const bf16*)y, nullptr, (const bf16*)y_prev, ab_prev, sums_prev, sums, raw, mean_rstd, ab, dgamma, dbeta},
                          post, w, res, dx, dx2, dw, dbias, M, K, N, red_post, stream);
}
/* ... with x NOT materialised: `y_prev` is the input of the BatchNorm in front (train mode, Hardswish behind it) and x = hswish(a_prev y_prev + b_prev)
 * is rebuilt while the tiles are staged (ab_prev = {a[K], b[K]}).  K = N = 64: with the reduction epilogue (red_post = TCCT_ACT_HSWISH, sums_prev as
 * above); K = N = 96: without it (red_post = -1: the BatchNorm in front runs its own reduction).  post (the BatchNorm BEHIND) must be TCCT_ACT_NONE. */
extern "C" int tcct_pw_bwd_bn_sums_xaff(const void* y_prev, const float* ab_prev, const void* dz, const void* y, const double* sums, int raw,
                                        const float* mean_rstd, const float* ab, float* dgamma, float* dbeta, const float* w, const void* res, void* dx,
                                        float* dw, float* dbias, int64_t M, int K, int N, int red_post, double* sums_prev, tcct_stream_t stream) {
    TCCT_CHECK(y_prev && ab_prev && sums && mean_rstd && ab && dgamma && dbeta, "pw_bwd_bn_sums_xaff: NULL argument");
    TCCT_CHECK(K == N && ((K == 64 && red_post == TCCT_ACT_HSWISH && sums_prev) || (K == 96 && red_post < 0)), "pw_bwd_bn_sums_xaff: K=%d N=%d red_post=%d unsupported", K, N, red_post);
    if (red_post >= 0 && !tcct_skip_zero_fill() && hipMemsetAsync(sums_prev, 0, sizeof(double) * 2 * K, (hipStream_t)stream) != hipSuccess) { tcct_set_error("pw_bwd_bn_sums_xaff: memset failed"); return -2; }
    return pw_bwd_impl(y_prev, dz, w, res, dx, nullptr, dw, dbias, M, K, N, stream, false, TCCT_ACT_NONE, red_post,
                       PwBnBwd{(const bf16*)y, nullptr, (const bf16*)y_prev, ab_prev, sums_prev, sums, raw, mean_rstd, ab, dgamma, dbeta}, false, TCCT_ACT_HSWISH);
}
static int pw_bwd_bn_impl(const void* x, const void* x2, const void* dz, PwBnBwd bn, int post, const float* w, const void* res, void* dx, void* dx2,
                          float* dw, float* dbias, int64_t M, int K, int N, int red_post, tcct_stream_t stream) {
    const void* y = bn.y; const void* y_prev = bn.yprev; const float* ab_prev = bn.abprev; double* sums_prev = bn.sums_prev;
    const bool split = x2 != nullptr;
    TCCT_CHECK(tcct_pw_bwd_bn_supported(K, N, post, red_post, split ? 1 : 0), "pw_bwd_bn: K=%d N=%d post=%d red_post=%d split=%d unsupported", K, N, post,
               red_post, (int)split);
    TCCT_CHECK(y != nullptr && (red_post < 0 || (y_prev && ab_prev && sums_prev)), "pw_bwd_bn: NULL argument");
    TCCT_CHECK(!split || (dx2 != nullptr && res == nullptr), "pw_bwd_bn: the concatenated form takes x2 / dx2 and no residual");
    if (red_post >= 0 && !tcct_skip_zero_fill() &&
        hipMemsetAsync(sums_prev, 0, sizeof(double) * 2 * (split ? K / 2 : K), (hipStream_t)stream) != hipSuccess) { tcct_set_error("pw_bwd_bn: memset failed"); return -2; }
    return pw_bwd_impl(x, dz, w, split ? x2 : res, dx, split ? dx2 : nullptr, dw, dbias, M, K, N, stream, split, post, red_post, bn);
}
/* Backward of  y = gelu(x1) W^T + b  given dy, with x1 the PRE-activation (Mlp.fc1's output, reference nets/tcct.py:29-53): dx1 = (dy W) gelu'(x1),
 * dw += dy^T gelu(x1), dbias += sum dy in one pass; gelu(x1) is rebuilt while the tile is staged, the activation's own backward pass does not exist.
 * K = N in {64, 96}. */
extern "C" int tcct_pw_bwd_gelu(const void* x1, const void* dy, const float* w, void* dx1, float* dw, float* dbias, int64_t M, int K, int N,
                                tcct_stream_t stream) {
    TCCT_CHECK(K == N && (K == 64 || K == 96), "pw_bwd_gelu: K=%d N=%d unsupported (64 or 96 square; 128 spills 30 VGPRs)", K, N);
    return pw_bwd_impl(x1, dy, w, nullptr, dx1, nullptr, dw, dbias, M, K, N, stream, false, -1, -1,
                       PwBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr}, true);
}
/* tcct_pw_bwd_residual2 with dy = dy_a + dy_b summed while the tile is staged (rounded to bf16 like the tensor a separate add pass wrote); K = N = 32 */
extern "C" int tcct_pw_bwd_residual2_sum(const void* x, const void* dy_a, const void* dy_b, const float* w, const void* res, void* dx_sum, void* dx_plain, float* dw,
                                         float* dbias, int64_t M, int K, int N, tcct_stream_t stream) {
    TCCT_CHECK(K == 32 && N == 32, "pw_bwd_residual2_sum: K=%d N=%d unsupported (32 x 32)", K, N);
    TCCT_CHECK(dy_b != nullptr && res != nullptr && dx_plain != nullptr && dx_plain != dx_sum, "pw_bwd_residual2_sum: needs dy_b, res and two distinct outputs");
    return pw_bwd_impl(x, dy_a, w, res, dx_sum, dx_plain, dw, dbias, M, K, N, stream, false, -1, -1,
                       PwBnBwd{(const bf16*)dy_b, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr}, false, -1, false, true);
}
/* Backward of  y = LN(t; gamma, beta) W^T + b  (MHCABlock.norm2 -> Mlp.fc1, reference nets/tcct.py:466-468) given dy, x = LN(t) as stored, t, the LayerNorm's
 * saved statistics mean_rstd [M][2] and res = the gradient reaching t through the residual path: dt = LN^T(dy W) + res, dw, dbias as tcct_pw_bwd, dgamma / dbeta
 * [K] of the LayerNorm (cleared here unless the outputs are pre-zeroed) -- one pass, the gradient of x is never written.  K = N = 64. */
extern "C" int tcct_pw_bwd_lnb(const void* x, const void* dy, const float* w, const void* t, const float* mean_rstd, const float* gamma, const void* res,
                               void* dt, float* dw, float* dbias, float* dgamma, float* dbeta, int64_t M, int K, int N, tcct_stream_t stream) {
    TCCT_CHECK(K == 64 && N == 64, "pw_bwd_lnb: K=%d N=%d unsupported (64 x 64)", K, N);
    TCCT_CHECK(t && mean_rstd && gamma && res && dgamma && dbeta, "pw_bwd_lnb: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && (hipMemsetAsync(dgamma, 0, sizeof(float) * K, st) != hipSuccess || hipMemsetAsync(dbeta, 0, sizeof(float) * K, st) != hipSuccess)) {
        tcct_set_error("pw_bwd_lnb: memset failed"); return -2;
    }
    return pw_bwd_impl(x, dy, w, res, dt, nullptr, dw, dbias, M, K, N, stream, false, -1, -1,
                       PwBnBwd{nullptr, nullptr, (const bf16*)t, gamma, nullptr, nullptr, 0, mean_rstd, nullptr, dgamma, dbeta}, false, -1, true);
}
static int pw_bwd_impl(const void* x, const void* dy, const float* w, const void* res, void* dx, void* dx_plain, float* dw, float* dbias,
                       int64_t M, int K, int N, tcct_stream_t stream, bool split, int bnp, int redp, PwBnBwd bn, bool gelu_x, int xap, bool lnb, bool dy2) {
    TCCT_CHECK(K % 32 == 0 && N % 32 == 0 && K >= 32 && N >= 32 && K <= 128 && N <= 128, "pw_bwd: K=%d N=%d unsupported (32..128)", K, N);
    TCCT_CHECK(M > 0 && M * (int64_t)(K > N ? K : N) * 2 < (1LL << 31), "pw_bwd: tensor exceeds the 2 GiB buffer-descriptor range");
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * (size_t)N * K, st) != hipSuccess) { tcct_set_error("pw_bwd: memset failed"); return -2; }
    if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * N, st) != hipSuccess) { tcct_set_error("pw_bwd: memset failed"); return -2; }
    const int NT = N / 32, KT = K / 32;
    const int SX = 64 * KT + ((KT & 1) ? 0 : 64), SD = 64 * NT + ((NT & 1) ? 0 : 64), SW = 2 * N + 16;
    const size_t lds = (size_t)PB_P * (SX + SD) + (((size_t)K * SW + 15) & ~(size_t)15) + 4 * 2560 + (bnp >= 0 ? (size_t)5 * N * 4 : 0) +
                       ((redp >= 0 || xap >= 0 || lnb) ? (size_t)2 * K * 4 : 0);
    TCCT_CHECK(lds <= 160 * 1024, "pw_bwd: %zu B of LDS", lds);
    const int64_t tiles = (M + PB_P - 1) / PB_P;
    int per_cu = (int)((160 * 1024) / (lds + 256));
    if (per_cu > 2) per_cu = 2;
    // (one block per CU looked better in tools/kbench.py -- 64->64 @L1 0.135 -> 0.125 ms, 32->32 @L0 0.258 -> 0.235 -- but inside the training
    // step, where the tensors are not the same two buffers over and over, it was slower: k_pw_bwd 2.46 -> 2.57 ms per step; two per CU stay)
    int64_t gx = 256 * per_cu;
    if (gx > tiles) gx = tiles;
#define BLX(NTV, KTV, SPV, BNV, RDV) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_bwd<NTV, KTV, SPV, BNV, RDV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_bwd<NTV, KTV, SPV, BNV, RDV>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)dy, w, (const bf16*)res, (bf16*)dx, (bf16*)dx_plain, dw, dbias, M, bn); }
    if (dy2) {          // decoder block tail: the two output gradients summed on load (K = N = 32)
        { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_bwd<1, 1, false, -1, -1, false, -1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; }
          hipLaunchKernelGGL((k_pw_bwd<1, 1, false, -1, -1, false, -1, false, true>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)dy, w, (const bf16*)res, (bf16*)dx, (bf16*)dx_plain, dw, dbias, M, bn); }
        TCCT_LAUNCH_OK();
    }
    if (lnb) {          // fc1 behind MHCABlock.norm2: the LayerNorm backward in the dx epilogue (K = N = 64)
#define BLN(T) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_bwd<T, T, false, -1, -1, false, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_bwd<T, T, false, -1, -1, false, -1, true>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)dy, w, (const bf16*)res, (bf16*)dx, (bf16*)dx_plain, dw, dbias, M, bn); }
        BLN(2)
#undef BLN
        TCCT_LAUNCH_OK();
    }
    if (xap >= 0) {     // x = hswish(a y_prev + b) applied on load (InvRes.norm -> conv2): BN behind without activation; 64: + the reduction epilogue
#define BLXA(NTV, RDV) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_bwd<NTV, NTV, false, 0, RDV, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_bwd<NTV, NTV, false, 0, RDV, false, 2>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)dy, w, (const bf16*)res, (bf16*)dx, (bf16*)dx_plain, dw, dbias, M, bn); }
        if (NT == 2 && redp == TCCT_ACT_HSWISH) BLXA(2, 2) else if (NT == 3 && redp < 0) BLXA(3, -1)
        else { tcct_set_error("pw_bwd: x-affine form K=N=%d red_post=%d has no kernel", K, redp); return -1; }
#undef BLXA
        TCCT_LAUNCH_OK();
    }
    if (bnp >= 0) {
        // the combinations the network has (tcct_pw_bwd_bn_supported): every extra instantiation of this kernel costs ~a minute of hipcc
        const int H = TCCT_ACT_HSWISH;
#define BN_SQ(T) \
        if (bnp == 0 && redp < 0) BLX(T, T, false, 0, -1) else BLX(T, T, false, 2, -1)
#define BN_SQR(T) \
        if (bnp == 0 && redp == 0) BLX(T, T, false, 0, 0) else if (bnp == 0 && redp == H) BLX(T, T, false, 0, 2) \
        else if (bnp == H && redp == 0) BLX(T, T, false, 2, 0) else BLX(T, T, false, 2, 2)
        static_assert(TCCT_ACT_HSWISH == 2 && TCCT_ACT_NONE == 0, "activation codes are template arguments below");
        if (split) BLX(3, 4, true, 2, -1)
        else if (NT == 1 && KT == 1) BLX(1, 1, false, 0, -1)
        else if (NT == 1 && KT == 3) BLX(1, 3, false, 0, -1)
        else if (NT == 1 && KT == 4) BLX(1, 4, false, 0, -1)
        else if (NT == 2) { if (redp < 0) { BN_SQ(2) } else { BN_SQR(2) } }
        else { BN_SQ(3) }
#undef BN_SQ
#undef BN_SQR
        TCCT_LAUNCH_OK();
    }
    if (gelu_x) {
#define BLG(T) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_bwd<T, T, false, -1, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_bwd<T, T, false, -1, -1, true>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)dy, w, (const bf16*)res, (bf16*)dx, (bf16*)dx_plain, dw, dbias, M, bn); }
        if (NT == 2) BLG(2) else BLG(3)
#undef BLG
        TCCT_LAUNCH_OK();
    }
    if (split && KT == 2) { BLX(1, 2, true, -1, -1) TCCT_LAUNCH_OK(); }     // two halves of 32 channels (checked by the caller: N = 32)
#define BL(NTV, KTV) BLX(NTV, KTV, false, -1, -1)
#define BLS(NTV) BLX(NTV, 4, true, -1, -1)
#define BLK(NTV) switch (KT) { case 1: BL(NTV, 1) break; case 2: BL(NTV, 2) break; case 3: BL(NTV, 3) break; default: if (split) BLS(NTV) else BL(NTV, 4) break; }
    switch (NT) { case 1: BLK(1) break; case 2: BLK(2) break; case 3: BLK(3) break; default: BLK(4) break; }
#undef BLK
#undef BLS
#undef BL
#undef BLX
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ forward, tile-staged
// Y[M,N] = X[M,K] W^T + bias for K, N <= 128 with the activations staged through LDS: k_pw_fwd reads its B fragments straight from
// global memory, 16 bytes per lane in the MFMA operand layout, i.e. every wave instruction touches 32 different rows and 25 % (K = 64)
// or 12.5 % (K = 128) of each 128-byte line -- 3.8 TB/s where the K = 32 case (whole rows per instruction pair) reaches 5.7.  Here a
// 128-pixel tile is ONE contiguous span of memory, loaded fully coalesced (thread i takes bytes [16 i, 16 i + 16) of it), written to
// LDS rows of 2K+16 bytes (conflict-free ds_read_b128 for the fragment layout) and prefetched one tile ahead; wave w multiplies pixels
// 32w..32w+31 against all N-tiles.  Same tile pipeline as k_pw_bwd / k_conv32_mfma (staging wait behind the MFMA phase, stores after
// the next prefetch).  SPLIT: the rows are the concatenation [x | x2] of two tensors of K/2 channels each (MHCA_stage.aggregate).
// RES: y = res + rscale[m / per_sample] * (x W^T + bias) with the product rounded to bf16 first, like the op-by-op path (Mlp.fc2 with the
// residual add and the DropPath scale of MHCABlock folded in, reference nets/tcct.py:468); yplain (nullable) also receives the product.
// AFF (round 6, inference): y = act(aff_post, a[c] * (x W^T + bias[c]) + b[c]) in the epilogue, ab = {a[N], b[N]} from the eval-mode BatchNorm (NULL: a = 1, b = 0) --
// what k_pw_fwd's affine epilogue does for the direct-from-global kernel (0.187 ms for 64 -> 64 at level 1 against 0.10 here); combines with RES and SPLIT.
template <int NT, int KT, bool STATS, bool SPLIT, bool RES = false, bool GX = false, int XAP = -1, bool AFF = false>
__global__ void __launch_bounds__(PWB, 2)
k_pw_fwd2(const bf16* __restrict__ x, const bf16* __restrict__ x2, const float* __restrict__ w, const float* __restrict__ bias,
          bf16* __restrict__ y, int64_t M, double* __restrict__ stats, int stat_pre, PwRes pr) {
    constexpr int K = 32 * KT, N = 32 * NT;
    constexpr int SX = 2 * K + 16, SW = 2 * K + 16;
    constexpr int XS = K / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sW = sX + PB_P * SX;
    float* sB = reinterpret_cast<float*>(sW + N * SW);
    unsigned char* sS = reinterpret_cast<unsigned char*>(sB + N);       // per-wave epilogue transpose scratch: 32 px x 80 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    for (int i = tid; i < N * (K / 8); i += PWB) {
        const int n = i / (K / 8), c8 = i - n * (K / 8);
        const float4 a = *reinterpret_cast<const float4*>(w + (int64_t)n * K + c8 * 8);
        const float4 b = *reinterpret_cast<const float4*>(w + (int64_t)n * K + c8 * 8 + 4);
        uint4 o;
        o.x = pack_bf16x2(a.x, a.y); o.y = pack_bf16x2(a.z, a.w); o.z = pack_bf16x2(b.x, b.y); o.w = pack_bf16x2(b.z, b.w);
        *reinterpret_cast<uint4*>(sW + n * SW + c8 * 16) = o;
    }
    if (tid < N) sB[tid] = bias ? bias[tid] : 0.f;
    float* sXA = reinterpret_cast<float*>(sS + 4 * 2560);             // XAP: a[K], b[K] of the BatchNorm in front;  AFF: a[N], b[N] of the one behind
    if (XAP >= 0) for (int i = tid; i < 2 * K; i += PWB) sXA[i] = pr.xab[i];
    float* sAF = sXA + (XAP >= 0 ? 2 * K : 0);                         // (both at once: InvRes.norm applied on load AND conv2.bn + residual in the epilogue, inference)
    if (AFF) for (int i = tid; i < 2 * N; i += PWB) sAF[i] = pr.aff ? pr.aff[i] : (i < N ? 1.f : 0.f);
    float ss[STATS ? NT : 1][8], sq[STATS ? NT : 1][8];
#pragma unroll
    for (int a = 0; a < (STATS ? NT : 1); ++a)
#pragma unroll
        for (int k = 0; k < 8; ++k) ss[a][k] = sq[a][k] = 0.f;
    const int64_t tiles = (M + PB_P - 1) / PB_P;
    constexpr int KS = SPLIT ? K / 2 : K;                  // channels per source tensor
    const uint32_t xbytes = (uint32_t)(M * KS * 2), ybytes = (uint32_t)(M * N * 2);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rx2 = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? x2 : x), 0, xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)y, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rres = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? (const void*)pr.res : (const void*)y), 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rpl = __builtin_amdgcn_make_buffer_rsrc((void*)((RES && pr.yplain) ? pr.yplain : y), 0, ybytes, 0x00020000);
    u32x4 px[XS];
    auto prefetch = [&](int64_t tile) {
        const uint32_t bx = (uint32_t)(tile * PB_P * KS * 2);
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            if (SPLIT && j >= XS / 2) px[j] = __builtin_amdgcn_raw_buffer_load_b128(rx2, bx + (uint32_t)(tid + (j - XS / 2) * PWB) * 16u, 0, 0);
            else px[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, bx + (uint32_t)(tid + j * PWB) * 16u, 0, 0);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < XS; ++j) {
            const int jj = (SPLIT && j >= XS / 2) ? j - XS / 2 : j;
            const int q = tid + jj * PWB, p = q / (KS / 8), c = q - p * (KS / 8);
            u32x4 xv_ = GX ? gelu8(px[j]) : px[j];
            if (XAP >= 0) xv_ = affine8<XAP < 0 ? 0 : XAP>(xv_, sXA + c * 8, sXA + K + c * 8);
            *reinterpret_cast<u32x4*>(sX + p * SX + ((SPLIT && j >= XS / 2) ? KS * 2 : 0) + c * 16) = xv_;
        }
    };
    const unsigned char* bB = sX + (32 * wave + r) * SX + hh * 16;
    int64_t tile = blockIdx.x;
    if (tile < tiles) {
        prefetch(tile);
        __syncthreads();
        stage();
        prefetch(tile + gridDim.x);                 // beyond the last tile every row is out of range: zeros, never used
    }
    __syncthreads();
    for (; tile < tiles; tile += gridDim.x) {
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[nt][k] = 0.f;
#pragma unroll
        for (int st = 0; st < 2 * KT; ++st) {
            const bf16x8 fb = *reinterpret_cast<const bf16x8*>(bB + st * 32);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8 fa = *reinterpret_cast<const bf16x8*>(sW + (nt * 32 + r) * SW + st * 32 + hh * 16);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[nt], 0, 0, 0);
            }
        }
        __syncthreads();
        if (GX || XAP >= 0) {
            // the on-load transform (GELU / BatchNorm + Hardswish) is ~200 VALU instructions per slot: with stage() first and the prefetch behind it, the
            // loads of tile t + 2 left ~1000 cycles late per tile and the kernel lost a quarter of its bandwidth (fc2 + GELU: 0.200 ms against
            // 0.143 without the transform).  Slot by slot instead: copy the arrived registers, re-issue the slot's load, THEN transform and store.
            const bool do_stage = tile + (int64_t)gridDim.x < tiles;
            const uint32_t bx = (uint32_t)((tile + 2 * (int64_t)gridDim.x) * PB_P * KS * 2);
#pragma unroll
            for (int j = 0; j < XS; ++j) {
                u32x4 xv_ = px[j];
                px[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, bx + (uint32_t)(tid + j * PWB) * 16u, 0, 0);
                if (do_stage) {
                    const int q = tid + j * PWB, p = q / (KS / 8), c = q - p * (KS / 8);
                    if (GX) xv_ = gelu8(xv_);
                    if (XAP >= 0) xv_ = affine8<XAP < 0 ? 0 : XAP>(xv_, sXA + c * 8, sXA + K + c * 8);
                    *reinterpret_cast<u32x4*>(sX + p * SX + c * 16) = xv_;
                }
            }
        } else {
            if (tile + (int64_t)gridDim.x < tiles) stage();
            prefetch(tile + 2 * (int64_t)gridDim.x);
        }
        unsigned char* sc = sS + wave * 2560;
        const int64_t m0 = tile * PB_P + 32 * wave;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bq = *reinterpret_cast<const float4*>(sB + nt * 32 + 8 * q + 4 * hh);
                float v0 = acc[nt][4 * q] + bq.x, v1 = acc[nt][4 * q + 1] + bq.y, v2 = acc[nt][4 * q + 2] + bq.z, v3 = acc[nt][4 * q + 3] + bq.w;
                if (AFF) {
                    const float4 aq = *reinterpret_cast<const float4*>(sAF + nt * 32 + 8 * q + 4 * hh);
                    const float4 cq = *reinterpret_cast<const float4*>(sAF + N + nt * 32 + 8 * q + 4 * hh);
                    float v4[4] = {v0, v1, v2, v3};
                    const float a4[4] = {aq.x, aq.y, aq.z, aq.w}, c4[4] = {cq.x, cq.y, cq.z, cq.w};
                    affine4(v4, a4, c4, TCCT_ACT_NONE, pr.aff_post);      // (the kind resolved once per four values, not per value: common.h)
                    v0 = v4[0]; v1 = v4[1]; v2 = v4[2]; v3 = v4[3];
                }
                uint2 o;
                o.x = pack_bf16x2(v0, v1);
                o.y = pack_bf16x2(v2, v3);
                *reinterpret_cast<uint2*>(sc + r * 80 + (8 * q + 4 * hh) * 2) = o;
            }
            wave_lds_fence();
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int p = (lane >> 2) + 16 * h2, cch = lane & 3;
                u32x4 o = *reinterpret_cast<const u32x4*>(sc + p * 80 + cch * 16);
                const int64_t mm = m0 + p;
                const bool inb = mm < M;
                const uint32_t off = inb ? (uint32_t)((mm * N + nt * 32 + cch * 8) * 2) : 0x80000000u;
                if (RES) {
                    if (pr.yplain) __builtin_amdgcn_raw_buffer_store_b128(o, rpl, off, 0, 0);
                    const u32x4 rv = __builtin_amdgcn_raw_buffer_load_b128(rres, off, 0, 0);
                    const float sc_ = pr.rscale ? pr.rscale[(inb ? mm : 0) / pr.per_sample] : 1.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        o[k] = pack_bf16x2(__uint_as_float(rv[k] << 16) + sc_ * __uint_as_float(o[k] << 16),
                                           __uint_as_float(rv[k] & 0xffff0000u) + sc_ * __uint_as_float(o[k] & 0xffff0000u));
                }
                __builtin_amdgcn_raw_buffer_store_b128(o, ro, off, 0, 0);
                if (STATS) {
                    if (inb) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float u0 = __uint_as_float(o[k] << 16), u1 = __uint_as_float(o[k] & 0xffff0000u);
                            if (stat_pre != TCCT_ACT_NONE) { u0 = act_fwd(stat_pre, u0); u1 = act_fwd(stat_pre, u1); }
                            ss[nt][2 * k] += u0; sq[nt][2 * k] += u0 * u0; ss[nt][2 * k + 1] += u1; sq[nt][2 * k + 1] += u1 * u1;
                        }
                    }
                }
            }
            wave_lds_fence();
        }
        __syncthreads();
    }
    if (STATS) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);           // [4 waves][2][N] (the staged tile is dead)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int k = 0; k < 8; ++k) {       // lanes with equal (lane & 3) hold different pixels of channels 8*(lane&3)+k
                float a = ss[nt][k], b = sq[nt][k];
#pragma unroll
                for (int o = 32; o > 2; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (lane < 4) {
                    red[wave * 2 * N + nt * 32 + 8 * lane + k] = a;
                    red[wave * 2 * N + N + nt * 32 + 8 * lane + k] = b;
                }
            }
        __syncthreads();
        for (int i2 = tid; i2 < N; i2 += PWB) {
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) { a += (double)red[wv * 2 * N + i2]; b += (double)red[wv * 2 * N + N + i2]; }
            atomicAdd(&stats[i2], a); atomicAdd(&stats[N + i2], b);
        }
    }
}

static bool pw_fwd2_ok(int64_t M, int K, int N, int K1, bool has_x2) {
    if (K % 32 || N % 32 || K < 64 || K > 128 || N < 32 || N > 128) return false;      // K = 32 already streams whole rows in k_pw_fwd
    if (M * (int64_t)(K > N ? K : N) * 2 >= (1LL << 31)) return false;
    if (has_x2 && !(K == 128 && K1 == 64)) return false;
    return true;
}
static int pw_fwd2_launch(const void* x, const void* x2, const float* w, const float* bias, void* y, int64_t M, int K, int N, double* stats,
                          int stat_pre, tcct_stream_t stream, PwRes pr, bool gelu_x) {
    const int NT = N / 32, KT = K / 32;
    const size_t lds = (size_t)PB_P * (2 * K + 16) + (size_t)N * (2 * K + 16) + (size_t)N * 4 + 4 * 2560 + (pr.xab ? (size_t)2 * K * 4 : 0) + (pr.has_aff ? (size_t)2 * N * 4 : 0);
    const int64_t tiles = (M + PB_P - 1) / PB_P;
    int per_cu = (int)((160 * 1024) / (lds + 256));
    if (per_cu > 2) per_cu = 2;
    // (as for k_pw_bwd: one block per CU won by 3-10 % in the micro-benchmark and lost inside the step -- the statistics variants by 60 %)
    int64_t gx = 256 * per_cu;
    if (gx > tiles) gx = tiles;
    hipStream_t st = (hipStream_t)stream;
#define F2(NTV, KTV, SV, PV) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_fwd2<NTV, KTV, SV, PV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_fwd2<NTV, KTV, SV, PV>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)x2, w, bias, (bf16*)y, M, stats, stat_pre, pr); }
#define F2R(NTV, KTV) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_fwd2<NTV, KTV, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_fwd2<NTV, KTV, false, false, true>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)x2, w, bias, (bf16*)y, M, stats, stat_pre, pr); }
#define F2K(NTV, SV) switch (KT) { case 2: F2(NTV, 2, SV, false) break; case 3: F2(NTV, 3, SV, false) break; default: if (x2) F2(NTV, 4, SV, true) else F2(NTV, 4, SV, false) break; }
#define F2RK(NTV) switch (KT) { case 2: F2R(NTV, 2) break; case 3: F2R(NTV, 3) break; default: F2R(NTV, 4) break; }
#define F2N(SV) switch (NT) { case 1: F2K(1, SV) break; case 2: F2K(2, SV) break; case 3: F2K(3, SV) break; default: F2K(4, SV) break; }
#define F2G(T) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_fwd2<T, T, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_fwd2<T, T, false, false, true, true>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)x2, w, bias, (bf16*)y, M, stats, stat_pre, pr); }
#define F2A(T) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_fwd2<T, T, true, false, false, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_fwd2<T, T, true, false, false, false, 2>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)x2, w, bias, (bf16*)y, M, stats, stat_pre, pr); }
#define F2F(NTV, KTV, SPV, RSV) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_fwd2<NTV, KTV, false, SPV, RSV, false, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_fwd2<NTV, KTV, false, SPV, RSV, false, -1, true>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)x2, w, bias, (bf16*)y, M, stats, stat_pre, pr); }
#define F2X(T) { static bool at_ = false; if (!at_) { (void)hipFuncSetAttribute((const void*)k_pw_fwd2<T, T, false, false, true, false, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); at_ = true; } \
        hipLaunchKernelGGL((k_pw_fwd2<T, T, false, false, true, false, 2, true>), dim3((unsigned)gx), dim3(PWB), lds, st, (const bf16*)x, (const bf16*)x2, w, bias, (bf16*)y, M, stats, stat_pre, pr); }
    if (pr.has_aff && pr.xab) {           // inference: BatchNorm + Hardswish in front applied on load, BatchNorm behind + residual in the epilogue (square 64 / 96 / 128)
        if (NT == 2) F2X(2) else if (NT == 3) F2X(3) else F2X(4)
    }
    else
    if (pr.has_aff) {           // inference epilogues (pw_fwd2_aff_ok lists the shapes)
        if (x2) F2F(3, 4, true, false)
        else if (pr.res) { if (NT == 2) F2F(2, 2, false, true) else if (NT == 3) F2F(3, 3, false, true) else F2F(4, 4, false, true) }
        else { if (NT == 2) F2F(2, 2, false, false) else if (NT == 3) F2F(3, 3, false, false) else F2F(4, 4, false, false) }
    }
    else
    if (pr.xab) { if (NT == 2) F2A(2) else F2A(3) }
    else
    if (gelu_x) { if (NT == 2) F2G(2) else F2G(3) }
    else
    if (pr.res) { switch (NT) { case 1: F2RK(1) break; case 2: F2RK(2) break; case 3: F2RK(3) break; default: F2RK(4) break; } }
    else if (stats) { F2N(true) } else { F2N(false) }
#undef F2X
#undef F2F
#undef F2G
#undef F2A
#undef F2RK
#undef F2R
#undef F2N
#undef F2K
#undef F2
    TCCT_LAUNCH_OK();
}

static int pw_fwd2_route(const void* x, const void* x2, const float* w, const float* bias, void* y, int64_t M, int K, int N, double* stats,
                         int stat_pre, tcct_stream_t stream, const bf16* res, const float* rscale, int64_t per_sample, bf16* yplain) {
    return pw_fwd2_launch(x, x2, w, bias, y, M, K, N, stats, stat_pre, stream, PwRes{res, rscale, per_sample, yplain});
}
/* y = hswish(a_prev y_prev + b_prev) W^T + bias with the statistics of y for the BatchNorm behind (stats fp64 [2N], zero on entry): the convolution
 * whose input is a train-mode BatchNorm + Hardswish it alone consumes (InvRes.norm -> conv2, nets/tcct.py:563-572), the normalisation applied while the
 * tile is staged; ab_prev = {a[K], b[K]}.  K = N in {64, 96}. */
extern "C" int tcct_pw_fwd_bnstats_xaff(const void* y_prev, const float* ab_prev, const float* w, const float* bias, void* y, int64_t M, int K, int N,
                                        double* stats, tcct_stream_t stream) {
    TCCT_CHECK(K == N && (K == 64 || K == 96) && ab_prev && stats, "pw_fwd_bnstats_xaff: K=%d N=%d unsupported (64 or 96 square) or NULL argument", K, N);
    TCCT_CHECK(M > 0 && M * (int64_t)K * 2 < (1LL << 31), "pw_fwd_bnstats_xaff: tensor exceeds the 2 GiB buffer-descriptor range");
    return pw_fwd2_launch(y_prev, nullptr, w, bias, y, M, K, N, stats, 0, stream, PwRes{nullptr, nullptr, 1, nullptr, ab_prev});
}
/* y = res + scale[m / per_sample] * (gelu(x1) W^T + bias) with x1 the PRE-activation of Mlp.fc1 (reference nets/tcct.py:29-53,468): Mlp.fc2 with the
 * activation applied while the tile is staged and the DropPath scale + residual add in the epilogue; scale nullable.  K = N in {64, 96}. */
extern "C" int tcct_pw_fwd_gelu_residual(const void* x1, const float* w, const float* bias, const void* res, const float* scale, int64_t per_sample,
                                         void* y, int64_t M, int K, int N, tcct_stream_t stream) {
    TCCT_CHECK(K == N && (K == 64 || K == 96), "pw_fwd_gelu_residual: K=%d N=%d unsupported (64 or 96 square, as tcct_pw_bwd_gelu)", K, N);
    TCCT_CHECK(res != nullptr && per_sample >= 1 && M > 0 && M * (int64_t)K * 2 < (1LL << 31), "pw_fwd_gelu_residual: needs res, per_sample >= 1, < 2 GiB tensors");
    return pw_fwd2_launch(x1, nullptr, w, bias, y, M, K, N, nullptr, 0, stream, PwRes{(const bf16*)res, scale, per_sample, nullptr}, true);
}

// ------------------------------------------------------------------------------------------------ first-layer im2col
// The two 3-channel 3x3 convolutions (CrossResNet.cnn[0], reference nets/tcct.py:873; MPViT stem[0] stride 2, :674-681) have
// K = 27: expand the 4-channel NHWC image once into 32-channel "patch pixels" (k = (ky*3+kx)*3 + ch, k >= 27 zero) so that
// both run as 32->32 pointwise MFMA GEMMs (forward AND weight gradient); the image itself needs no gradient.
template <typename T>
__global__ void k_im2col3x3_c3(const T* __restrict__ x4, T* __restrict__ out, int N, int H, int W, int stride, int Ho, int Wo) {
    const int64_t total = (int64_t)N * Ho * Wo * 8;       // 8 lanes per output pixel, 4 patch channels each
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int sub = (int)(i & 7);
        int64_t p = i >> 3;
        const int wo = (int)(p % Wo);
        int64_t q = p / Wo;
        const int ho = (int)(q % Ho);
        const int64_t n = q / Ho;
        f4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = sub * 4 + j;
            float v = 0.f;
            if (k < 27) {
                const int tap = k / 3, ch = k - tap * 3;
                const int hi = ho * stride + tap / 3 - 1, wi = wo * stride + tap % 3 - 1;
                if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = ldf(x4 + ((n * H + hi) * (int64_t)W + wi) * 4 + ch);
            }
            o.v[j] = v;
        }
        st4(out + p * 32 + sub * 4, o);
    }
}
extern "C" int tcct_im2col3x3_c3(const void* x4, void* out, int N, int H, int W, int stride, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "im2col3x3_c3: stride %d", stride);
    int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    int64_t total = (int64_t)N * Ho * Wo * 8;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL(k_im2col3x3_c3<T>, dim3(tcct_grid(total, PWB, 1 << 16)), dim3(PWB), 0, (hipStream_t)stream, (const T*)x4, (T*)out, N, H, W, stride, Ho, Wo));
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ small-N weight gradient
// dW[n][k] (N <= 8 outputs: the 5-class aux heads, reference nets/tcct.py:994-997,1041-1044) with fp32 or bf16 dy: HBM-bound
// streaming reduction; thread = (4 input channels, pixel slot), N x 4 register sums, LDS combine, fp32 atomics.
template <typename Tx, typename Td>
__global__ void k_pw_wgrad_smalln(const Tx* __restrict__ x, const Td* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
                                  int64_t M, int K, int N) {
    extern __shared__ float smf[];                     // [R][N*K]
    const int KV = K >> 2, R = PWB / KV;
    const int t = threadIdx.x, kv = t % KV, r = t / KV;
    float acc[8][4], bs[8];
#pragma unroll
    for (int n = 0; n < 8; ++n) { bs[n] = 0.f; acc[n][0] = acc[n][1] = acc[n][2] = acc[n][3] = 0.f; }
    if (r < R) {
        const int64_t step = (int64_t)gridDim.x * R;
        for (int64_t m0 = (int64_t)blockIdx.x * R + r; m0 < M; m0 += 2 * step) {
            f4 xv[2];
            float d[2][8];
#pragma unroll
            for (int u = 0; u < 2; ++u) {               // 2 rows in flight per thread
                const int64_t m = m0 + u * step;
                const bool ok = m < M;
                xv[u] = ok ? ld4(x + m * K + kv * 4) : f4zero();
#pragma unroll
                for (int n = 0; n < 8; ++n) d[u][n] = (ok && n < N) ? ldf(dy + m * N + n) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int n = 0; n < 8; ++n) {
                    bs[n] += d[u][n];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[n][j] += d[u][n] * xv[u].v[j];
                }
        }
    }
    for (int i = t; i < R * N * K; i += PWB) smf[i] = 0.f;
    __syncthreads();
    if (r < R) {
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (n < N) {
#pragma unroll
                for (int j = 0; j < 4; ++j) smf[(r * N + n) * K + kv * 4 + j] = acc[n][j];
            }
    }
    __syncthreads();
    for (int i = t; i < N * K; i += PWB) {
        float a = 0.f;
        for (int rr = 0; rr < R; ++rr) a += smf[rr * N * K + i];
        atomicAdd(&dw[i], a);
    }
    if (dbias && kv == 0 && r < R) {
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (n < N) atomicAdd(&dbias[n], bs[n]);
    }
}
// MFMA form for the hot case (bf16 x with K = 32, fp32 dy, N <= 8: the aux heads): dy rows are converted to bf16 and padded to 32
// channels in LDS (chunks 1..3 of every row are written once: they stay zero), then the 32x32 transposing-read MFMA of
// k_pw_wgrad does the pixel contraction; only rows < N of the accumulator are written back.  HBM-bound streaming of x.
__global__ void __launch_bounds__(PWB, 2)
k_pw_wgrad_smalln_mfma(const bf16* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
                       int64_t M, int N) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * PW_P * 64];
    unsigned char* sX = smem;
    unsigned char* sD = smem + PW_P * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    for (int i = tid; i < PW_P * 4; i += PWB) *reinterpret_cast<uint4*>(sD + i * 16) = make_uint4(0, 0, 0, 0);
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    float bsum = 0.f;
    uint4 px[2];
    float pdv[8];
    const int dp = tid;                                   // dy pixel handled by this thread (tid < PW_P)
    auto prefetch = [&](int64_t tile) {
        const int64_t m0 = tile * PW_P;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = tid + j * PWB, p = i >> 2, c = i & 3;
            px[j] = make_uint4(0, 0, 0, 0);
            if (m0 + p < M) px[j] = *reinterpret_cast<const uint4*>(x + (m0 + p) * 32 + c * 8);
        }
#pragma unroll
        for (int n = 0; n < 8; ++n) pdv[n] = (dp < PW_P && n < N && m0 + dp < M) ? dy[(m0 + dp) * N + n] : 0.f;
    };
    const int li = lane & 15, lq = li >> 2, lpp = li & 3, lg = lane >> 4;
    const int loff = (8 * (lg >> 1) + lq) * 64 + (16 * (lg & 1) + 4 * lpp) * 2;
    auto tr2 = [&](const unsigned char* p) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 256));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };
    const int64_t tiles = (M + PW_P - 1) / PW_P;
    int64_t tile = blockIdx.x;
    if (tile < tiles) prefetch(tile);
    for (; tile < tiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) { const int i = tid + j * PWB; *reinterpret_cast<uint4*>(sX + (i >> 2) * 64 + (i & 3) * 16) = px[j]; }
        if (dp < PW_P) {
            uint4 o;
            o.x = pack_bf16x2(pdv[0], pdv[1]); o.y = pack_bf16x2(pdv[2], pdv[3]); o.z = pack_bf16x2(pdv[4], pdv[5]); o.w = pack_bf16x2(pdv[6], pdv[7]);
            *reinterpret_cast<uint4*>(sD + dp * 64) = o;
        }
        __syncthreads();
        if (tile + gridDim.x < tiles) prefetch(tile + gridDim.x);
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int ch = wave + 4 * c2;
            const bf16x8 a = tr2(sD + loff + ch * 1024);
            const bf16x8 b = tr2(sX + loff + ch * 1024);
#pragma unroll
            for (int j = 0; j < 8; ++j) bsum += (float)a[j];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);          // [8][32]; the four waves take turns (no LDS float atomics)
    for (int turn = 0; turn < 4; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                   // k < 4 <=> co = k + 4*hh < 8
                float* dst = &red[(k + 4 * hh) * 32 + r];
                *dst = turn == 0 ? acc[k] : *dst + acc[k];
            }
        }
        __syncthreads();
    }
    if (tid < N * 32) atomicAdd(&dw[tid], red[tid]);
    if (dbias) {
        float s = bsum + __shfl_xor(bsum, 32, 64);
        if (lane < N) atomicAdd(&dbias[lane], s);
    }
}

/* x [M,K] (K % 4 == 0, K <= 256), dy [M,N] fp32 or bf16 (N <= 8) -> dw fp32 [N,K], dbias [N] (nullable); overwritten */
extern "C" int tcct_pw_wgrad_smalln(const void* x, const void* dy, float* dw, float* dbias, int64_t M, int K, int N, int x_dtype,
                                    int dy_dtype, tcct_stream_t stream) {
    TCCT_CHECK(K % 4 == 0 && K >= 4 && K <= 256 && N >= 1 && N <= 8, "pw_wgrad_smalln: unsupported K=%d N=%d", K, N);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * N * K, st) != hipSuccess) { tcct_set_error("pw_wgrad_smalln: memset failed"); return -2; }
    if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * N, st) != hipSuccess) { tcct_set_error("pw_wgrad_smalln: memset failed"); return -2; }
    if (K == 32 && x_dtype == TCCT_BF16 && dy_dtype == TCCT_F32) {
        const int64_t tiles = (M + PW_P - 1) / PW_P;
        hipLaunchKernelGGL(k_pw_wgrad_smalln_mfma, dim3((unsigned)(tiles < 512 ? tiles : 512)), dim3(PWB), 0, st, (const bf16*)x, (const float*)dy, dw,
                           dbias, M, N);
        TCCT_LAUNCH_OK();
    }
    int R = PWB / (K / 4);
    size_t lds = sizeof(float) * (size_t)R * N * K;
    int grid = tcct_grid(M, R, 1024);
#define SL(TX, TD) hipLaunchKernelGGL((k_pw_wgrad_smalln<TX, TD>), dim3(grid), dim3(PWB), lds, st, (const TX*)x, (const TD*)dy, dw, dbias, M, K, N)
    if (x_dtype == TCCT_BF16 && dy_dtype == TCCT_F32) SL(bf16, float);
    else if (x_dtype == TCCT_BF16 && dy_dtype == TCCT_BF16) SL(bf16, bf16);
    else if (x_dtype == TCCT_F32 && dy_dtype == TCCT_F32) SL(float, float);
    else { tcct_set_error("pw_wgrad_smalln: bad dtypes"); return -1; }
#undef SL
    TCCT_LAUNCH_OK();
}
