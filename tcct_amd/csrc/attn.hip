// Factorised attention with convolutional relative position encoding: reference nets/tcct.py:219-287 (ConvRelPosEnc) and :289-341
// (FactorAtt_ConvRelPosEnc), the token mixer the reference keeps commented out in MHCABlock (tcct.py:436-449) -- SURVEY 8(f)4.
//
//   qkv [B,N,3C] = Linear(x) (the pointwise MFMA GEMM), channel = which*C + head*Ch + ch  (tcct.py:316-318)
//   P = softmax_N(k)                               (tcct.py:321)       k_fatt_kstats_* : online max / sum over the tokens
//   M[b,h] = P^T v   [Ch x Ch]                     (tcct.py:322)       k_fatt_ktv      : reduction over N
//   out = scale * q M + q * (dwconv_{3,5,7}(v)+b)  (tcct.py:323-330)   k_dwk + k_fatt_apply_fwd
//
// The contractions are block-diagonal per head with Ch = C/8 = 8..20: far below an MFMA tile (a 32x32x16 tile would be >= 84 % padding)
// and ~Ch FMAs per element moved, so they run on the VALU with M in LDS and are bound by HBM like the rest of the ViT branch; the two
// dense contractions of the module (qkv and proj Linear layers) are the MFMA pointwise GEMMs of pw_mfma.hip.
// Functional tier: coalesced 4-channel vector accesses, fp32 accumulation, no tuning beyond that.
#include "common.h"

#define FA_TR 32            // token rows staged per tile of k_fatt_ktv

// ------------------------------------------------------------------ softmax statistics of k over the tokens
// part [B,S,C,2] = (max, sum exp(k - max)) of segment s; element (b,n,c) of k at k[(b*N+n)*ld + c]
template <typename T>
__global__ void __launch_bounds__(256) k_fatt_kstats_partial(const T* __restrict__ k, int64_t ld, float* __restrict__ part, int N, int C,
                                                              int rows_per_seg) {
    __shared__ float sm_m[256], sm_s[256];
    const int b = blockIdx.y, seg = blockIdx.x, S = gridDim.x;
    const int R = 256 / C;                       // token rows walked in parallel (C <= 256)
    const int t = threadIdx.x, c = t % C, r = t / C;
    float m = -INFINITY, s = 0.f;
    if (r < R) {
        const int n1 = min(N, (seg + 1) * rows_per_seg);
        const T* p = k + (int64_t)b * N * ld + c;
        // four rows per step: the loads are independent of the running (max, sum) and go out together
        for (int n = seg * rows_per_seg + r; n < n1; n += 4 * R) {
            float xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = (n + u * R < n1) ? ldf(p + (int64_t)(n + u * R) * ld) : -INFINITY;
            const float mx = fmaxf(fmaxf(xv[0], xv[1]), fmaxf(xv[2], xv[3]));
            if (mx > -INFINITY) {
                const float mn = fmaxf(m, mx);
                s = s * __expf(m - mn) + __expf(xv[0] - mn) + __expf(xv[1] - mn) + __expf(xv[2] - mn) + __expf(xv[3] - mn);
                m = mn;
            }
        }
    }
    sm_m[t] = m; sm_s[t] = s;
    __syncthreads();
    if (r == 0) {
        float M = m;
        for (int i = 1; i < R; ++i) M = fmaxf(M, sm_m[i * C + c]);
        float Ssum = 0.f;
        for (int i = 0; i < R; ++i) { const float mi = sm_m[i * C + c]; if (mi > -INFINITY) Ssum += sm_s[i * C + c] * __expf(mi - M); }
        float* o = part + (((int64_t)b * S + seg) * C + c) * 2;
        o[0] = M; o[1] = Ssum;
    }
}
// stats [B,C,2] = (max, 1 / sum); grid (B, C/4), one wave per channel, lanes over the segments
__global__ void __launch_bounds__(256) k_fatt_kstats_final(const float* __restrict__ part, float* __restrict__ stats, int S, int C) {
    const int b = blockIdx.x, c = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;         // whole waves leave together
    float M = -INFINITY;
    for (int s = lane; s < S; s += 64) M = fmaxf(M, part[(((int64_t)b * S + s) * C + c) * 2]);
    M = wave_max(M);
    float sum = 0.f;
    for (int s = lane; s < S; s += 64) {
        const float* p = part + (((int64_t)b * S + s) * C + c) * 2;
        if (p[0] > -INFINITY) sum += p[1] * __expf(p[0] - M);
    }
    sum = wave_sum(sum);
    if (lane == 0) {
        stats[((int64_t)b * C + c) * 2] = M;
        stats[((int64_t)b * C + c) * 2 + 1] = 1.f / sum;
    }
}

// ------------------------------------------------------------------ out[b, c = h*Ch+k, v] += alpha * sum_n f(A[b,n,c]) * Bm[b,n,h*Ch+v]
// f = softmax over n (from stats) when SOFTMAX, identity otherwise.  Used for M = P^T v (forward) and dM = scale q^T dout (backward).
template <typename TA, typename TB, bool SOFTMAX>
__global__ void __launch_bounds__(256) k_fatt_ktv(const TA* __restrict__ A, int64_t lda, const TB* __restrict__ Bm, int64_t ldb,
                                                  const float* __restrict__ stats, float* __restrict__ out, float alpha, int N, int C, int Ch) {
    extern __shared__ __align__(16) float sm[];
    float* sa = sm;                     // [FA_TR][C]
    float* sb = sm + FA_TR * C;         // [FA_TR][C]
    float* sst = sb + FA_TR * C;        // [C][2]
    const int b = blockIdx.y, t = threadIdx.x;
    const int V4 = Ch >> 2, nitems = C * V4, C4 = C >> 2;
    if (SOFTMAX) for (int i = t; i < 2 * C; i += 256) sst[i] = stats[(int64_t)b * C * 2 + i];
    f4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f4zero();
    const int ntiles = (N + FA_TR - 1) / FA_TR;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n0 = tile * FA_TR;
        __syncthreads();
        for (int i = t; i < FA_TR * C4; i += 256) {
            const int r = i / C4, c4 = (i - r * C4) * 4, n = n0 + r;
            f4 a = f4zero(), bv = f4zero();
            if (n < N) {
                a = ld4(A + ((int64_t)b * N + n) * lda + c4);
                bv = ld4(Bm + ((int64_t)b * N + n) * ldb + c4);
                if (SOFTMAX) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a.v[e] = __expf(a.v[e] - sst[(c4 + e) * 2]) * sst[(c4 + e) * 2 + 1];
                }
            }
            *reinterpret_cast<float4*>(sa + r * C + c4) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
            *reinterpret_cast<float4*>(sb + r * C + c4) = make_float4(bv.v[0], bv.v[1], bv.v[2], bv.v[3]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = t + 256 * j;
            if (o < nitems) {
                const int c = o / V4, v4 = o - c * V4, col = (c / Ch) * Ch + v4 * 4;
#pragma unroll 8
                for (int r = 0; r < FA_TR; ++r) {
                    const float a = sa[r * C + c];
                    const float4 bv = *reinterpret_cast<const float4*>(sb + r * C + col);
                    acc[j].v[0] += a * bv.x; acc[j].v[1] += a * bv.y; acc[j].v[2] += a * bv.z; acc[j].v[3] += a * bv.w;
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int o = t + 256 * j;
        if (o < nitems) {
            const int c = o / V4, v4 = o - c * V4;
            float* dst = out + ((int64_t)b * C + c) * Ch + v4 * 4;
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dst + e, alpha * acc[j].v[e]);
        }
    }
}

// ------------------------------------------------------------------ depthwise K x K 'same' convolution on a channel group of strided NHWC rows
// y[pix, c] (+)= bias[c] + sum_taps w[c, tap] * x[pix + tap - K/2, c]; FLIP = 180-degree rotated taps (the input gradient); w is OIHW [Cg,1,K,K]
template <typename T, int K, bool FLIP, bool ACC>
__global__ void __launch_bounds__(256) k_dwk(const T* __restrict__ x, int64_t ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                             T* __restrict__ y, int64_t ldy, int B, int H, int W, int Cg) {
    extern __shared__ __align__(16) float sw[];       // [K*K][Cg]
    for (int i = threadIdx.x; i < K * K * Cg; i += 256) {
        const int tap = i / Cg, c = i - tap * Cg;
        sw[i] = w[c * K * K + (FLIP ? K * K - 1 - tap : tap)];
    }
    __syncthreads();
    const int C4 = Cg >> 2;
    const int64_t total = (int64_t)B * H * W * C4;
    constexpr int R = K / 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4) * 4;
        const int pix = (int)(i / C4);
        const int x0 = pix % W, y0 = (pix / W) % H;
        f4 acc = f4zero();
        if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc.v[e] = bias[c4 + e];
        }
        // K <= 5, branch-free taps: out-of-image taps read the centre pixel (always valid) and are zeroed, so that the K loads of a window
        // row are independent straight-line code the compiler issues together (3x3: 181 -> 146 us, 5x5: 364 -> 323 us at level 1 of the
        // bench shape); the 7x7 window is faster with skipped taps (671 vs 796 us)
        if constexpr (K > 5) {
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int yy = y0 + ky - R;
                if (yy < 0 || yy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int xx = x0 + kx - R;
                    if (xx < 0 || xx >= W) continue;
                    const f4 xv = ld4(x + (int64_t)(pix + (ky - R) * W + (kx - R)) * ldx + c4);
                    const float4 wv = *reinterpret_cast<const float4*>(sw + (ky * K + kx) * Cg + c4);
                    acc.v[0] += wv.x * xv.v[0]; acc.v[1] += wv.y * xv.v[1]; acc.v[2] += wv.z * xv.v[2]; acc.v[3] += wv.w * xv.v[3];
                }
            }
        } else {
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const int yy = y0 + ky - R;
                const bool oky = yy >= 0 && yy < H;
                f4 xv[K];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int xx = x0 + kx - R;
                    const bool ok = oky && xx >= 0 && xx < W;
                    xv[kx] = ld4(x + (int64_t)(ok ? pix + (ky - R) * W + (kx - R) : pix) * ldx + c4);
                    if (!ok) xv[kx] = f4zero();
                }
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float4 wv = *reinterpret_cast<const float4*>(sw + (ky * K + kx) * Cg + c4);
                    acc.v[0] += wv.x * xv[kx].v[0]; acc.v[1] += wv.y * xv[kx].v[1]; acc.v[2] += wv.z * xv[kx].v[2]; acc.v[3] += wv.w * xv[kx].v[3];
                }
            }
        }
        T* dst = y + (int64_t)pix * ldy + c4;
        if (ACC) {
            const f4 old = ld4(dst);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc.v[e] += old.v[e];
        }
        st4(dst, acc);
    }
}

// dw[c, tap] += sum_pix dy[pix, c] * x[pix + tap - K/2, c], db[c] += sum_pix dy[pix, c].
// Block = 32 channel lanes x 8 image rows of one column segment; a thread walks its row from left to right with the K x K window of x
// in registers (K new loads per pixel instead of K*K), sums its K*K products in registers, the 8 rows are combined by one wave shuffle and
// an LDS pass, and the block issues ONE atomic per element.  (The first version let 128 blocks walk 1 700 pixels each with K*K loads per
// pixel: 21 ms for the 7x7 group at level 1 of the bench shape; now 0.97 ms.  Tried without gain: 4 pixels per trip with their loads issued
// together -- 182 VGPRs, 1.57 ms.)
template <typename T, int K>
__global__ void __launch_bounds__(256) k_dwk_wgrad(const T* __restrict__ x, int64_t ldx, const T* __restrict__ dy, int64_t ldy,
                                                   float* __restrict__ dw, float* __restrict__ db, int B, int H, int W, int Cg, int segw) {
    constexpr int R = K / 2, NE = K * K + 1;
    __shared__ float red[4][NE][32];
    const int t = threadIdx.x, cl = t & 31, c = blockIdx.y * 32 + cl, j = t >> 5;
    const int nseg = (W + segw - 1) / segw, rows8 = (H + 7) / 8;
    int bid = blockIdx.x;
    const int seg = bid % nseg; bid /= nseg;
    const int rg = bid % rows8, b = bid / rows8;
    const int y0 = rg * 8 + j, xb = seg * segw, xe = min(W, xb + segw);
    const bool live = c < Cg && y0 < H;
    float acc[K][K], accb = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) acc[ky][kx] = 0.f;
    if (live) {
        // row pointers of the window (nullptr = outside the image: zero padding)
        const T* rowp[K];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int yy = y0 + ky - R;
            rowp[ky] = (yy >= 0 && yy < H) ? x + ((int64_t)(b * H + yy) * W) * ldx + c : nullptr;
        }
        const T* dyp = dy + ((int64_t)(b * H + y0) * W) * ldy + c;
        float win[K][K];        // win[ky][kx] = x[y0 + ky - R][xx + kx - R] once column xx + R has been shifted in
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            win[ky][0] = 0.f;
#pragma unroll
            for (int kx = 1; kx < K; ++kx) {
                const int xx = xb + kx - 1 - R;
                win[ky][kx] = (rowp[ky] && xx >= 0 && xx < W) ? ldf(rowp[ky] + (int64_t)xx * ldx) : 0.f;
            }
        }
        for (int xx = xb; xx < xe; ++xx) {
            const int xn = xx + R;
            const float g = ldf(dyp + (int64_t)xx * ldy);
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const float nv = (rowp[ky] && xn < W) ? ldf(rowp[ky] + (int64_t)xn * ldx) : 0.f;
#pragma unroll
                for (int kx = 0; kx < K - 1; ++kx) win[ky][kx] = win[ky][kx + 1];
                win[ky][K - 1] = nv;
#pragma unroll
                for (int kx = 0; kx < K; ++kx) acc[ky][kx] += g * win[ky][kx];
            }
            accb += g;
        }
    }
    // rows j and j+1 share a wave (lanes l and l+32 hold the same channel); then the four waves through LDS
    const int wv = t >> 6;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const float v = acc[ky][kx] + __shfl_xor(acc[ky][kx], 32, 64);
            if ((t & 32) == 0) red[wv][ky * K + kx][cl] = v;
        }
    {
        const float v = accb + __shfl_xor(accb, 32, 64);
        if ((t & 32) == 0) red[wv][K * K][cl] = v;
    }
    __syncthreads();
    if (c < Cg) {
        for (int e = j; e < NE; e += 8) {
            const float v = red[0][e][cl] + red[1][e][cl] + red[2][e][cl] + red[3][e][cl];
            if (e < K * K) atomicAdd(dw + c * K * K + e, v);
            else if (db) atomicAdd(db + c, v);
        }
    }
}

// ------------------------------------------------------------------ out = scale * q M + q * cv      (tcct.py:323-330, 284-285)
template <typename T>
__global__ void __launch_bounds__(256) k_fatt_apply_fwd(const T* __restrict__ qkv, const float* __restrict__ M, const T* __restrict__ cv,
                                                        T* __restrict__ out, float scale, int N, int C, int Ch) {
    extern __shared__ __align__(16) float sM[];       // [C][Ch]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C * Ch; i += 256) sM[i] = M[(int64_t)b * C * Ch + i];
    __syncthreads();
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / C4), c4 = (int)(i - (int64_t)n * C4) * 4;
        const int h0 = (c4 / Ch) * Ch, v = c4 - h0;
        const T* qrow = qkv + ((int64_t)b * N + n) * 3 * C;
        f4 acc = f4zero();
        for (int k = 0; k < Ch; k += 4) {
            const f4 q4 = ld4(qrow + h0 + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4 m4 = *reinterpret_cast<const float4*>(sM + (h0 + k + e) * Ch + v);
                acc.v[0] += q4.v[e] * m4.x; acc.v[1] += q4.v[e] * m4.y; acc.v[2] += q4.v[e] * m4.z; acc.v[3] += q4.v[e] * m4.w;
            }
        }
        const f4 q = ld4(qrow + c4);
        const f4 cc = ld4(cv + ((int64_t)b * N + n) * C + c4);
        f4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = scale * acc.v[e] + q.v[e] * cc.v[e];
        st4(out + ((int64_t)b * N + n) * C + c4, o);
    }
}

// ------------------------------------------------------------------ backward of the above w.r.t. q, k, v (attention part) and cv
//   dq   = scale * dout M^T + dout * cv
//   dv   = P dM                       (the crpe convolution adds its share afterwards: k_dwk<FLIP, ACC> on dcv)
//   dk   = P * (v dM^T - D),  D[c] = sum_v dM[c,v] M[c,v]   (= sum_n P dP, the softmax-over-N correction, without a pass over N)
//   dcv  = dout * q
// with dM = scale * q^T dout from k_fatt_ktv.
template <typename T>
__global__ void __launch_bounds__(256) k_fatt_apply_bwd(const T* __restrict__ qkv, const float* __restrict__ stats, const float* __restrict__ M,
                                                        const float* __restrict__ dM, const T* __restrict__ cv, const T* __restrict__ dout,
                                                        T* __restrict__ dqkv, T* __restrict__ dcv, float scale, int N, int C, int Ch) {
    extern __shared__ __align__(16) float sm[];
    float* sM = sm;                 // [C][Ch]
    float* sdM = sM + C * Ch;       // [C][Ch]
    float* sD = sdM + C * Ch;       // [C]
    float* sst = sD + C;            // [C][2]
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < C * Ch; i += 256) { sM[i] = M[(int64_t)b * C * Ch + i]; sdM[i] = dM[(int64_t)b * C * Ch + i]; }
    for (int i = threadIdx.x; i < 2 * C; i += 256) sst[i] = stats[(int64_t)b * C * 2 + i];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float d = 0.f;
        for (int v = 0; v < Ch; ++v) d += sdM[c * Ch + v] * sM[c * Ch + v];
        sD[c] = d;
    }
    __syncthreads();
    const int C4 = C >> 2;
    const int64_t total = (int64_t)N * C4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int n = (int)(i / C4), c4 = (int)(i - (int64_t)n * C4) * 4;
        const int h0 = (c4 / Ch) * Ch, j0 = c4 - h0;
        const int64_t row = (int64_t)b * N + n;
        const T* qrow = qkv + row * 3 * C;
        const T* krow = qrow + C;
        const T* vrow = qrow + 2 * C;
        const T* drow = dout + row * C;
        f4 dq = f4zero(), dkk = f4zero(), dv = f4zero();
        for (int u = 0; u < Ch; u += 4) {
            const f4 do4 = ld4(drow + h0 + u);
            const f4 v4 = ld4(vrow + h0 + u);
            const f4 k4 = ld4(krow + h0 + u);
#pragma unroll
            for (int e = 0; e < 4; ++e) {       // rows j0+e of M / dM, columns u..u+3
                const float4 m4 = *reinterpret_cast<const float4*>(sM + (h0 + j0 + e) * Ch + u);
                const float4 g4 = *reinterpret_cast<const float4*>(sdM + (h0 + j0 + e) * Ch + u);
                dq.v[e] += do4.v[0] * m4.x + do4.v[1] * m4.y + do4.v[2] * m4.z + do4.v[3] * m4.w;
                dkk.v[e] += v4.v[0] * g4.x + v4.v[1] * g4.y + v4.v[2] * g4.z + v4.v[3] * g4.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {       // rows u+e of dM, columns j0..j0+3
                const float p = __expf(k4.v[e] - sst[(h0 + u + e) * 2]) * sst[(h0 + u + e) * 2 + 1];
                const float4 g4 = *reinterpret_cast<const float4*>(sdM + (h0 + u + e) * Ch + j0);
                dv.v[0] += p * g4.x; dv.v[1] += p * g4.y; dv.v[2] += p * g4.z; dv.v[3] += p * g4.w;
            }
        }
        const f4 q = ld4(qrow + c4), kk = ld4(krow + c4), cc = ld4(cv + row * C + c4), dd = ld4(drow + c4);
        f4 oq, ok, oc;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float p = __expf(kk.v[e] - sst[(c4 + e) * 2]) * sst[(c4 + e) * 2 + 1];
            oq.v[e] = scale * dq.v[e] + dd.v[e] * cc.v[e];
            ok.v[e] = p * (dkk.v[e] - sD[c4 + e]);
            oc.v[e] = dd.v[e] * q.v[e];
        }
        T* orow = dqkv + row * 3 * C;
        st4(orow + c4, oq);
        st4(orow + C + c4, ok);
        st4(orow + 2 * C + c4, dv);
        st4(dcv + row * C + c4, oc);
    }
}

// ================================================================== C-ABI
static int fatt_shape_ok(const char* who, int B, int64_t N, int C, int heads) {
    TCCT_CHECK(B > 0 && N > 0 && C > 0 && heads > 0, "%s: empty shape B=%d N=%lld C=%d heads=%d", who, B, (long long)N, C, heads);
    TCCT_CHECK(C % heads == 0 && (C / heads) % 4 == 0, "%s: C=%d must be heads=%d x a multiple of 4", who, C, heads);
    TCCT_CHECK(C <= 256 && (int64_t)C * (C / heads) / 4 <= 1024, "%s: C=%d (Ch=%d) too large", who, C, C / heads);
    TCCT_CHECK((int64_t)B * N * 3 * C < ((int64_t)1 << 40) && N < ((int64_t)1 << 30), "%s: N=%lld too large", who, (long long)N);
    return 0;
}

// token segments of the statistics pass: short segments (>= 256 rows) so that thousands of blocks keep loads in flight; the final
// kernel combines <= 1024 partials per channel
static int fatt_segments(int64_t N) {
    int64_t S = (N + 255) / 256;
    if (S > 1024) S = 1024;
    if (S < 1) S = 1;
    return (int)S;
}
extern "C" int64_t tcct_fatt_kstats_workspace_bytes(int B, int64_t N, int C) {
    return (int64_t)B * fatt_segments(N) * C * 2 * sizeof(float);
}
extern "C" int tcct_fatt_kstats(const void* qkv, void* workspace, float* stats, int B, int64_t N, int C, int heads, int dtype,
                                tcct_stream_t stream) {
    if (fatt_shape_ok("fatt_kstats", B, N, C, heads)) return -1;
    TCCT_CHECK(qkv && workspace && stats, "fatt_kstats: NULL buffer");
    const int S = fatt_segments(N);
    const int rows = (int)((N + S - 1) / S);
    hipStream_t st = (hipStream_t)stream;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_fatt_kstats_partial<T>), dim3(S, B), dim3(256), 0, st, (const T*)qkv + C, (int64_t)3 * C,
                                            (float*)workspace, (int)N, C, rows));
    hipLaunchKernelGGL(k_fatt_kstats_final, dim3(B, (C + 3) / 4), dim3(256), 0, st, (const float*)workspace, stats, S, C);
    TCCT_LAUNCH_OK();
}

static int ktv_launch(const char* who, const void* A, int64_t lda, const void* Bm, int64_t ldb, const float* stats, float* out, float alpha,
                      int B, int64_t N, int C, int heads, int dtype, hipStream_t st) {
    if (fatt_shape_ok(who, B, N, C, heads)) return -1;
    const int Ch = C / heads;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(out, 0, sizeof(float) * (size_t)B * C * Ch, st) != hipSuccess) {
        tcct_set_error("%s: memset failed", who); return -2;
    }
    const int ntiles = (int)((N + FA_TR - 1) / FA_TR);
    const int gx = ntiles < 128 ? ntiles : 128;
    const size_t lds = sizeof(float) * (2 * FA_TR * C + 2 * C);
    if (stats) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_fatt_ktv<T, T, true>), dim3(gx, B), dim3(256), lds, st, (const T*)A, lda, (const T*)Bm, ldb, stats, out, alpha, (int)N, C, Ch)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_fatt_ktv<T, T, false>), dim3(gx, B), dim3(256), lds, st, (const T*)A, lda, (const T*)Bm, ldb, stats, out, alpha, (int)N, C, Ch)); }
    TCCT_LAUNCH_OK();
}
/* M[b,h,k,v] = sum_n softmax_N(k)[b,n,h,k] * v[b,n,h,v]  (fp32 [B,heads,Ch,Ch]) */
extern "C" int tcct_fatt_ktv(const void* qkv, const float* stats, float* M, int B, int64_t N, int C, int heads, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(qkv && stats && M, "fatt_ktv: NULL buffer");
    const int es = dtype == TCCT_BF16 ? 2 : 4;
    return ktv_launch("fatt_ktv", (const char*)qkv + (size_t)C * es, (int64_t)3 * C, (const char*)qkv + (size_t)2 * C * es, (int64_t)3 * C, stats, M, 1.f,
                      B, N, C, heads, dtype, (hipStream_t)stream);
}
/* dM[b,h,k,v] = scale * sum_n q[b,n,h,k] * dout[b,n,h,v] */
extern "C" int tcct_fatt_dktv(const void* qkv, const void* dout, float* dM, float scale, int B, int64_t N, int C, int heads, int dtype,
                              tcct_stream_t stream) {
    TCCT_CHECK(qkv && dout && dM, "fatt_dktv: NULL buffer");
    return ktv_launch("fatt_dktv", qkv, (int64_t)3 * C, dout, (int64_t)C, nullptr, dM, scale, B, N, C, heads, dtype, (hipStream_t)stream);
}

extern "C" int tcct_fatt_apply_fwd(const void* qkv, const float* M, const void* cv, void* out, float scale, int B, int64_t N, int C, int heads,
                                   int dtype, tcct_stream_t stream) {
    if (fatt_shape_ok("fatt_apply_fwd", B, N, C, heads)) return -1;
    TCCT_CHECK(qkv && M && cv && out, "fatt_apply_fwd: NULL buffer");
    const int Ch = C / heads;
    const int gx = tcct_grid(N * (C / 4), 256, 2048);
    const size_t lds = sizeof(float) * C * Ch;
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_fatt_apply_fwd<T>), dim3(gx, B), dim3(256), lds, (hipStream_t)stream, (const T*)qkv, M, (const T*)cv,
                                            (T*)out, scale, (int)N, C, Ch));
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_fatt_apply_bwd(const void* qkv, const float* stats, const float* M, const float* dM, const void* cv, const void* dout,
                                   void* dqkv, void* dcv, float scale, int B, int64_t N, int C, int heads, int dtype, tcct_stream_t stream) {
    if (fatt_shape_ok("fatt_apply_bwd", B, N, C, heads)) return -1;
    TCCT_CHECK(qkv && stats && M && dM && cv && dout && dqkv && dcv, "fatt_apply_bwd: NULL buffer");
    const int Ch = C / heads;
    const int gx = tcct_grid(N * (C / 4), 256, 2048);
    const size_t lds = sizeof(float) * (2 * C * Ch + 3 * C);
    TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_fatt_apply_bwd<T>), dim3(gx, B), dim3(256), lds, (hipStream_t)stream, (const T*)qkv, stats, M, dM,
                                            (const T*)cv, (const T*)dout, (T*)dqkv, (T*)dcv, scale, (int)N, C, Ch));
    TCCT_LAUNCH_OK();
}

static int dwk_shape_ok(const char* who, int B, int H, int W, int Cg, int K, int64_t ldx, int64_t ldy) {
    TCCT_CHECK(B > 0 && H > 0 && W > 0 && Cg > 0, "%s: empty shape", who);
    TCCT_CHECK(K == 3 || K == 5 || K == 7, "%s: window %d not in {3,5,7}", who, K);
    TCCT_CHECK(Cg % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldx >= Cg && ldy >= Cg, "%s: Cg=%d ldx=%lld ldy=%lld must be multiples of 4 with ld >= Cg",
               who, Cg, (long long)ldx, (long long)ldy);
    TCCT_CHECK((int64_t)B * H * W < ((int64_t)1 << 31) && Cg <= 256, "%s: shape too large", who);
    return 0;
}
/* depthwise K x K (K in 3,5,7), stride 1, 'same' zero padding on a channel group: x / y point at the group's first channel of NHWC rows with
 * pixel strides ldx / ldy (elements); w fp32 OIHW [Cg,1,K,K]; bias fp32 [Cg] or NULL; flip = rotate the taps by 180 degrees (input gradient);
 * accumulate = add to y */
extern "C" int tcct_dwk_strided_fwd(const void* x, int64_t ldx, const float* w, const float* bias, void* y, int64_t ldy, int B, int H, int W, int Cg,
                                    int K, int flip, int accumulate, int dtype, tcct_stream_t stream) {
    if (dwk_shape_ok("dwk_strided_fwd", B, H, W, Cg, K, ldx, ldy)) return -1;
    TCCT_CHECK(x && w && y, "dwk_strided_fwd: NULL buffer");
    const int gx = tcct_grid((int64_t)B * H * W * (Cg / 4), 256, 4096);
    const size_t lds = sizeof(float) * K * K * Cg;
    hipStream_t st = (hipStream_t)stream;
#define DWK(KK, F, A) hipLaunchKernelGGL((k_dwk<T, KK, F, A>), dim3(gx), dim3(256), lds, st, (const T*)x, ldx, w, bias, (T*)y, ldy, B, H, W, Cg)
#define DWK_K(KK)                                               \
    do {                                                        \
        if (flip && accumulate) DWK(KK, true, true);            \
        else if (flip) DWK(KK, true, false);                    \
        else if (accumulate) DWK(KK, false, true);              \
        else DWK(KK, false, false);                             \
    } while (0)
    TCCT_DISPATCH(dtype, if (K == 3) DWK_K(3); else if (K == 5) DWK_K(5); else DWK_K(7));
#undef DWK_K
#undef DWK
    TCCT_LAUNCH_OK();
}
/* dw [Cg,1,K,K] += sum dy * shifted x, dbias [Cg] += sum dy (fp32; cleared here unless tcct_set_outputs_prezeroed(1)) */
extern "C" int tcct_dwk_strided_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* dbias, int B, int H, int W, int Cg,
                                      int K, int dtype, tcct_stream_t stream) {
    if (dwk_shape_ok("dwk_strided_wgrad", B, H, W, Cg, K, ldx, ldy)) return -1;
    TCCT_CHECK(x && dy && dw, "dwk_strided_wgrad: NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill()) {
        if (hipMemsetAsync(dw, 0, sizeof(float) * Cg * K * K, st) != hipSuccess) { tcct_set_error("dwk_strided_wgrad: memset failed"); return -2; }
        if (dbias && hipMemsetAsync(dbias, 0, sizeof(float) * Cg, st) != hipSuccess) { tcct_set_error("dwk_strided_wgrad: memset failed"); return -2; }
    }
    const int rows8 = (H + 7) / 8;
    int segw = 256;             // column segment per block: as long as possible (halo = K-1 columns) while the grid still fills the chip
    while (segw > 32 && (int64_t)B * rows8 * ((W + segw - 1) / segw) * ((Cg + 31) / 32) < 1024) segw >>= 1;
    const int64_t nblk = (int64_t)B * rows8 * ((W + segw - 1) / segw);
    TCCT_CHECK(nblk < ((int64_t)1 << 31), "dwk_strided_wgrad: grid too large");
    dim3 grid((unsigned)nblk, (Cg + 31) / 32);
#define DWKW(KK) hipLaunchKernelGGL((k_dwk_wgrad<T, KK>), grid, dim3(256), 0, st, (const T*)x, ldx, (const T*)dy, ldy, dw, dbias, B, H, W, Cg, segw)
    TCCT_DISPATCH(dtype, if (K == 3) DWKW(3); else if (K == 5) DWKW(5); else DWKW(7));
#undef DWKW
    TCCT_LAUNCH_OK();
}
