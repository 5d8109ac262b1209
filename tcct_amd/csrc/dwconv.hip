// Depthwise 3x3 convolution (pad 1, stride 1 or 2) on NHWC, HBM-bound VALU stencil.  VEC=4 channels per lane when
// C % 4 == 0, scalar otherwise (the 1-channel lap_map convs of the boundary-regression loss).
//
// All kernels march DOWN the image: a thread owns (output column, channel vector) and a strip of output rows, keeps the
// 3x3 input window in registers (three row buffers rotated by unrolling, a fourth one prefetching the next row) and the nine
// taps of its channels in registers.  Lanes of a wave cover consecutive channel vectors of consecutive pixels, so every
// load/store instruction touches one contiguous run of a row; the left/right neighbours a lane needs are the centre loads
// of the adjacent lanes (L1 hits).  Every input row is fetched from HBM once per strip (+2 halo rows per strip); the first
// version slid along W instead, one output row per thread, and fetched every input row three times from three different
// XCDs (1.4-1.8 TB/s effective).
#include "common.h"
#include <cstdlib>
#include <type_traits>

#define DB 256
// (forcing 4 waves per SIMD with __launch_bounds__(256, 4) spills ~10 registers in the marching loop: 0.166 -> 0.262 ms at level 1)

template <typename T, int VEC>
__device__ __forceinline__ void ldv(const T* p, float* o) {
    if (VEC == 4) { f4 a = ld4(p); o[0] = a.v[0]; o[1] = a.v[1]; o[2] = a.v[2]; o[3] = a.v[3]; }
    else o[0] = ldf(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void stv(T* p, const float* o) {
    if (VEC == 4) { f4 a; a.v[0] = o[0]; a.v[1] = o[1]; a.v[2] = o[2]; a.v[3] = o[3]; st4(p, a); }
    else stf(p, o[0]);
}

// block -> (image n, row strip hs, column block wb); thread -> (column offset, channel vector).  false: thread has no work.
struct DwPos { int n, ho0, ho1, wo, c; };
// cpt: consecutive output columns per thread (p.wo = the first).  With one column per thread every input element is requested by
// three threads (8-byte requests, served by L1 but paid for in the address / tag pipeline): 2.7 TB/s at level 1.  Four columns per
// thread load six columns for four outputs.
__device__ __forceinline__ bool dw_pos(int C, int VEC, int Ho, int Wo, int segh, int wblocks, int hstrips, DwPos& p, int cpt = 1) {
    const int CV = C / VEC, PW = DB / CV, t = threadIdx.x;
    if (t >= PW * CV) return false;
    // XCD-banded block order (common.h): vertically adjacent strips share their halo rows -- with the hardware's round-robin order they were
    // fetched through different L2s (PMC: the stride-1 forward read 1.5x what it wrote)
    int bid = (int)xcd_band(blockIdx.x, gridDim.x);
    const int wb = bid % wblocks; bid /= wblocks;
    const int hs = bid % hstrips;
    p.n = bid / hstrips;
    p.wo = (wb * PW + t / CV) * cpt;
    p.c = (t % CV) * VEC;
    p.ho0 = hs * segh;
    p.ho1 = min(Ho, p.ho0 + segh);
    return p.wo < Wo && p.ho0 < p.ho1;
}

// raw (as loaded) channel vector: kept packed in registers and unpacked at use, so that a chunk of rows fits in few VGPRs
template <typename T, int VEC> struct Raw;
typedef __attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int dw_u32x2;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int dw_u32x4;
#define DW_OOB 0x80000000u      // a byte offset beyond every buffer these kernels bind: the load returns zeros, the store is dropped
template <> struct Raw<bf16, 4> {
    uint2 v;
    __device__ __forceinline__ void load(const bf16* p) { v = *reinterpret_cast<const uint2*>(p); }
    __device__ __forceinline__ void loadb(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        const dw_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0); v.x = t[0]; v.y = t[1];
    }
    __device__ __forceinline__ void zero() { v.x = 0u; v.y = 0u; }
    __device__ __forceinline__ float get(int k) const {
        const uint32_t w = k < 2 ? v.x : v.y;
        return (k & 1) ? __uint_as_float(w & 0xffff0000u) : __uint_as_float(w << 16);
    }
};
template <> struct Raw<float, 4> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void loadb(__amdgpu_buffer_rsrc_t r, uint32_t off) {
        const dw_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        v = make_float4(__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3]));
    }
    __device__ __forceinline__ void zero() { v = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ float get(int k) const { return k == 0 ? v.x : (k == 1 ? v.y : (k == 2 ? v.z : v.w)); }
};
template <typename T> struct Raw<T, 1> {
    float v;
    __device__ __forceinline__ void load(const T* p) { v = ldf(p); }
    __device__ __forceinline__ void loadb(__amdgpu_buffer_rsrc_t, uint32_t) { v = 0.f; }        // (the scalar kernels keep pointer loads)
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ float get(int) const { return v; }
};

__device__ __forceinline__ void dw_storeb(__amdgpu_buffer_rsrc_t r, uint32_t off, const float* o, const bf16*) {
    dw_u32x2 t; t[0] = pack_bf16x2(o[0], o[1]); t[1] = pack_bf16x2(o[2], o[3]);
    __builtin_amdgcn_raw_buffer_store_b64(t, r, off, 0, 0);
}
__device__ __forceinline__ void dw_storeb(__amdgpu_buffer_rsrc_t r, uint32_t off, const float* o, const float*) {
    dw_u32x4 t; t[0] = __float_as_uint(o[0]); t[1] = __float_as_uint(o[1]); t[2] = __float_as_uint(o[2]); t[3] = __float_as_uint(o[3]);
    __builtin_amdgcn_raw_buffer_store_b128(t, r, off, 0, 0);
}
// one input row: the three columns wi0, wi0+1, wi0+2 of input row hi (zeros outside the image)
template <typename T, int VEC, int NC = 3>
__device__ __forceinline__ void dw_load_row(Raw<T, VEC> (&r)[NC], const T* __restrict__ img, int hi, int H, int W, int C, int wi0) {
    const bool rok = hi >= 0 && hi < H;
    const T* row = img + (int64_t)(rok ? hi : 0) * W * C;
#pragma unroll
    for (int kx = 0; kx < NC; ++kx) {
        const int wi = wi0 + kx;
        if (rok && wi >= 0 && wi < W) r[kx].load(row + (int64_t)wi * C);
        else r[kx].zero();
    }
}

// Chunked marching window: a chunk produces RB output rows from CARRY rows kept from the previous chunk + NEW = RB*STRIDE
// freshly loaded rows; the NEW rows of the *next* chunk are requested before the current chunk is computed, so 3*NEW
// independent loads per lane are in flight during the arithmetic (a one-row-ahead window left the kernel latency-bound).
template <int STRIDE> struct DwChunk {
    static constexpr int RB = STRIDE == 1 ? 4 : 2;      // output rows per chunk
    static constexpr int NEW = RB * STRIDE;             // input rows loaded per chunk
    static constexpr int CARRY = 3 - STRIDE;            // input rows shared with the previous chunk
    static constexpr int ROWS = CARRY + NEW;
};

// stats != NULL (VEC == 4 only): also accumulate per-channel sum / sum of squares of y AS STORED into stats[0..C) / stats[C..2C) (fp64, zero on
// entry) -- the statistics of the train-mode BatchNorm behind the convolution (InvRes: dwconv -> norm, reference nets/tcct.py:548-551), which
// otherwise cost a separate pass over y
__device__ __forceinline__ float dw_rnd(float v, const bf16*) { return __bfloat162float(__float2bfloat16(v)); }
__device__ __forceinline__ float dw_rnd(float v, const float*) { return v; }
// Round 4: the convolution's input may be y_prev, the input of a train-mode BatchNorm + Hardswish whose normalisation pass was never run (xab = {a[C], b[C]}):
// z = hswish(a y_prev + b) is applied to every channel vector ONCE, when its row enters the register window, and rounded to the activation type -- the
// value the separate normalisation pass stored.  Elements outside the image stay zero (the convolution pads z, not y_prev).
__device__ __forceinline__ Raw<bf16, 4> dw_xf(const Raw<bf16, 4>& r, const float* xa, const float* xb, bool inside) {
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float z = act_c<TCCT_ACT_HSWISH>(xa[k] * r.get(k) + xb[k]); o[k] = inside ? z : 0.f; }
    Raw<bf16, 4> q;
    q.v.x = pack_bf16x2(o[0], o[1]); q.v.y = pack_bf16x2(o[2], o[3]);
    return q;
}
__device__ __forceinline__ Raw<float, 4> dw_xf(const Raw<float, 4>& r, const float* xa, const float* xb, bool inside) {
    float o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const float z = act_c<TCCT_ACT_HSWISH>(xa[k] * r.get(k) + xb[k]); o[k] = inside ? z : 0.f; }
    Raw<float, 4> q;
    q.v = make_float4(o[0], o[1], o[2], o[3]);
    return q;
}
template <typename T> __device__ __forceinline__ Raw<T, 1> dw_xf(const Raw<T, 1>& r, const float*, const float*, bool) { return r; }      // (4-channel vectors only)
// MODE (compile time: a run-time flag puts scalar branches into the marching loop and the compiler then waits for every load in flight at their joins):
// 0 plain; 1 `xab` = the pending BatchNorm + Hardswish of the input, applied as rows enter the window; 2 (input-gradient launches) `res` is y_prev of that
// BatchNorm and is NOT added -- its two backward sums are accumulated into `stats` (see below)
template <typename T, int VEC, int STRIDE, bool FLIP, int CPT = 1, int MODE = 0>
__global__ void __launch_bounds__(DB) k_dw_fwd(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                               T* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo, int add_input,
                                               int segh, int wblocks, int hstrips, const T* __restrict__ res, double* __restrict__ stats = nullptr,
                                               const float* __restrict__ xab = nullptr) {
    constexpr bool XAFF = MODE == 1, aux_mode = MODE == 2;
    // aux_mode (round 4, input-gradient launches): `res` is NOT added -- it is y_prev, the input of the train-mode BatchNorm + Hardswish (coefficients xab) whose
    // output this convolution consumed: the kernel accumulates that BatchNorm's two backward sums {sum dz', sum dz' y_prev}, dz' = dz hswish'(a y_prev + b) with
    // dz as stored, into `stats` (raw form, fp64 [2C]) -- its separate reduction pass (read dz, read y_prev) is gone
    typedef typename std::conditional<sizeof(T) == 4, double, float>::type SAcc;       // fp32 parity mode: fp64 partial sums, like tcct_bn_stats' block combine
    __shared__ SAcc s_red[VEC == 4 ? DB * 8 : 1];
    // res != NULL (output-shaped): y += res -- as input gradient: the gradient reaching the convolution's input through its other consumers
    // CPT > 1 (stride 1 only): the thread owns CPT consecutive output columns and loads CPT + 2 input columns per row
    static_assert(CPT == 1 || STRIDE == 1, "several columns per thread: stride 1 only");
    constexpr int NC = CPT + 2;
    typedef DwChunk<STRIDE> K;
    DwPos p;
    const bool has_work = dw_pos(C, VEC, Ho, Wo, segh, wblocks, hstrips, p, CPT);
    if (!has_work && !(VEC == 4 && stats)) return;
    SAcc st_s[VEC], st_q[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) st_s[k] = st_q[k] = 0;
    if (has_work) {
    float wk[9][VEC], bv[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        bv[k] = bias ? bias[p.c + k] : 0.f;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) wk[tp][k] = w[(p.c + k) * 9 + (FLIP ? 8 - tp : tp)];
    }
    const T* img = x + (int64_t)p.n * H * W * C + p.c;
    T* out = y + (int64_t)p.n * Ho * Wo * C + p.c;
    const T* rin = res ? res + (int64_t)p.n * Ho * Wo * C + p.c : nullptr;
    const int wi0 = p.wo * STRIDE - 1;
    constexpr int RBC = CPT == 1 ? K::RB : 2;              // output rows per chunk (two with several columns: register budget)
    constexpr int NEWC = RBC * STRIDE, ROWSC = K::CARRY + NEWC;
    Raw<T, VEC> R[ROWSC][NC], NX[NEWC][NC];
    if (VEC == 4) {
        // Channel vectors go through buffer descriptors (one per image): a pixel outside the image, a row of the next strip or a column
        // beyond Wo is an out-of-range offset -- zeros / dropped store -- so the marching loop has NO branch around a memory operation.
        // With `if (inside) load` hipcc cannot count the loads in flight and waits with vmcnt(0): the ISA of the pointer version waited
        // for the just-issued next-chunk rows before EVERY output pixel (its residual load sat in a branch), i.e. nothing overlapped.
        const uint32_t ES = sizeof(T), in_bytes = (uint32_t)H * W * C * ES, out_bytes = (uint32_t)Ho * Wo * C * ES;
        // (the image index is block-uniform, but it comes out of dw_pos() behind a per-thread early return: without readfirstlane the
        // compiler treats the descriptors as divergent and wraps every buffer access in a waterfall loop)
        const int nu = __builtin_amdgcn_readfirstlane(p.n);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)nu * H * W * C), 0, in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (int64_t)nu * Ho * Wo * C), 0, out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)((res ? res : y) + (int64_t)nu * Ho * Wo * C), 0, res ? out_bytes : 0u, 0x00020000);
        uint32_t cin[NC], cout[CPT];        // byte offset of the thread's channels in column wi0 + kx of an input row / wo + cc of an output row
#pragma unroll
        for (int kx = 0; kx < NC; ++kx) cin[kx] = (unsigned)(wi0 + kx) < (unsigned)W ? (uint32_t)((wi0 + kx) * C + p.c) * ES : DW_OOB;
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) cout[cc] = p.wo + cc < Wo ? (uint32_t)((p.wo + cc) * C + p.c) * ES : DW_OOB;
        auto load_row = [&](Raw<T, VEC> (&r)[NC], int hi, bool live) {
            const bool rok = live && (unsigned)hi < (unsigned)H;
            const uint32_t ro = (uint32_t)hi * (uint32_t)(W * C) * ES;
#pragma unroll
            for (int kx = 0; kx < NC; ++kx) r[kx].loadb(rx, (rok && cin[kx] != DW_OOB) ? ro + cin[kx] : DW_OOB);
        };
        Raw<T, VEC> RS[RBC][CPT], RSN[RBC][CPT];            // the second gradient (res) of the current / next chunk
        auto load_res = [&](Raw<T, VEC> (&q)[RBC][CPT], int ho) {
#pragma unroll
            for (int j = 0; j < RBC; ++j)
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc)
                    q[j][cc].loadb(rr, (ho + j < p.ho1 && cout[cc] != DW_OOB) ? (uint32_t)(ho + j) * (uint32_t)(Wo * C) * ES + cout[cc] : DW_OOB);
        };
#pragma unroll
        for (int i = 0; i < ROWSC; ++i) load_row(R[i], p.ho0 * STRIDE - 1 + i, true);
        float xa[VEC], xb[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) { xa[k] = MODE ? xab[p.c + k] : 1.f; xb[k] = MODE ? xab[C + p.c + k] : 0.f; }
        auto xf_row = [&](Raw<T, VEC> (&r)[NC], int hi, bool live) {        // the pending normalisation of a freshly loaded row (see dw_xf)
            const bool rok = live && (unsigned)hi < (unsigned)H;
#pragma unroll
            for (int kx = 0; kx < NC; ++kx) r[kx] = dw_xf(r[kx], xa, xb, rok && cin[kx] != DW_OOB);
        };
        if (XAFF) {      // the carried rows of the first chunk; its other rows -- like every later chunk's new rows -- are transformed at the top of the loop
#pragma unroll
            for (int i = 0; i < K::CARRY; ++i) xf_row(R[i], p.ho0 * STRIDE - 1 + i, true);
        }
        load_res(RS, p.ho0);
        for (int ho = p.ho0; ho < p.ho1; ho += RBC) {
            const bool more = ho + RBC < p.ho1;           // rows of a chunk beyond the strip are not fetched
#pragma unroll
            for (int i = 0; i < NEWC; ++i) load_row(NX[i], (ho + RBC) * STRIDE - 1 + K::CARRY + i, more);
            load_res(RSN, ho + RBC);
            if (XAFF) {  // AFTER the next chunk's loads are in flight: transforming at the copy below left the VALU work with nothing outstanding (0.115 -> 0.150 ms)
#pragma unroll
                for (int i = 0; i < NEWC; ++i) xf_row(R[K::CARRY + i], ho * STRIDE - 1 + K::CARRY + i, true);
            }
#pragma unroll
            for (int j = 0; j < RBC; ++j) {
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
                    float acc[VEC];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float a = bv[k] + (aux_mode ? 0.f : RS[j][cc].get(k));
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) a += R[j * STRIDE + ky][cc + kx].get(k) * wk[ky * 3 + kx][k];
                        if (add_input) a += R[j * STRIDE + 1][cc + 1].get(k);
                        acc[k] = a;
                    }
                    const bool live = ho + j < p.ho1 && cout[cc] != DW_OOB;
                    dw_storeb(ry, live ? (uint32_t)(ho + j) * (uint32_t)(Wo * C) * ES + cout[cc] : DW_OOB, acc, (const T*)nullptr);
                    if (stats && live) {
                        if (aux_mode) {
#pragma unroll
                            for (int k = 0; k < VEC; ++k) {
                                const float yv = RS[j][cc].get(k), u = xa[k] * yv + xb[k];
                                const float d = dw_rnd(acc[k], (const T*)nullptr) * (u < -3.f ? 0.f : (u <= 3.f ? (2.f * u + 3.f) * (1.f / 6.f) : 1.f));
                                st_s[k] += (SAcc)d; st_q[k] += (SAcc)d * (SAcc)yv;
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < VEC; ++k) { const SAcc r = dw_rnd(acc[k], (const T*)nullptr); st_s[k] += r; st_q[k] += r * r; }
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < K::CARRY; ++i)
#pragma unroll
                for (int kx = 0; kx < NC; ++kx) R[i][kx] = R[NEWC + i][kx];
#pragma unroll
            for (int i = 0; i < NEWC; ++i)
#pragma unroll
                for (int kx = 0; kx < NC; ++kx) R[K::CARRY + i][kx] = NX[i][kx];
#pragma unroll
            for (int j = 0; j < RBC; ++j)
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) RS[j][cc] = RSN[j][cc];
        }
    } else {
#pragma unroll
    for (int i = 0; i < ROWSC; ++i) dw_load_row<T, VEC, NC>(R[i], img, p.ho0 * STRIDE - 1 + i, H, W, C, wi0);
    for (int ho = p.ho0; ho < p.ho1; ho += RBC) {
        if (ho + RBC < p.ho1) {
#pragma unroll
            for (int i = 0; i < NEWC; ++i) dw_load_row<T, VEC, NC>(NX[i], img, (ho + RBC) * STRIDE - 1 + K::CARRY + i, H, W, C, wi0);
        }
#pragma unroll
        for (int j = 0; j < RBC; ++j) {
            if (ho + j < p.ho1) {
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
                    if (CPT > 1 && p.wo + cc >= Wo) break;
                    float acc[VEC];
                    Raw<T, VEC> rr;
                    if (rin) rr.load(rin + ((int64_t)(ho + j) * Wo + p.wo + cc) * C); else rr.zero();
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float a = bv[k] + rr.get(k);
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) a += R[j * STRIDE + ky][cc + kx].get(k) * wk[ky * 3 + kx][k];
                        if (add_input) a += R[j * STRIDE + 1][cc + 1].get(k);
                        acc[k] = a;
                    }
                    stv<T, VEC>(out + ((int64_t)(ho + j) * Wo + p.wo + cc) * C, acc);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < K::CARRY; ++i)
#pragma unroll
            for (int kx = 0; kx < NC; ++kx) R[i][kx] = R[NEWC + i][kx];
#pragma unroll
        for (int i = 0; i < NEWC; ++i)
#pragma unroll
            for (int kx = 0; kx < NC; ++kx) R[K::CARRY + i][kx] = NX[i][kx];
    }
    }       // VEC == 4 / scalar
    }       // has_work
    if (VEC == 4 && stats) {        // threads t, t + CV, t + 2 CV ... hold the same four channels: LDS, then one fp64 atomic per channel and block
        const int CV = C / VEC, PW = DB / CV, t = threadIdx.x;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { s_red[t * 8 + k] = st_s[k]; s_red[t * 8 + 4 + k] = st_q[k]; }
        __syncthreads();
        if (t < C) {
            const int cvv = t / VEC, k = t % VEC;
            double a = 0.0, b = 0.0;
            for (int pp = 0; pp < PW; ++pp) { a += (double)s_red[(pp * CV + cvv) * 8 + k]; b += (double)s_red[(pp * CV + cvv) * 8 + 4 + k]; }
            atomicAdd(&stats[t], a); atomicAdd(&stats[C + t], b);
        }
    }
}

// strip height: 32 output rows when that still leaves >= 1024 blocks, else 16, else 8 (the halo re-read is 2/segh)
static void dw_geometry(int N, int Ho, int Wo, int C, int vec, int target_blocks, int max_segh, int& segh, int& wblocks, int& hstrips,
                        int cpt = 1) {
    const int PW = DB / (C / vec) * cpt;
    wblocks = (Wo + PW - 1) / PW;
    segh = max_segh;
    while (segh > 8 && (int64_t)N * wblocks * ((Ho + segh - 1) / segh) < target_blocks) segh >>= 1;
    hstrips = (Ho + segh - 1) / segh;
}

template <typename T, int VEC, bool FLIP, int MODE = 0>
static void dw_fwd_launch(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C, int stride,
                          int Ho, int Wo, int add_input, hipStream_t st, const void* res = nullptr, double* stats = nullptr, const float* xab = nullptr) {
    int segh, wblocks, hstrips;
    // four columns per thread for the wide bf16 stride-1 images (levels 1-2 of the ViT branch); set cpt_on = 0 below to keep one column (A/B, rebuild)
    static int cpt_on = -1;
    if (cpt_on < 0) cpt_on = 1;
    if (stride == 1 && VEC == 4 && sizeof(T) == 2 && cpt_on && Wo >= 128 && C >= 32) {      // measured +4 % at 64 channels, slower at 4
        dw_geometry(N, Ho, Wo, C, VEC, 1024, 32, segh, wblocks, hstrips, 4);
        dim3 g4((unsigned)((int64_t)N * wblocks * hstrips));
        hipLaunchKernelGGL((k_dw_fwd<T, VEC, 1, FLIP, (VEC == 4 ? 4 : 1), MODE>), g4, dim3(DB), 0, st, (const T*)x, w, bias, (T*)y, N, H, W, C, Ho, Wo, add_input, segh, wblocks, hstrips, (const T*)res, stats, xab);
        return;
    }
    dw_geometry(N, Ho, Wo, C, VEC, 1024, 32, segh, wblocks, hstrips);
    dim3 g((unsigned)((int64_t)N * wblocks * hstrips)), b(DB);
    if (stride == 1) hipLaunchKernelGGL((k_dw_fwd<T, VEC, 1, FLIP, 1, MODE>), g, b, 0, st, (const T*)x, w, bias, (T*)y, N, H, W, C, Ho, Wo, add_input, segh, wblocks, hstrips, (const T*)res, stats, xab);
    else hipLaunchKernelGGL((k_dw_fwd<T, VEC, 2, FLIP, 1, MODE>), g, b, 0, st, (const T*)x, w, bias, (T*)y, N, H, W, C, Ho, Wo, add_input, segh, wblocks, hstrips, (const T*)res, stats, xab);
}

extern "C" int tcct_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C,
                                  int stride, int add_input, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "dwconv3x3_fwd: stride %d", stride);
    TCCT_CHECK(!(add_input && stride != 1), "dwconv3x3_fwd: add_input needs stride 1");
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1 && C >= 1, "dwconv3x3_fwd: empty tensor");
    int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    int vec = (C % 4 == 0) ? 4 : 1;
    TCCT_CHECK(C / vec <= DB, "dwconv3x3_fwd: C=%d too large", C);
    TCCT_CHECK((int64_t)H * W * C * 4 < (1ll << 31), "dwconv3x3_fwd: one image of %d x %d x %d elements exceeds the 32-bit byte offsets of the kernels", H, W, C);
    hipStream_t st = (hipStream_t)stream;
    if (vec == 4) { TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 4, false>(x, w, bias, y, N, H, W, C, stride, Ho, Wo, add_input, st))); }
    else { TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 1, false>(x, w, bias, y, N, H, W, C, stride, Ho, Wo, add_input, st))); }
    TCCT_LAUNCH_OK();
}
/* forward + fused statistics of the train-mode BatchNorm that consumes y (ResBlock: dwconv -> norm, reference nets/tcct.py:548-551): stats fp64
 * [2C] (zero on entry) += {sum, sum of squares} of y as stored.  C % 4 == 0, C <= 256. */
extern "C" int tcct_dwconv3x3_fwd_bnstats(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C, int stride,
                                          int add_input, double* stats, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "dwconv3x3_fwd_bnstats: stride %d", stride);
    TCCT_CHECK(!(add_input && stride != 1), "dwconv3x3_fwd_bnstats: add_input needs stride 1");
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1 && C >= 4 && C % 4 == 0 && C <= DB && stats != nullptr, "dwconv3x3_fwd_bnstats: needs C %% 4 == 0, C <= 256, stats");
    TCCT_CHECK((int64_t)H * W * C * 4 < (1ll << 31), "dwconv3x3_fwd_bnstats: one image of %d x %d x %d elements exceeds the 32-bit byte offsets of the kernels", H, W, C);
    int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 4, false>(x, w, bias, y, N, H, W, C, stride, Ho, Wo, add_input, (hipStream_t)stream, nullptr, stats)));
    TCCT_LAUNCH_OK();
}
/* forward with the normalisation of the train-mode BatchNorm + Hardswish IN FRONT applied on load (round 4): x is that BatchNorm's input y_prev, xab = {a[C], b[C]}
 * (z = hswish(a y_prev + b), rounded to the activation type, zero outside the image); stats nullable (as tcct_dwconv3x3_fwd_bnstats).  C % 4 == 0, C <= 256. */
extern "C" int tcct_dwconv3x3_fwd_xaff(const void* x, const float* xab, const float* w, const float* bias, void* y, int N, int H, int W, int C, int stride,
                                       double* stats, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "dwconv3x3_fwd_xaff: stride %d", stride);
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1 && C >= 4 && C % 4 == 0 && C <= DB && xab != nullptr, "dwconv3x3_fwd_xaff: needs C %% 4 == 0, C <= 256, xab");
    TCCT_CHECK((int64_t)H * W * C * 4 < (1ll << 31), "dwconv3x3_fwd_xaff: one image of %d x %d x %d elements exceeds the 32-bit byte offsets of the kernels", H, W, C);
    int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 4, false, 1>(x, w, bias, y, N, H, W, C, stride, Ho, Wo, 0, (hipStream_t)stream, nullptr, stats, xab)));
    TCCT_LAUNCH_OK();
}

// stride-2 input gradient.  With hi = 2a + pa, wi = 2b + pb the transposed convolution splits by parity:
//   dx[2a  ][2b  ] = dy[a][b] w11
//   dx[2a  ][2b+1] = dy[a][b] w12 + dy[a][b+1] w10
//   dx[2a+1][2b  ] = dy[a][b] w21 + dy[a+1][b] w01
//   dx[2a+1][2b+1] = dy[a][b] w22 + dy[a][b+1] w20 + dy[a+1][b] w02 + dy[a+1][b+1] w00
// A thread owns the 2x2 quad (a, b) of one channel vector and marches down a: two new dy loads per quad, four dx stores.
template <typename T, int VEC>
__global__ void __launch_bounds__(DB) k_dw_dgrad_s2(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                                    int N, int H, int W, int C, int Ho, int Wo, int segh, int wblocks, int hstrips,
                                                    const T* __restrict__ res) {
    // res != NULL ([N,H,W,C]): dx += res (gradient of the input's other consumers); quads cover a = 0 .. ceil(H/2)-1, b = 0 .. ceil(W/2)-1 (>= Ho, Wo when H or W is odd ... they are equal: Ho = ceil(H/2))
    const int Qh = (H + 1) / 2, Qw = (W + 1) / 2;
    DwPos p;
    if (!dw_pos(C, VEC, Qh, Qw, segh, wblocks, hstrips, p)) return;
    float wk[9][VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) wk[tp][k] = w[(p.c + k) * 9 + tp];
    const T* g = dy + (int64_t)p.n * Ho * Wo * C + p.c;
    T* out = dx + (int64_t)p.n * H * W * C + p.c;
    const T* rin = res ? res + (int64_t)p.n * H * W * C + p.c : nullptr;
    const int b = p.wo;
    if (VEC == 4) {         // branch-free memory operations through buffer descriptors (see k_dw_fwd)
        const uint32_t ES = sizeof(T), in_bytes = (uint32_t)H * W * C * ES;
        const int nu = __builtin_amdgcn_readfirstlane(p.n);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)nu * Ho * Wo * C), 0, (uint32_t)Ho * Wo * C * ES, 0x00020000);
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(dx + (int64_t)nu * H * W * C), 0, in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)((res ? res : dx) + (int64_t)nu * H * W * C), 0, res ? in_bytes : 0u, 0x00020000);
        const uint32_t cg0 = (uint32_t)(b * C + p.c) * ES, cg1 = b + 1 < Wo ? cg0 + (uint32_t)C * ES : DW_OOB;
        const uint32_t co0 = (uint32_t)(2 * b * C + p.c) * ES, co1 = 2 * b + 1 < W ? co0 + (uint32_t)C * ES : DW_OOB;
        auto ldb = [&](Raw<T, VEC> (&d)[2], int a) {
            const bool rok = a < Ho;
            const uint32_t r0 = (uint32_t)a * (uint32_t)(Wo * C) * ES;
            d[0].loadb(rg, rok ? r0 + cg0 : DW_OOB);
            d[1].loadb(rg, (rok && cg1 != DW_OOB) ? r0 + cg1 : DW_OOB);
        };
        auto ldr = [&](Raw<T, VEC> (&q)[4], int a, bool live) {       // the second gradient of the quad's four pixels
            const uint32_t r0 = (uint32_t)(2 * a) * (uint32_t)(W * C) * ES, r1 = r0 + (uint32_t)(W * C) * ES;
            const bool row1 = live && 2 * a + 1 < H;
            q[0].loadb(rr, live ? r0 + co0 : DW_OOB);
            q[1].loadb(rr, (live && co1 != DW_OOB) ? r0 + co1 : DW_OOB);
            q[2].loadb(rr, row1 ? r1 + co0 : DW_OOB);
            q[3].loadb(rr, (row1 && co1 != DW_OOB) ? r1 + co1 : DW_OOB);
        };
        auto emitb = [&](const Raw<T, VEC> (&u)[2], const Raw<T, VEC> (&v)[2], const Raw<T, VEC> (&q)[4], int a, bool live) {
            float o00[VEC], o01[VEC], o10[VEC], o11[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float u0 = u[0].get(k), u1 = u[1].get(k), v0 = v[0].get(k), v1 = v[1].get(k);
                o00[k] = u0 * wk[4][k] + q[0].get(k);
                o01[k] = u0 * wk[5][k] + u1 * wk[3][k] + q[1].get(k);
                o10[k] = u0 * wk[7][k] + v0 * wk[1][k] + q[2].get(k);
                o11[k] = u0 * wk[8][k] + u1 * wk[6][k] + v0 * wk[2][k] + v1 * wk[0][k] + q[3].get(k);
            }
            const uint32_t r0 = (uint32_t)(2 * a) * (uint32_t)(W * C) * ES, r1 = r0 + (uint32_t)(W * C) * ES;
            const bool row1 = live && 2 * a + 1 < H;
            dw_storeb(ro, live ? r0 + co0 : DW_OOB, o00, (const T*)nullptr);
            dw_storeb(ro, (live && co1 != DW_OOB) ? r0 + co1 : DW_OOB, o01, (const T*)nullptr);
            dw_storeb(ro, row1 ? r1 + co0 : DW_OOB, o10, (const T*)nullptr);
            dw_storeb(ro, (row1 && co1 != DW_OOB) ? r1 + co1 : DW_OOB, o11, (const T*)nullptr);
        };
        // two quad rows per iteration, the dy rows and the second gradient of the NEXT pair requested before this pair is computed
        Raw<T, VEC> d0[2], d1[2], d2[2], n1[2], n2[2], q0[4], q1[4], m0[4], m1[4];
        ldb(d0, p.ho0); ldb(d1, p.ho0 + 1); ldb(d2, p.ho0 + 2);
        ldr(q0, p.ho0, true); ldr(q1, p.ho0 + 1, p.ho0 + 1 < p.ho1);
        for (int a = p.ho0; a < p.ho1; a += 2) {
            const bool more = a + 2 < p.ho1;
            ldb(n1, more ? a + 3 : Ho); ldb(n2, more ? a + 4 : Ho);
            ldr(m0, a + 2, more); ldr(m1, a + 3, a + 3 < p.ho1);
            emitb(d0, d1, q0, a, true);
            emitb(d1, d2, q1, a + 1, a + 1 < p.ho1);
#pragma unroll
            for (int j = 0; j < 2; ++j) { d0[j] = d2[j]; d1[j] = n1[j]; d2[j] = n2[j]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) { q0[j] = m0[j]; q1[j] = m1[j]; }
        }
        return;
    }
    auto ld = [&](float (&d)[2][VEC], int a) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (a < Ho && b + j < Wo) ldv<T, VEC>(g + ((int64_t)a * Wo + b + j) * C, d[j]);
            else {
#pragma unroll
                for (int k = 0; k < VEC; ++k) d[j][k] = 0.f;
            }
        }
    };
    auto emit = [&](const float (&u)[2][VEC], const float (&v)[2][VEC], int a) {      // u = dy[a][b..b+1], v = dy[a+1][b..b+1]
        float o00[VEC], o01[VEC], o10[VEC], o11[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            o00[k] = u[0][k] * wk[4][k];
            o01[k] = u[0][k] * wk[5][k] + u[1][k] * wk[3][k];
            o10[k] = u[0][k] * wk[7][k] + v[0][k] * wk[1][k];
            o11[k] = u[0][k] * wk[8][k] + u[1][k] * wk[6][k] + v[0][k] * wk[2][k] + v[1][k] * wk[0][k];
        }
        const int hi = 2 * a, wi = 2 * b;
        T* o = out + ((int64_t)hi * W + wi) * C;
        if (rin) {
            const T* ri = rin + ((int64_t)hi * W + wi) * C;
            float r[VEC];
            ldv<T, VEC>(ri, r);
#pragma unroll
            for (int k = 0; k < VEC; ++k) o00[k] += r[k];
            if (wi + 1 < W) {
                ldv<T, VEC>(ri + C, r);
#pragma unroll
                for (int k = 0; k < VEC; ++k) o01[k] += r[k];
            }
            if (hi + 1 < H) {
                ldv<T, VEC>(ri + (int64_t)W * C, r);
#pragma unroll
                for (int k = 0; k < VEC; ++k) o10[k] += r[k];
                if (wi + 1 < W) {
                    ldv<T, VEC>(ri + (int64_t)W * C + C, r);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) o11[k] += r[k];
                }
            }
        }
        stv<T, VEC>(o, o00);
        if (wi + 1 < W) stv<T, VEC>(o + C, o01);
        if (hi + 1 < H) {
            stv<T, VEC>(o + (int64_t)W * C, o10);
            if (wi + 1 < W) stv<T, VEC>(o + (int64_t)W * C + C, o11);
        }
    };
    float d0[2][VEC], d1[2][VEC];
    ld(d0, p.ho0);
    for (int a = p.ho0; a < p.ho1; a += 2) {
        ld(d1, a + 1);
        emit(d0, d1, a);
        if (a + 1 >= p.ho1) break;
        ld(d0, a + 2);
        emit(d1, d0, a + 1);
    }
}

static int dw_dgrad_impl(const void* dy, const float* w, const void* res, void* dx, int N, int H, int W, int C, int stride,
                         int add_input, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(stride == 1 || stride == 2, "dwconv3x3_dgrad: stride %d", stride);
    TCCT_CHECK(!(add_input && stride != 1), "dwconv3x3_dgrad: add_input needs stride 1");
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1 && C >= 1, "dwconv3x3_dgrad: empty tensor");
    int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    int vec = (C % 4 == 0) ? 4 : 1;
    TCCT_CHECK(C / vec <= DB, "dwconv3x3_dgrad: C=%d too large", C);
    TCCT_CHECK((int64_t)H * W * C * 4 < (1ll << 31), "dwconv3x3_dgrad: one image of %d x %d x %d elements exceeds the 32-bit byte offsets of the kernels", H, W, C);
    hipStream_t st = (hipStream_t)stream;
    if (stride == 1) {      // dx = conv(dy, flipped taps) (+ dy when the forward added its input) (+ res)
        if (vec == 4) { TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 4, true>(dy, w, nullptr, dx, N, H, W, C, 1, H, W, add_input, st, res))); }
        else { TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 1, true>(dy, w, nullptr, dx, N, H, W, C, 1, H, W, add_input, st, res))); }
        TCCT_LAUNCH_OK();
    }
    int segh, wblocks, hstrips;
    dw_geometry(N, (H + 1) / 2, (W + 1) / 2, C, vec, 1024, 32, segh, wblocks, hstrips);
    dim3 g((unsigned)((int64_t)N * wblocks * hstrips)), b(DB);
    if (vec == 4) { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_dw_dgrad_s2<T, 4>), g, b, 0, st, (const T*)dy, w, (T*)dx, N, H, W, C, Ho, Wo, segh, wblocks, hstrips, (const T*)res)); }
    else { TCCT_DISPATCH(dtype, hipLaunchKernelGGL((k_dw_dgrad_s2<T, 1>), g, b, 0, st, (const T*)dy, w, (T*)dx, N, H, W, C, Ho, Wo, segh, wblocks, hstrips, (const T*)res)); }
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_dwconv3x3_dgrad(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride,
                                    int add_input, int dtype, tcct_stream_t stream) {
    return dw_dgrad_impl(dy, w, nullptr, dx, N, H, W, C, stride, add_input, dtype, stream);
}
/* input gradient with a second gradient folded in: dx = dgrad(dy) (+ dy if add_input) + res, res [N,H,W,C] */
extern "C" int tcct_dwconv3x3_dgrad_add(const void* dy, const float* w, const void* res, void* dx, int N, int H, int W, int C,
                                        int stride, int add_input, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr, "dwconv3x3_dgrad_add: res is NULL");
    return dw_dgrad_impl(dy, w, res, dx, N, H, W, C, stride, add_input, dtype, stream);
}

/* stride-1 input gradient + the backward REDUCTION of the train-mode BatchNorm + Hardswish whose output the convolution consumed (round 4): y_prev = that
 * BatchNorm's input, ab_prev = {a[C], b[C]}; raw fp64 [2C] (zero on entry) += {sum dz', sum dz' y_prev}, dz' = dz hswish'(a y_prev + b), dz = dx as stored
 * (the form tcct_bn_sums_from_raw / tcct_pw_bwd_bn_sums take).  C % 4 == 0, C <= 256. */
extern "C" int tcct_dwconv3x3_dgrad_bnred(const void* dy, const float* w, const void* y_prev, const float* ab_prev, void* dx, double* raw, int N, int H, int W,
                                          int C, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1 && C >= 4 && C % 4 == 0 && C <= DB && y_prev && ab_prev && raw, "dwconv3x3_dgrad_bnred: needs C %% 4 == 0, C <= 256, y_prev, ab_prev, raw");
    TCCT_CHECK((int64_t)H * W * C * 4 < (1ll << 31), "dwconv3x3_dgrad_bnred: one image of %d x %d x %d elements exceeds the 32-bit byte offsets of the kernels", H, W, C);
    TCCT_DISPATCH(dtype, (dw_fwd_launch<T, 4, true, 2>(dy, w, nullptr, dx, N, H, W, C, 1, H, W, 0, (hipStream_t)stream, y_prev, raw, ab_prev)));
    TCCT_LAUNCH_OK();
}

// dw[c][ky][kx] = sum_p x[p@tap][c] * dy[p][c];  dbias[c] = sum_p dy[p][c].  Same marching window as the forward kernel with
// 10*VEC register sums per thread; LDS combine over the columns of the block, then one fp32 atomic per (channel, tap) per block.
// Few, tall strips (about 1024 blocks) keep the number of same-address atomics low.
// CPT (VEC == 4, stride 1): the thread owns CPT consecutive output columns and loads CPT + 2 input columns per row, so an input element is
// requested by (CPT + 2) / CPT instead of three threads (the requests are L1 hits, but the address / tag pipeline pays for each)
template <typename T, int VEC, int STRIDE, int CPT = 1, bool XAFF = false>
__global__ void __launch_bounds__(DB) k_dw_wgrad(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                 float* __restrict__ dbias, int N, int H, int W, int C, int Ho, int Wo, int segh,
                                                 int wblocks, int hstrips, const float* __restrict__ xab = nullptr) {
    // xab != NULL (4-channel vectors): x is y_prev with z = hswish(a y_prev + b) pending (see dw_xf): the weight gradient is taken against z
    static_assert(CPT == 1 || (STRIDE == 1 && VEC == 4), "several columns per thread: stride 1, 4-channel vectors");
    constexpr int NC = CPT + 2;
    extern __shared__ float sm[];   // [DB][10*VEC]
    const int CV = C / VEC, PW = DB / CV;
    const int t = threadIdx.x;
    float acc[10][VEC];
#pragma unroll
    for (int a = 0; a < 10; ++a)
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[a][k] = 0.f;
    typedef DwChunk<STRIDE> K;
    DwPos p;
    if (dw_pos(C, VEC, Ho, Wo, segh, wblocks, hstrips, p, CPT)) {
        const T* img = x + (int64_t)p.n * H * W * C + p.c;
        const T* gimg = dy + (int64_t)p.n * Ho * Wo * C + p.c;
        const int wi0 = p.wo * STRIDE - 1;
      if (VEC == 4) {       // branch-free loads through buffer descriptors (see k_dw_fwd)
        Raw<T, VEC> R[K::ROWS][NC], NX[K::NEW][NC], G[K::RB][CPT], GX[K::RB][CPT];
        const uint32_t ES = sizeof(T);
        const int nu = __builtin_amdgcn_readfirstlane(p.n);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)nu * H * W * C), 0, (uint32_t)H * W * C * ES, 0x00020000);
        const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)nu * Ho * Wo * C), 0, (uint32_t)Ho * Wo * C * ES, 0x00020000);
        uint32_t cin[NC], cg[CPT];
#pragma unroll
        for (int kx = 0; kx < NC; ++kx) cin[kx] = (unsigned)(wi0 + kx) < (unsigned)W ? (uint32_t)((wi0 + kx) * C + p.c) * ES : DW_OOB;
#pragma unroll
        for (int cc = 0; cc < CPT; ++cc) cg[cc] = p.wo + cc < Wo ? (uint32_t)((p.wo + cc) * C + p.c) * ES : DW_OOB;
        auto load_row = [&](Raw<T, VEC> (&r)[NC], int hi, bool live) {
            const bool rok = live && (unsigned)hi < (unsigned)H;
            const uint32_t ro = (uint32_t)hi * (uint32_t)(W * C) * ES;
#pragma unroll
            for (int kx = 0; kx < NC; ++kx) r[kx].loadb(rx, (rok && cin[kx] != DW_OOB) ? ro + cin[kx] : DW_OOB);
        };
        auto load_g = [&](Raw<T, VEC> (&g)[K::RB][CPT], int ho) {
#pragma unroll
            for (int j = 0; j < K::RB; ++j)
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc)
                    g[j][cc].loadb(rg, (ho + j < p.ho1 && cg[cc] != DW_OOB) ? (uint32_t)(ho + j) * (uint32_t)(Wo * C) * ES + cg[cc] : DW_OOB);
        };
#pragma unroll
        for (int i = 0; i < K::ROWS; ++i) load_row(R[i], p.ho0 * STRIDE - 1 + i, true);
        float xa[VEC], xb[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) { xa[k] = XAFF ? xab[p.c + k] : 1.f; xb[k] = XAFF ? xab[C + p.c + k] : 0.f; }
        auto xf_row = [&](Raw<T, VEC> (&r)[NC], int hi, bool live) {
            const bool rok = live && (unsigned)hi < (unsigned)H;
#pragma unroll
            for (int kx = 0; kx < NC; ++kx) r[kx] = dw_xf(r[kx], xa, xb, rok && cin[kx] != DW_OOB);
        };
        if (XAFF) {
#pragma unroll
            for (int i = 0; i < K::CARRY; ++i) xf_row(R[i], p.ho0 * STRIDE - 1 + i, true);
        }
        load_g(G, p.ho0);
        for (int ho = p.ho0; ho < p.ho1; ho += K::RB) {
            const bool more = ho + K::RB < p.ho1;
#pragma unroll
            for (int i = 0; i < K::NEW; ++i) load_row(NX[i], (ho + K::RB) * STRIDE - 1 + K::CARRY + i, more);
            load_g(GX, ho + K::RB);
            if (XAFF) {  // this chunk's new rows, with the next chunk's loads in flight (see k_dw_fwd)
#pragma unroll
                for (int i = 0; i < K::NEW; ++i) xf_row(R[K::CARRY + i], ho * STRIDE - 1 + K::CARRY + i, true);
            }
#pragma unroll
            for (int j = 0; j < K::RB; ++j) {
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const float g = G[j][cc].get(k);
                        acc[9][k] += g;
#pragma unroll
                        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                            for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx][k] += R[j * STRIDE + ky][cc + kx].get(k) * g;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < K::CARRY; ++i)
#pragma unroll
                for (int kx = 0; kx < NC; ++kx) R[i][kx] = R[K::NEW + i][kx];
#pragma unroll
            for (int i = 0; i < K::NEW; ++i)
#pragma unroll
                for (int kx = 0; kx < NC; ++kx) R[K::CARRY + i][kx] = NX[i][kx];
#pragma unroll
            for (int j = 0; j < K::RB; ++j)
#pragma unroll
                for (int cc = 0; cc < CPT; ++cc) G[j][cc] = GX[j][cc];
        }
      } else {
        Raw<T, VEC> R[K::ROWS][3], NX[K::NEW][3], G[K::RB], GX[K::RB];
#pragma unroll
        for (int i = 0; i < K::ROWS; ++i) dw_load_row<T, VEC>(R[i], img, p.ho0 * STRIDE - 1 + i, H, W, C, wi0);
#pragma unroll
        for (int j = 0; j < K::RB; ++j) {
            if (p.ho0 + j < p.ho1) G[j].load(gimg + ((int64_t)(p.ho0 + j) * Wo + p.wo) * C);
            else G[j].zero();
        }
        for (int ho = p.ho0; ho < p.ho1; ho += K::RB) {
            if (ho + K::RB < p.ho1) {
#pragma unroll
                for (int i = 0; i < K::NEW; ++i) dw_load_row<T, VEC>(NX[i], img, (ho + K::RB) * STRIDE - 1 + K::CARRY + i, H, W, C, wi0);
#pragma unroll
                for (int j = 0; j < K::RB; ++j) {
                    if (ho + K::RB + j < p.ho1) GX[j].load(gimg + ((int64_t)(ho + K::RB + j) * Wo + p.wo) * C);
                    else GX[j].zero();
                }
            }
            // rows past the end of the strip carry a zero gradient, so they need no guard here
#pragma unroll
            for (int j = 0; j < K::RB; ++j) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    const float g = G[j].get(k);
                    acc[9][k] += g;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx][k] += R[j * STRIDE + ky][kx].get(k) * g;
                }
            }
#pragma unroll
            for (int i = 0; i < K::CARRY; ++i)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) R[i][kx] = R[K::NEW + i][kx];
#pragma unroll
            for (int i = 0; i < K::NEW; ++i)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) R[K::CARRY + i][kx] = NX[i][kx];
#pragma unroll
            for (int j = 0; j < K::RB; ++j) G[j] = GX[j];
        }
      }
    }
#pragma unroll
    for (int a = 0; a < 10; ++a)
#pragma unroll
        for (int k = 0; k < VEC; ++k) sm[(t * 10 + a) * VEC + k] = acc[a][k];
    __syncthreads();
    for (int o = t; o < C * 10; o += DB) {
        int ch = o / 10, a = o % 10;
        int cvv = ch / VEC, k = ch % VEC;
        float s = 0.f;
        for (int pp = 0; pp < PW; ++pp) s += sm[((pp * CV + cvv) * 10 + a) * VEC + k];
        if (a < 9) atomicAdd(&dw[ch * 9 + a], s);
        else if (dbias) atomicAdd(&dbias[ch], s);
    }
}

static int dw_wgrad_impl(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int stride, int dtype, tcct_stream_t stream,
                         const float* xab);
extern "C" int tcct_dwconv3x3_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C,
                                    int stride, int dtype, tcct_stream_t stream) {
    return dw_wgrad_impl(x, dy, dw, dbias, N, H, W, C, stride, dtype, stream, nullptr);
}
/* weight gradient against z = hswish(a y_prev + b) rebuilt on load from x = y_prev (see tcct_dwconv3x3_fwd_xaff); C % 4 == 0 */
extern "C" int tcct_dwconv3x3_wgrad_xaff(const void* x, const float* xab, const void* dy, float* dw, float* dbias, int N, int H, int W, int C,
                                         int stride, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(xab != nullptr && C % 4 == 0, "dwconv3x3_wgrad_xaff: needs xab and C %% 4 == 0");
    return dw_wgrad_impl(x, dy, dw, dbias, N, H, W, C, stride, dtype, stream, xab);
}
static int dw_wgrad_impl(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int stride, int dtype, tcct_stream_t stream,
                         const float* xab) {
    TCCT_CHECK(stride == 1 || stride == 2, "dwconv3x3_wgrad: stride %d", stride);
    TCCT_CHECK(N >= 1 && H >= 1 && W >= 1 && C >= 1, "dwconv3x3_wgrad: empty tensor");
    int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    int vec = (C % 4 == 0) ? 4 : 1;
    TCCT_CHECK(C / vec <= DB, "dwconv3x3_wgrad: C=%d too large", C);
    TCCT_CHECK((int64_t)H * W * C * 4 < (1ll << 31), "dwconv3x3_wgrad: one image of %d x %d x %d elements exceeds the 32-bit byte offsets of the kernels", H, W, C);
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * C * 9, st) != hipSuccess) { tcct_set_error("dwconv3x3_wgrad: memset failed"); return -2; }
    if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * C, st) != hipSuccess) { tcct_set_error("dwconv3x3_wgrad: memset failed"); return -2; }
    int segh, wblocks, hstrips;
    size_t lds = sizeof(float) * DB * 10 * vec;
    static int cpt_on = -1;     // compile-time A/B switch (0: one column per thread for every shape)
    if (cpt_on < 0) cpt_on = 1;
    if (cpt_on && vec == 4 && stride == 1 && dtype == TCCT_BF16 && Wo >= 128 && C >= 32) {
        dw_geometry(N, Ho, Wo, C, vec, 256, 128, segh, wblocks, hstrips, 2);     // (256 blocks: 0.080 -> 0.060 ms at level 2, 0.119 -> 0.103 stride 2 at level 1; 512 the same at level 1 stride 1)
        if (xab) hipLaunchKernelGGL((k_dw_wgrad<bf16, 4, 1, 2, true>), dim3((unsigned)((int64_t)N * wblocks * hstrips)), dim3(DB), lds, st, (const bf16*)x, (const bf16*)dy, dw, dbias,
                                    N, H, W, C, Ho, Wo, segh, wblocks, hstrips, xab);
        else hipLaunchKernelGGL((k_dw_wgrad<bf16, 4, 1, 2>), dim3((unsigned)((int64_t)N * wblocks * hstrips)), dim3(DB), lds, st, (const bf16*)x, (const bf16*)dy, dw, dbias,
                                N, H, W, C, Ho, Wo, segh, wblocks, hstrips, xab);
        TCCT_LAUNCH_OK();
    }
    dw_geometry(N, Ho, Wo, C, vec, 256, 128, segh, wblocks, hstrips);
    dim3 grid((unsigned)((int64_t)N * wblocks * hstrips));
#define DWG(V, S) hipLaunchKernelGGL((k_dw_wgrad<T, V, S>), grid, dim3(DB), lds, st, (const T*)x, (const T*)dy, dw, dbias, N, H, W, C, Ho, Wo, segh, wblocks, hstrips, xab)
#define DWGX(S) hipLaunchKernelGGL((k_dw_wgrad<T, 4, S, 1, true>), grid, dim3(DB), lds, st, (const T*)x, (const T*)dy, dw, dbias, N, H, W, C, Ho, Wo, segh, wblocks, hstrips, xab)
    if (vec == 4 && xab) { TCCT_DISPATCH(dtype, if (stride == 1) DWGX(1); else DWGX(2)); }
    else if (vec == 4) { TCCT_DISPATCH(dtype, if (stride == 1) DWG(4, 1); else DWG(4, 2)); }
    else { TCCT_DISPATCH(dtype, if (stride == 1) DWG(1, 1); else DWG(1, 2)); }
#undef DWG
#undef DWGX
    TCCT_LAUNCH_OK();
}
