// Two dense 32 -> 32 3x3 convolutions with NOTHING between them as ONE launch (round 5): conv3x3 -> conv3x3 of CrossCNNBlock.block12 (reference
// nets/tcct.py:808-810) and the input-gradient chain of the same pair in the backward pass.  The intermediate is written to HBM once (the backward pass
// needs it: it is the x operand of the second convolution's weight gradient / the dy operand of the first one's) but is NOT read back: the second convolution
// takes it from LDS.
//
// Row streams with ROLES.  k_conv32_fwd33_stream keeps 18 weight fragments (72 VGPRs) and three rolling accumulators per wave; two convolutions in one wave
// do not fit the 256-register budget of two waves per SIMD.  So a block of four waves serves TWO adjacent strips, each with a PRODUCER wave (convolution 1,
// fed by its own LDS-DMA ring exactly like the single kernel) and a CONSUMER wave (convolution 2) on another SIMD, which finds its halo rows in a small LDS
// "mid ring" the producer's epilogue fills: the packed bf16 row that is on its way to HBM anyway.  One s_barrier per row hands a row over (RAW) and keeps a
// slot from being overwritten before it was read (WAR, ring of 3); both roles issue 18 MFMAs per row, so the SIMDs stay balanced.
//
// Geometry: a strip is 30 output pixels wide.  The consumer's 32 MFMA columns are output pixels w0 .. w0 + 31 (the last two are not stored); it reads mid
// pixels w0 - 1 .. w0 + 32 of which the producer computes w0 - 1 .. w0 + 30 (its 32 MFMA columns) from input pixels w0 - 2 .. w0 + 31 (34 pixels: the DMA row
// of the single kernel).  Mid pixels outside the image are ZERO (the second convolution pads its input with zeros; convolution 1 evaluated outside the
// image is not zero), rows likewise.  A run of L output rows needs L + 2 mid rows from L + 4 input rows.
//
// Arithmetic: every pixel of either convolution sums bias, then its taps in (dy, dx, half) order on v_mfma_f32_32x32x16_bf16, and the intermediate is
// rounded to bf16 exactly where the two-launch path stores it => results BIT-IDENTICAL to tcct_conv32_fwd twice (tests/test_kernels_gpu.py).
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 ch_bf16x8;
typedef __attribute__((ext_vector_type(16))) float ch_f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int ch_u32x4;

#define CH_T 256
#define CH_R 9                  // producer ring rows (a multiple of 3: ring slot and accumulator index are compile-time in the 9-fold unrolled body)
#define CH_P 7                  // input rows in flight per producer
#define CH_ROWB 2176            // 34 pixels x 64 B (input ring rows and mid ring rows alike)
#define CH_MR 3                 // mid ring rows per strip
#define CH_SW 30                // output pixels per strip
#define CH_OOB 0x80000000u

__device__ __forceinline__ void ch_lds_dma16(const ch_u32x4& rsrc, uint32_t voff, uint32_t lds_addr) {
    // 64 lanes x 16 B from per-lane buffer offsets to LDS bytes [lds_addr + 16 lane ..) (see lds_dma16 in conv_mfma.hip: inline asm so that hipcc does not drain
    // the DMA in front of the next ds_read; M0 saved and restored)
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ ch_u32x4 ch_rsrc_words(const void* base, uint32_t bytes) {
    const uint64_t b = (uint64_t)base;
    ch_u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}
// LDS writes of this wave have landed, then the block barrier: the hand-over of a mid row.  (__syncthreads() would also drain vmcnt -- the DMA pipeline.)
__device__ __forceinline__ void ch_row_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// MODE 0: plain; 2: statistics of LeakyReLU(y) of the SECOND convolution's output as stored -> stats[0..31], stats[32..63] (fp64 atomics; the forward pass of
// block12, whose BatchNorm follows); 3: y = conv2(conv1(x)) + res (the input gradient of a convolution whose input has a second consumer)
template <int MODE>
__global__ void __launch_bounds__(CH_T, 2)
k_conv32_chain33(const bf16* __restrict__ x, const bf16* __restrict__ wp1, const float* __restrict__ bias1, bf16* __restrict__ mid,
                 const bf16* __restrict__ wp2, const float* __restrict__ bias2, bf16* __restrict__ y, const bf16* __restrict__ res,
                 int N, int H, int W, int strips, int run, int rpi, double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const bool producer = wave < 2;
    const int sidx = wave & 1;                                  // strip of the pair
    // LDS: [2 producers][CH_R rows] input rings | [2 strips][CH_MR rows] mid rings | [2 consumers] 2 KB transpose scratch | bias1[32] bias2[32] | stats partials [2][64]
    unsigned char* ring = smem + sidx * (CH_R * CH_ROWB);
    unsigned char* midr = smem + 2 * CH_R * CH_ROWB + sidx * (CH_MR * CH_ROWB);
    unsigned char* scr = smem + 2 * CH_R * CH_ROWB + 2 * CH_MR * CH_ROWB + sidx * 2048;
    float* sB = reinterpret_cast<float*>(smem + 2 * CH_R * CH_ROWB + 2 * CH_MR * CH_ROWB + 2 * 2048);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
    const bf16* wp = producer ? wp1 : wp2;
    ch_bf16x8 Wf[9][2];         // A operands: row co = r, input channels 8 hh + 16 half ..+7 of tap t
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) Wf[t][kc] = *reinterpret_cast<const ch_bf16x8*>(wp + (t * 32 + r) * 32 + (hh + 2 * kc) * 8);
    if (tid < 32) sB[tid] = bias1 ? bias1[tid] : 0.f;
    else if (tid < 64) sB[tid] = bias2 ? bias2[tid - 32] : 0.f;
    // mid pixels 32, 33 are never written by the producer and only feed the two output columns nobody stores: give them a defined value once
    if (tid < 2 * CH_MR * 8) {
        const int st = tid / (CH_MR * 8), rem = tid - st * (CH_MR * 8), row = rem >> 3, q = rem & 7;
        *reinterpret_cast<uint4*>(smem + 2 * CH_R * CH_ROWB + st * (CH_MR * CH_ROWB) + row * CH_ROWB + 2048 + q * 16) = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    const float* bq = sB + (producer ? 0 : 32);
    const unsigned char* src = producer ? ring : midr;          // where this wave's B fragments come from
    const unsigned char* xB[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) { const int P = r + d; xB[d][kc] = src + P * 64 + (((hh + 2 * kc) ^ ((P >> 2) & 3)) << 4); }
    constexpr bool ST = MODE == 2;
    float ss[ST ? 8 : 1], sq[ST ? 8 : 1];
#pragma unroll
    for (int k = 0; k < (ST ? 8 : 1); ++k) ss[k] = sq[k] = 0.f;
    const int pairs = (strips + 1) >> 1;
    // consecutive blocks take the strip pairs of ONE run of image rows (whole image rows in flight together), then the next run, then the next image
    const int q_ = blockIdx.x / pairs, pair = blockIdx.x - q_ * pairs;
    const int n = q_ / rpi, r0 = (q_ - n * rpi) * run;
    const int L = __builtin_amdgcn_readfirstlane(H - r0 < run ? H - r0 : run);      // output rows r0 .. r0 + L - 1
    const int s = 2 * pair + sidx;
    const bool live = s < strips;
    const int w0 = s * CH_SW;
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    const uint32_t rowb = (uint32_t)W * 64u;
    const int groups = (L + 4 + CH_R - 1) / CH_R;               // iterations t = 0 .. L + 3 (see below), whole groups of CH_R
    const int p16 = lane >> 2, cch = lane & 3;
    if (producer) {
        // ---------------------------------------------------------------------------------------------- convolution 1
        // iteration t: input halo row t (image row r0 - 2 + t) arrives; mid row m = t - 2 (image row r0 - 1 + m, m = 0 .. L + 1) is complete after its MFMAs
        const ch_u32x4 rx = ch_rsrc_words(x + (int64_t)n * H * W * 32, img_bytes);
        const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(mid + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        const int pq = lane >> 2, cs = (lane & 3) ^ ((lane >> 4) & 3);          // LDS position lane & 3 of ring pixel 16 piece + pq holds chunk cs
        const int c0 = w0 - 2 + pq, c1 = w0 + 14 + pq, c2 = w0 + 30 + pq;
        const uint32_t o0 = (live && c0 >= 0 && c0 < W) ? (uint32_t)(c0 * 64 + cs * 16) : CH_OOB;
        const uint32_t o1 = (live && c1 < W) ? (uint32_t)(c1 * 64 + cs * 16) : CH_OOB;
        const uint32_t o2 = (live && c2 < W) ? (uint32_t)(c2 * 64 + cs * 16) : CH_OOB;
        // the 30 mid pixels this strip owns (mid pixel 1 + i <-> image column w0 + i), 16-byte pieces: piece 64 u + lane -> pixel 16 u + (lane >> 2)
        uint32_t so[2], sl[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = 16 * u + p16, wo = w0 + i;
            so[u] = (live && i < CH_SW && wo < W) ? (uint32_t)(wo * 64 + cch * 16) : CH_OOB;
            const int P = 1 + i;
            sl[u] = (uint32_t)(P * 64 + ((cch ^ ((P >> 2) & 3)) << 4));
        }
        const int mcol = w0 - 1 + r;                            // image column of this lane's mid pixel
        const bool colin = live && mcol >= 0 && mcol < W;
        const uint32_t mw = (uint32_t)(r * 64 + hh * 8);        // + ((q ^ swz) << 4): this lane's four 8-byte pieces of mid pixel r
        const int mswz = (r >> 2) & 3;
        const int64_t row_first = (int64_t)r0 - 2;              // image row of halo row 0 (may be negative: the range check of the descriptor is on the OFFSET,
        auto issue = [&](int a, int slot) {                     // so rows outside the image are masked here)
            const int64_t ri = row_first + a;
            const bool xin = a <= L + 3 && ri >= 0 && ri < H;
            const uint32_t ro = (uint32_t)(xin ? ri : 0) * rowb;
            const uint32_t base = ring_lds + (uint32_t)(slot * CH_ROWB);
            ch_lds_dma16(rx, xin ? o0 + ro : CH_OOB, base);
            ch_lds_dma16(rx, xin ? o1 + ro : CH_OOB, base + 1024u);
            if (lane < 8) ch_lds_dma16(rx, xin ? o2 + ro : CH_OOB, base + 2048u);
        };
#pragma unroll
        for (int a = 0; a < CH_P; ++a) issue(a, a);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (CH_P - 1)) : "memory");
        ch_bf16x8 X[3][2], Xn[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) X[d][kc] = *reinterpret_cast<const ch_bf16x8*>(xB[d][kc]);
        ch_f32x16 acc[3];
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int j = 0; j < CH_R; ++j) {
                const int t = g * CH_R + j;
                issue(t + CH_P, (j + CH_P) % CH_R);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (CH_P - 1)) : "memory");          // halo row t + 1 has landed (loads retire in order among loads)
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) Xn[d][kc] = *reinterpret_cast<const ch_bf16x8*>(xB[d][kc] + ((j + 1) % CH_R) * CH_ROWB);
                {           // mid row t starts at the bias of this lane's 16 channels
                    float4 b4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) b4[q] = *reinterpret_cast<const float4*>(bq + 8 * q + 4 * hh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[j % 3][4 * q] = b4[q].x; acc[j % 3][4 * q + 1] = b4[q].y; acc[j % 3][4 * q + 2] = b4[q].z; acc[j % 3][4 * q + 3] = b4[q].w; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)          // halo row t is tap row dy of mid row t - dy
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int kc = 0; kc < 2; ++kc)
                            acc[(j + 3 - dy) % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[dy * 3 + d][kc], X[d][kc], acc[(j + 3 - dy) % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // mid row m = t - 2 is complete: pack; zero outside the image (the zero padding of convolution 2); into the mid ring for the consumer
                const int m = t - 2;
                const int64_t mrow = (int64_t)r0 - 1 + m;
                const bool rowin = m >= 0 && m <= L + 1 && mrow >= 0 && mrow < H;
                const ch_f32x16& A = acc[(j + 1) % 3];
                unsigned char* slot = midr + ((j + 1) % CH_MR) * CH_ROWB;           // (t - 2) mod 3 == (j + 1) mod 3 because CH_R is a multiple of 3
                const bool keep = rowin && colin;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint2 o;
                    o.x = keep ? pack_bf16x2(A[4 * q], A[4 * q + 1]) : 0u;
                    o.y = keep ? pack_bf16x2(A[4 * q + 2], A[4 * q + 3]) : 0u;
                    *reinterpret_cast<uint2*>(slot + mw + ((q ^ mswz) << 4)) = o;
                }
                ch_row_barrier();                       // B_t: mid row t - 2 is in LDS for the consumer
                // ... and goes to HBM from the same slot (rows r0 .. r0 + L - 1 only: the two halo mid rows belong to the neighbouring runs)
                ch_u32x4 pend[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) pend[u] = *reinterpret_cast<const ch_u32x4*>(slot + sl[u]);
                const bool own = m >= 1 && m <= L && mid != nullptr;      // mid == NULL (inference): the intermediate lives in LDS only
                const uint32_t oro = (uint32_t)(own ? mrow : 0) * rowb;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    __builtin_amdgcn_raw_buffer_store_b128(pend[u], ws, (own && so[u] != CH_OOB) ? so[u] + oro : CH_OOB, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) X[d][kc] = Xn[d][kc];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // ---------------------------------------------------------------------------------------------- convolution 2
        // iteration t: after B_t mid row m = t - 2 (halo row m of this convolution) is in the ring; output row o = m - 2 = t - 4 (image row r0 + o) is complete after it
        const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)((MODE == 3 ? res : y) + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        uint32_t so[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = 16 * u + p16, wo = w0 + i;
            so[u] = (live && i < CH_SW && wo < W) ? (uint32_t)(wo * 64 + cch * 16) : CH_OOB;
        }
        const int ocol = w0 + r;
        const uint32_t radd = (live && r < CH_SW && ocol < W) ? (uint32_t)(ocol * 64 + hh * 8) : CH_OOB;         // + 16 q: this lane's channels 8 q + 4 hh ..+3 of `res`
        ch_f32x16 acc[3];
        ch_bf16x8 X[3][2];
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int j = 0; j < CH_R; ++j) {
                const int t = g * CH_R + j;
                const int o = t - 4;
                const bool ovalid = o >= 0 && o < L;
                uint2 rv[4];
                if (MODE == 3) {            // the other consumer's gradient of this output row: requested before the barrier, added in the epilogue
                    const uint32_t oro = (uint32_t)(ovalid ? r0 + o : 0) * rowb;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const auto v = __builtin_amdgcn_raw_buffer_load_b64(rr, (ovalid && radd != CH_OOB) ? radd + oro + 16u * q : CH_OOB, 0, 0);
                        rv[q].x = v[0]; rv[q].y = v[1];
                    }
                }
                ch_row_barrier();                       // B_t
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) X[d][kc] = *reinterpret_cast<const ch_bf16x8*>(xB[d][kc] + ((j + 1) % CH_MR) * CH_ROWB);
                {           // output row t - 2 starts at the bias (its first tap row is this halo row)
                    float4 b4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) b4[q] = *reinterpret_cast<const float4*>(bq + 8 * q + 4 * hh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[(j + 1) % 3][4 * q] = b4[q].x; acc[(j + 1) % 3][4 * q + 1] = b4[q].y; acc[(j + 1) % 3][4 * q + 2] = b4[q].z; acc[(j + 1) % 3][4 * q + 3] = b4[q].w; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)          // mid row m = t - 2 is tap row dy of output row m - dy; accumulator of output row o' lives at (o' + 3) % 3 ... (j + 1 - dy) % 3
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int kc = 0; kc < 2; ++kc)
                            acc[(j + 4 - dy) % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[dy * 3 + d][kc], X[d][kc], acc[(j + 4 - dy) % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // output row o = t - 4 is complete (accumulator (j + 2) % 3): (+ res), pack, transpose through this wave's scratch, store 30 pixels
                ch_f32x16& A = acc[(j + 2) % 3];
                uint2 ov[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v4[4] = {A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]};
                    if (MODE == 3) {
                        v4[0] += __uint_as_float(rv[q].x << 16); v4[1] += __uint_as_float(rv[q].x & 0xffff0000u);
                        v4[2] += __uint_as_float(rv[q].y << 16); v4[3] += __uint_as_float(rv[q].y & 0xffff0000u);
                    }
                    ov[q].x = pack_bf16x2(v4[0], v4[1]); ov[q].y = pack_bf16x2(v4[2], v4[3]);
                }
                const int f = (r >> 1) & 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(scr + r * 64 + ((q ^ f) << 4) + hh * 8) = ov[q];
                wave_lds_fence();
                ch_u32x4 pend[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) pend[u] = *reinterpret_cast<const ch_u32x4*>(scr + (16 * u + p16) * 64 + ((cch ^ ((p16 >> 1) & 3)) << 4));
                wave_lds_fence();
                const uint32_t oro = (uint32_t)(ovalid ? r0 + o : 0) * rowb;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool inb = ovalid && so[u] != CH_OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(pend[u], ws, inb ? so[u] + oro : CH_OOB, 0, 0);
                    if (ST) {
                        if (inb) {
                            const uint32_t wv[4] = {pend[u][0], pend[u][1], pend[u][2], pend[u][3]};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float u0 = __uint_as_float(wv[k] << 16), u1 = __uint_as_float(wv[k] & 0xffff0000u);
                                u0 = fmaxf(u0, 0.01f * u0); u1 = fmaxf(u1, 0.01f * u1);
                                ss[2 * k] += u0; sq[2 * k] += u0 * u0; ss[2 * k + 1] += u1; sq[2 * k + 1] += u1 * u1;
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if (ST) {
        // a consumer lane owns channels 8 (lane & 3) ..+7 of the pixels it stored: butterfly over lane bits 2..5, per-wave LDS slots, fp64 atomics
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* red = sB + 64;
        if (!producer) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float a = ss[k], b = sq[k];
#pragma unroll
                for (int o = 32; o > 2; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
                if (lane < 4) {
                    red[sidx * 64 + 8 * lane + k] = a;
                    red[sidx * 64 + 32 + 8 * lane + k] = b;
                }
            }
        }
        __syncthreads();
        if (tid < 64) atomicAdd(&stats[tid], (double)red[tid] + (double)red[64 + tid]);
    }
}

/* y = conv3x3(conv3x3(x; w1, b1); w2, b2) on bf16 NHWC [N,H,W,32], both 'same'; mid = the first convolution's output (written, never read here; NULL: not
 * written at all -- inference, where nothing needs it: 2 tensors of traffic instead of 4).
 * wp1 / wp2: packed bf16 [9][32][32] (tcct_conv32_pack_weights; the flipped / transposed packs for an input-gradient chain).  stats (fp64 [64], zero on entry,
 * or NULL): += {sum, sum of squares} per channel of LeakyReLU(y) as stored -- the train-mode BatchNorm behind block12 (reference nets/tcct.py:808-811).
 * res (or NULL): y += res before the store (a second consumer's gradient, as tcct_conv32_fwd_add).  stats and res exclude each other.
 * Replaces tcct_conv32_fwd(x -> mid) + tcct_conv32_fwd[_bnstats | _add](mid -> y): bit-identical results, `mid` is not read back from HBM. */
extern "C" int tcct_conv32_chain33(const void* x, const void* wp1, const float* bias1, void* mid, const void* wp2, const float* bias2, void* y,
                                   const void* res, int N, int H, int W, double* stats, tcct_stream_t stream) {
    TCCT_CHECK(x && wp1 && wp2 && y && x != mid && mid != y && x != y, "conv32_chain33: x, mid, y must be three tensors (mid may be NULL)");
    TCCT_CHECK(!(stats && res), "conv32_chain33: statistics and a residual exclude each other");
    TCCT_CHECK(N > 0 && H > 0 && W > 0 && (int64_t)H * W * 64 < (1LL << 31), "conv32_chain33: one image of %d x %d exceeds the 2 GiB buffer-descriptor range", H, W);
    const int strips = (W + CH_SW - 1) / CH_SW, pairs = (strips + 1) / 2;
    // runs per image: as many as fill the 512 block slots; a run pays 4 extra input rows + the pipeline fill, so >= 24 rows on the large maps, >= 8 on the
    // small ones (where the block count matters more than the fill)
    const int minrun = H >= 96 ? 24 : 8;
    int rpi = 512 / (N * pairs);
    if (rpi > H / minrun) rpi = H / minrun;
    if (rpi < 1) rpi = 1;
    const int run = (H + rpi - 1) / rpi;
    rpi = (H + run - 1) / run;
    const int64_t blocks = (int64_t)N * rpi * pairs;
    TCCT_CHECK(blocks < (1LL << 31), "conv32_chain33: too many blocks");
    const size_t lds = (size_t)2 * CH_R * CH_ROWB + (size_t)2 * CH_MR * CH_ROWB + 2 * 2048 + 64 * 4 + 128 * 4;
    hipStream_t st = (hipStream_t)stream;
#define CH_LAUNCH(M)                                                                                                                                          \
    do {                                                                                                                                                      \
        static bool attr = false;                                                                                                                             \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_chain33<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; }       \
        hipLaunchKernelGGL((k_conv32_chain33<M>), dim3((unsigned)blocks), dim3(CH_T), lds, st, (const bf16*)x, (const bf16*)wp1, bias1, (bf16*)mid,            \
                           (const bf16*)wp2, bias2, (bf16*)y, (const bf16*)res, N, H, W, strips, run, rpi, stats);                                             \
    } while (0)
    if (stats) CH_LAUNCH(2);
    else if (res) CH_LAUNCH(3);
    else CH_LAUNCH(0);
#undef CH_LAUNCH
    tcct_census_hit(TCCT_CENSUS_CHAIN33);
    TCCT_LAUNCH_OK();
}
