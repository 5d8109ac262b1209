// MFMA implicit-GEMM convolution for the hot family of the path: dense 32 -> 32 channels, stride 1, bf16 NHWC,
// any KH x KW (3x3, 1xk "cross" row convs, kx1 column convs; reference nets/tcct.py:808-822).
//
// GEMM view per tap: D[co][pixel] += Wt[tap][co][ci] * X[pixel@tap][ci]   (M = 32 co, N = 32 pixels, K = 32 ci)
//   v_mfma_f32_32x32x16_bf16, A = weights (rows = co), B = activations (cols = pixels): each lane then owns ONE pixel
//   and 16 of its output channels, so the epilogue packs 4 bf16 -> 8-byte stores and per-pixel fusions are lane-local.
// LDS image: one 64-byte row per pixel (32 ch bf16), 16-byte chunks XOR-swizzled with ((p>>2)&3) so that the
//   ds_read_b128 of 16 consecutive pixels (one MFMA lane group) hits 16 distinct 16-byte slots (conflict-free); the
//   weights use the same image with row = tap*32+co.  HORZ tiles (8 rows x 64 cols, M-tiles along W) serve 3x3 and 1xk;
//   VERT tiles (64 rows x 8 cols, column-major LDS image, M-tiles along H) serve kx1 with a 1.19x halo instead of 2.5x.
// Each block stages the packed weights once and walks tiles grid-stride; 2 blocks/CU overlap staging with MFMA.
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4;

#define MB 256
#define OOB_OFF 0x80000000u     // buffer offset beyond every descriptor range used here (< 2 GiB): loads return 0, stores are dropped

// OIHW fp32 -> [tap][co][ci] bf16 (optionally flipped + transposed: the weights of the input-gradient convolution)
// The 32x32 block (o_off.., i_off..) of an OIHW weight with `ldi` input channels is selected, so 32->64 / 64->32 convolutions
// decompose into 32x32 sub-GEMMs of the same kernels.
__global__ void k_pack_w32(const float* __restrict__ w, bf16* __restrict__ wp, int KH, int KW, int transposed, int ldi,
                           int o_off, int i_off) {
    int total = KH * KW * 32 * 32;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
        int dy = tap / KW, dx = tap % KW;
        const float vt = w[(((int64_t)(o_off + ci) * ldi + i_off + co) * KH + (KH - 1 - dy)) * KW + (KW - 1 - dx)];
        const float vn = w[(((int64_t)(o_off + co) * ldi + i_off + ci) * KH + dy) * KW + dx];
        if (transposed == 2) { wp[i] = __float2bfloat16(vn); wp[total + i] = __float2bfloat16(vt); }     // both packs, one launch
        else wp[i] = __float2bfloat16(transposed ? vt : vn);
    }
}
extern "C" int tcct_conv32_pack_weights(const float* w, void* wp, int KH, int KW, int transposed, tcct_stream_t stream) {
    int total = KH * KW * 1024;
    hipLaunchKernelGGL(k_pack_w32, dim3((total + MB - 1) / MB), dim3(MB), 0, (hipStream_t)stream, w, (bf16*)wp, KH, KW, transposed, 32, 0, 0);
    TCCT_LAUNCH_OK();
}
/* wp2 [2][KH*KW][32][32]: the forward pack followed by the input-gradient (flipped + transposed) pack -- the backward pass of the same
 * step reuses the second half instead of launching another pack */
extern "C" int tcct_conv32_pack_weights_both(const float* w, void* wp2, int KH, int KW, tcct_stream_t stream) {
    int total = KH * KW * 1024;
    hipLaunchKernelGGL(k_pack_w32, dim3((total + MB - 1) / MB), dim3(MB), 0, (hipStream_t)stream, w, (bf16*)wp2, KH, KW, 2, 32, 0, 0);
    TCCT_LAUNCH_OK();
}
// every 32 -> 32 convolution weight of the network in ONE launch: desc[i] = {source fp32 OIHW pointer, destination (forward pack followed by the
// flipped / transposed pack, 2 * KH*KW*1024 bf16), KH, KW}; blockIdx.y = convolution
struct PackDesc { const float* w; bf16* wp; int64_t KH, KW; };
__global__ void k_pack_w32_multi(const PackDesc* __restrict__ desc) {
    const PackDesc d = desc[blockIdx.y];
    const int KH = (int)d.KH, KW = (int)d.KW, total = KH * KW * 1024;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int ci = i & 31, co = (i >> 5) & 31, tap = i >> 10;
        const int dy = tap / KW, dx = tap % KW;
        d.wp[i] = __float2bfloat16(d.w[(((int64_t)co * 32 + ci) * KH + dy) * KW + dx]);
        d.wp[total + i] = __float2bfloat16(d.w[(((int64_t)ci * 32 + co) * KH + (KH - 1 - dy)) * KW + (KW - 1 - dx)]);
    }
}
/* desc: device array of n {const float* w; void* wp2; int64 KH; int64 KW} records (32 bytes each): tcct_conv32_pack_weights_both for all of them */
extern "C" int tcct_conv32_pack_weights_multi(const void* desc, int n, tcct_stream_t stream) {
    TCCT_CHECK(n >= 1 && n <= 65535 && desc != nullptr, "conv32_pack_weights_multi: bad count %d", n);
    hipLaunchKernelGGL(k_pack_w32_multi, dim3(13 * 1024 / MB, n), dim3(MB), 0, (hipStream_t)stream, (const PackDesc*)desc);
    TCCT_LAUNCH_OK();
}
extern "C" int tcct_conv32_pack_weights_sub(const float* w, void* wp, int KH, int KW, int transposed, int cin_total, int o_off,
                                            int i_off, tcct_stream_t stream) {
    int total = KH * KW * 1024;
    hipLaunchKernelGGL(k_pack_w32, dim3((total + MB - 1) / MB), dim3(MB), 0, (hipStream_t)stream, w, (bf16*)wp, KH, KW, transposed,
                       cin_total, o_off, i_off);
    TCCT_LAUNCH_OK();
}

// Staging is software-pipelined across tiles: every thread owns up to MAXL fixed 16-byte slots of the LDS image (slot
// geometry precomputed once), issues ALL its global loads for the NEXT tile into registers before the MFMA phase of the
// current tile (one memory latency per tile instead of one per slot), and writes them to LDS after the barrier.
// Tile order of the persistent convolution kernels.  Workgroup b runs on XCD b % 8 (round-robin dispatch) and every XCD has its own
// 4 MB L2, so with the plain `tile = b + k * grid` walk the eight neighbours of a tile -- whose halos are the SAME pixels -- are
// staged by eight different L2s at about the same time: the halo (1.195x for 16 x 32 tiles of a 3x3, 1.19x for 8 x 64 of a 1x13)
// crossed the fabric once per reader (PMC: 1.26x the algorithmic reads).  Here XCD x owns the contiguous eighth [x*Q, (x+1)*Q) of a
// tile SEQUENCE, its grid/8 resident blocks take consecutive sequence numbers per round, and the sequence walks the image in strips
// of 8 tiles (column strips, row-major inside, for the horizontal / square tiles; row strips, column-major inside, for the 64 x 8
// tiles of the k x 1 kernels): the 64 tiles an XCD has in flight form one 8 x 8 patch whose inner halos are L2 hits, and the next
// round continues right below (beside) it.  Everything here is block-uniform (scalar unit).
template <bool ROWSTRIP>
struct TileSeq {
    int base, loc, nl, Q, nt, tH, tW, per_img;
    __device__ __forceinline__ TileSeq(int ntiles, int tilesH, int tilesW) {
        nt = ntiles; tH = tilesH; tW = tilesW; per_img = tilesH * tilesW;
        const int g = gridDim.x, b = blockIdx.x;
        if ((g & 7) == 0) { nl = g >> 3; loc = b >> 3; Q = (ntiles + 7) >> 3; base = (b & 7) * Q; }
        else { nl = g; loc = b; Q = ntiles; base = 0; }
    }
    // k-th tile of this block, packed (n << 20) | (th << 10) | tw (shifts to decode: a linear id cost the callers three more divisions per tile,
    // and with two waves per SIMD every bookkeeping instruction is on the critical path), or -1 behind the end
    __device__ __forceinline__ int at(int k) const {
        const int q = k * nl + loc, seq = base + q;
        if (q >= Q || seq >= nt) return -1;
        const int n = seq / per_img, rem = seq - n * per_img;
        const int L = ROWSTRIP ? tW : tH;               // strip length (tiles), strips are 8 tiles wide across the other axis
        const int X = ROWSTRIP ? tH : tW;
        const int full = X >> 3, A = L << 3;
        int along, across;
        if (rem >= full * A) {
            const int wr = X - (full << 3), r2 = rem - full * A;
            along = r2 / wr; across = (full << 3) + (r2 - along * wr);
        } else {
            const int sidx = rem / A, r2 = rem - sidx * A;
            along = r2 >> 3; across = (sidx << 3) + (r2 & 7);
        }
        const int th = ROWSTRIP ? across : along, tw = ROWSTRIP ? along : across;
        return (n << 20) | (th << 10) | tw;
    }
    // the same, coordinates instead of the linear id (saves the caller three divisions per tile): false behind the end
    __device__ __forceinline__ bool at3(int k, int& n, int& th, int& tw) const {
        const int q = k * nl + loc, seq = base + q;
        if (q >= Q || seq >= nt) return false;
        n = seq / per_img;
        const int rem = seq - n * per_img;
        const int L = ROWSTRIP ? tW : tH, X = ROWSTRIP ? tH : tW;
        const int full = X >> 3, A = L << 3;
        int along, across;
        if (rem >= full * A) {
            const int wr = X - (full << 3), r2 = rem - full * A;
            along = r2 / wr; across = (full << 3) + (r2 - along * wr);
        } else {
            const int sidx = rem / A, r2 = rem - sidx * A;
            along = r2 >> 3; across = (sidx << 3) + (r2 & 7);
        }
        th = ROWSTRIP ? across : along; tw = ROWSTRIP ? along : across;
        return true;
    }
};
#define MAXL 11
#define IPS 80      // LDS bytes per image pixel: 64 B of channels + 16 B pad => conflict-free ds_read_b128 with LINEAR addressing
// An ablation of the first version showed the kernel was instruction-bound, not memory-bound: with loads, stores and MFMAs all
// removed it still took 0.20 of 0.29 ms -- XOR-swizzle address arithmetic (6 VALU ops per fragment read, ~1000 per tile and wave)
// in front of every ds_read.  Now the image rows are padded instead of swizzled and KH/KW are template constants, so a B
// fragment read is `ds_read_b128 v, base_t offset:imm` with NO per-read VALU work (KH_ = 0 keeps a runtime-tap fallback).
template <bool VERT, int STATS, int KH_, int KW_>      // STATS: 0 none, 1 stats of y, 2 stats of LeakyReLU(y), 3 stats of act(stat_pre, y),
                                                           // 4: inference epilogue y = post(a[c] * pre(conv + bias) + b[c]) (eval-mode BatchNorm folded in)
__global__ void __launch_bounds__(MB, 2)
k_conv32_mfma(const bf16* __restrict__ x, const bf16* __restrict__ wp, const float* __restrict__ bias, bf16* __restrict__ y,
              int N, int H, int W, int KHr, int KWr, int PH, int PW, int tilesH, int tilesW, int ntiles, int xs, int xo, int ys,
              int yo, int accum, double* __restrict__ stats, int stat_pre, const float* __restrict__ aff, int aff_post,
              const bf16* yadd, int stats_sq_off) {
    // stats_sq_off: the sums of squares go to stats[stats_sq_off ..+32) (32 for a 32-channel tensor; C when this launch covers a 32-channel
    // slab of a C-channel tensor whose statistics buffer is {sum[C], sum of squares[C]} and `stats` points at the slab's first channel)
    // xs/xo, ys/yo: channels per pixel in memory and channel offset of the 32-channel slab read / written; accum: y = result + yadd
    // (yadd == NULL: + the previous contents of y; same layout as y)
    // stats != NULL (STATS): also accumulate per-channel sum / sum-of-squares of pre_act(y) (y as stored, i.e. bf16-rounded) into
    // stats[0..31] / stats[32..63] -- the train-mode BatchNorm statistics of the consumer, fused into this epilogue
    // tile: 16 M-tiles of 32 pixels.  1xk: 8 rows x 64; kx1: 64 x 8 (column-major image); 3x3: 16 rows x 32 -- a squarer tile reads a
    // 1.195x halo instead of 1.29x and wastes 1.4 % instead of 4.3 % of the pixel slots on the 1104-wide level-0 rows
    // (4 x 128 / 128 x 4 tiles for the long 1-D kernels, halo 1.09x instead of 1.19x, measured the same as 8 x 64 / 64 x 8)
    constexpr bool SQ = !VERT && KH_ == 3 && KW_ == 3;
    constexpr int SEGS = SQ ? 1 : 2;                 // 32-pixel segments per tile row (HORZ) / tile column (VERT)
    constexpr bool WIDE = SQ || (KH_ * KW_ >= 1 && KH_ * KW_ <= 11);      // epilogue scratch of 2 KB per wave (does not fit next to 13 taps + the 8 x 76 image)
    constexpr int TH = VERT ? 32 * SEGS : 16 / SEGS, TW = VERT ? 16 / SEGS : 32 * SEGS;
    const int KH = KH_ ? KH_ : KHr, KW = KH_ ? KW_ : KWr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LH = TH + KH - 1, LW = TW + KW - 1;
    const int TAPS = KH * KW;
    unsigned char* sW = smem;
    unsigned char* sX = smem + TAPS * 32 * 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    for (int i = tid; i < TAPS * 32 * 4; i += MB) {
        int row = i >> 2, c = i & 3;
        uint4 v = reinterpret_cast<const uint4*>(wp)[i];
        *reinterpret_cast<uint4*>(sW + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = v;
    }
    // bias lives in LDS (behind the input image) and is re-read in the epilogue: 16 fewer live VGPRs in the MFMA loop
    float* sB = reinterpret_cast<float*>(sX + LH * LW * IPS);
    if (tid < 32) sB[tid] = bias ? bias[tid] : 0.f;
    if (STATS >= 4 && tid < 64) sB[32 + tid] = aff ? aff[tid] : (tid < 32 ? 1.f : 0.f);     // a[32], b[32] of the folded BatchNorm
    unsigned char* sS = reinterpret_cast<unsigned char*>(sB + 96);          // epilogue transpose scratch: 4 waves x 1 KB (2 KB: WIDE)

    // slot geometry (tile independent): slot j covers 16-byte chunk c of tile-local pixel (lr, lc); (lr,lc) packed in one int
    const int c = tid & 3;
    const int npix = LH * LW;
    int s_rc[MAXL];              // the LDS offset and the global offset of a slot are recomputed from (lr, lc): 22 fewer live VGPRs
#pragma unroll
    for (int j = 0; j < MAXL; ++j) {
        int pl = (tid >> 2) + j * (MB / 4);
        bool in = pl < npix;
        int lr = in ? pl / LW : 0x3fff, lc = in ? pl - (pl / LW) * LW : 0;
        s_rc[j] = (lr << 16) | lc;
    }
    // Global loads and stores go through buffer descriptors (one per image of the batch): an out-of-image halo pixel is a lane whose
    // offset lies beyond the descriptor's range -- the hardware returns zeros / drops the store -- so the staging code has NO
    // branches.  That matters beyond the instruction count: with `if (in bounds) load` hipcc branches around every load and, not
    // knowing how many are in flight, waits with vmcnt(0) -- which also waits for the previous tile's STORES (ISA of the first version:
    // `s_barrier; s_waitcnt vmcnt(0)` at the top of every tile, i.e. one HBM write latency per tile and wave on the critical path).
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * (uint32_t)xs * 2u - (uint32_t)xo * 2u;
    const uint32_t out_bytes = (uint32_t)H * (uint32_t)W * (uint32_t)ys * 2u - (uint32_t)yo * 2u;
    u32x4 pre[MAXL];
    auto prefetch = [&](int tile) {
        const int tw = tile & 1023, th = (tile >> 10) & 1023, n = tile >> 20;         // TileSeq::at packing
        const int hb = th * TH - PH, wb = tw * TW - PW;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)n * H * W * xs + xo), 0, img_bytes, 0x00020000);
        if (hb >= 0 && wb >= 0 && hb + LH <= H && wb + LW <= W) {      // interior tile (block-uniform): one add per slot
            const uint32_t base = (uint32_t)((hb * W + wb) * xs * 2 + c * 16), xs2 = (uint32_t)xs * 2u;
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {        // unused slots (lr = 0x3fff) land far beyond the image: out of range by construction
                int rc = s_rc[j];
                asm volatile("" : "+v"(rc));        // opaque: or the per-slot offsets are hoisted out of the tile loop (11 more live VGPRs -> spills,
                                                    // and a scratch reload waits with vmcnt(0), i.e. for the very loads issued just before it)
                const uint32_t lr = (uint32_t)(rc >> 16), lc = (uint32_t)(rc & 0xffff);
                pre[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lr == 0x3fffu ? OOB_OFF : base + (lr * (uint32_t)W + lc) * xs2, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                const int hi = hb + (s_rc[j] >> 16), wi = wb + (s_rc[j] & 0xffff);
                const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;      // unused slots: lr = 0x3fff fails this
                const uint32_t off = ok ? (uint32_t)((hi * W + wi) * xs * 2 + c * 16) : OOB_OFF;
                pre[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            }
        }
    };
    auto stage = [&]() {        // registers of the prefetched tile -> LDS image
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            int rc = s_rc[j];
            asm volatile("" : "+v"(rc));            // opaque, see prefetch(): recompute the LDS address instead of keeping 11 of them live
            const int lr = rc >> 16, lc = rc & 0xffff;
            if (lr != 0x3fff) *reinterpret_cast<u32x4*>(sX + (VERT ? lc * LH + lr : lr * LW + lc) * IPS + c * 16) = pre[j];
        }
    };
    // per-lane fragment bases: weights (swizzled 64-byte rows, row = tap*32 + r: the swizzle term only depends on r) and image
    const int wsw = (r >> 2) & 3;
    const unsigned char* wA0 = sW + r * 64 + ((hh ^ wsw) << 4);
    const unsigned char* wA1 = sW + r * 64 + (((2 + hh) ^ wsw) << 4);
    const unsigned char* xB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int mt = wave * 4 + t;
        int a = mt / SEGS, seg = mt % SEGS;      // HORZ: a = row; VERT: a = col
        int pb = VERT ? a * LH + seg * 32 + r : a * LW + seg * 32 + r;
        xB[t] = sX + pb * IPS + hh * 16;
    }
    float ss[(STATS >= 1 && STATS <= 3) ? 8 : 1], sq[(STATS >= 1 && STATS <= 3) ? 8 : 1];        // BN statistics: after the transpose a lane owns channels 8*(lane&3)..+7
#pragma unroll
    for (int k = 0; k < ((STATS >= 1 && STATS <= 3) ? 8 : 1); ++k) ss[k] = sq[k] = 0.f;
    // Tile pipeline of one block (two blocks per CU alternate):
    //     MFMA(t) | barrier | registers of t+1 -> LDS | issue loads of t+2 | epilogue(t): stores | barrier | MFMA(t+1) ...
    // The wait for the loads of t+1 sits directly behind an MFMA phase: everything still in flight there (those loads, the stores
    // of t-1) was issued at least one whole MFMA phase earlier.  In the first version the wait followed the stores of the SAME tile.
    const TileSeq<VERT> seq(ntiles, tilesH, tilesW);
    int tile = seq.at(0), tile1 = seq.at(1), tile2 = -1;
    if (tile >= 0) {
        prefetch(tile);
        __syncthreads();        // weights / bias staged
        stage();
        if (tile1 >= 0) prefetch(tile1);
    }
    __syncthreads();
    for (int kt = 0; tile >= 0; tile = tile1, tile1 = tile2, ++kt) {
        tile2 = seq.at(kt + 2);
        const int tw = tile & 1023, th = (tile >> 10) & 1023, n = tile >> 20;         // TileSeq::at packing
        const int h0 = th * TH, w0 = tw * TW;

        // the accumulators start at the bias of this lane's 16 output channels (four LDS reads, one wait per tile): the epilogue --
        // where the registers are scarce -- has no bias arithmetic left
        f32x16 acc[4];
        {
            float4 bq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(sB + 8 * q + 4 * hh);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) { acc[t][4 * q] = bq[q].x; acc[t][4 * q + 1] = bq[q].y; acc[t][4 * q + 2] = bq[q].z; acc[t][4 * q + 3] = bq[q].w; }
        }
        // Fragment reads are software-pipelined one stage (= half a tap: 16 of the 32 input channels, 4 MFMAs) ahead of the MFMAs
        // that consume them: left to itself hipcc emits `ds_read; s_waitcnt lgkmcnt(0); v_mfma` per MFMA, exposing the full LDS
        // latency 72 times per tile.  Half-tap stages keep the double buffer at 2 x 20 VGPRs (whole taps: 2 x 40, which spilled
        // in the BN-statistics variant).
        struct Frag { bf16x8 a, b[4]; };
        auto load_stage = [&](Frag& f, int dy, int dx, int half) {
            const int woff = (dy * KW + dx) * 2048;                        // 32 rows x 64 B per tap
            f.a = *reinterpret_cast<const bf16x8*>((half ? wA1 : wA0) + woff);
            const int poff = (VERT ? dx * LH + dy : dy * LW + dx) * IPS + half * 32;
#pragma unroll
            for (int t = 0; t < 4; ++t) f.b[t] = *reinterpret_cast<const bf16x8*>(xB[t] + poff);
        };
        auto mma_stage = [&](const Frag& f) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a, f.b[t], acc[t], 0, 0, 0);
        };
        if (KH_) {
            constexpr int KWc = KW_ ? KW_ : 1;
            constexpr int NS = 2 * (KH_ ? KH_ : 1) * KWc;
            Frag f[2];
            load_stage(f[0], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (k + 1 < NS) load_stage(f[(k + 1) & 1], ((k + 1) >> 1) / KWc, ((k + 1) >> 1) % KWc, (k + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);          // keep the next stage's reads ahead of this stage's MFMAs
                mma_stage(f[k & 1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            Frag f;
            for (int dy = 0; dy < KH; ++dy)
                for (int dx = 0; dx < KW; ++dx) {
                    load_stage(f, dy, dx, 0); mma_stage(f);
                    load_stage(f, dy, dx, 1); mma_stage(f);
                }
        }
        __syncthreads();                                    // every wave has read its fragments: the image may be replaced
        if (tile1 >= 0) stage();
        if (tile2 >= 0) prefetch(tile2);
        // epilogue: lane owns pixel r of each M-tile and channels co = 8q + 4*hh + k.  The packed bf16 values go through a per-wave
        // LDS transpose (chunks XOR-swizzled) so that every lane stores 16 contiguous bytes and one wave instruction writes 16 whole
        // pixels (1 KB contiguous for HORZ tiles) -- four 8-byte stores per lane, i.e. 16 B of every 64-B line per instruction, were
        // the throughput limit of the store-heavy kernels.
        // The first version of this epilogue was one LDS round trip after the other: per M-tile four bias reads, each followed by
        // `s_waitcnt lgkmcnt(0)` and a branch around the `accum` loads, then per 16-pixel round `4 writes; read; wait; store` under an
        // exec-mask branch -- 24 exposed LDS latencies per tile and wave, as long as the whole MFMA phase.  Now: the bias is the initial
        // value of the accumulators, the `accum` add as one block-uniform pre-pass, whole M-tiles per round
        // where the LDS allows it (WIDE: 2 KB of scratch per wave, all lanes write, no exec masking), and the global store of a round
        // is issued AFTER the pack arithmetic of the next one, which hides the round trip.
        const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (int64_t)n * H * W * ys + yo), 0, out_bytes, 0x00020000);
        unsigned char* sc = sS + wave * (WIDE ? 2048 : 1024);
        if (accum) {                                        // block-uniform: y = conv + bias + yadd
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int mt = wave * 4 + t;
                const int a = mt / SEGS, seg = mt % SEGS;
                const int ho = VERT ? h0 + seg * 32 + r : h0 + a;
                const int wo = VERT ? w0 + a : w0 + seg * 32 + r;
                const bool inb = ho < H && wo < W;
                const bf16* yold = (yadd ? yadd : y) + (((int64_t)n * H + (inb ? ho : 0)) * W + (inb ? wo : 0)) * ys + yo + 4 * hh;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f4 old = ld4(yold + 8 * q);
                    acc[t][4 * q] += old.v[0]; acc[t][4 * q + 1] += old.v[1]; acc[t][4 * q + 2] += old.v[2]; acc[t][4 * q + 3] += old.v[3];
                }
                __builtin_amdgcn_sched_barrier(0);          // one M-tile's loads at a time (register pressure)
            }
        }
        constexpr int RND = WIDE ? 1 : 2, NPEND = WIDE ? 2 : 1;
        u32x4 pend[NPEND];
        auto flush = [&](int t, int h2) {       // global stores (+ statistics) of the round whose transposed chunks sit in pend[]
            const int mt = wave * 4 + t;
            const int a = mt / SEGS, seg = mt % SEGS;
            const int p16 = lane >> 2, cch = lane & 3;
#pragma unroll
            for (int u = 0; u < NPEND; ++u) {
                const int pr = 16 * (WIDE ? u : h2) + p16;                      // pixel index inside the M-tile
                const int ho = VERT ? h0 + seg * 32 + pr : h0 + a;
                const int wo = VERT ? w0 + a : w0 + seg * 32 + pr;
                const bool inb = ho < H && wo < W;
                const u32x4 ov = pend[u];
                __builtin_amdgcn_raw_buffer_store_b128(ov, ws, inb ? (uint32_t)((ho * W + wo) * ys * 2 + cch * 16) : OOB_OFF, 0, 0);
                if (STATS >= 1 && STATS <= 3) {
                    if (inb) {
                        const uint32_t wv[4] = {ov[0], ov[1], ov[2], ov[3]};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float u0 = __uint_as_float(wv[k] << 16), u1 = __uint_as_float(wv[k] & 0xffff0000u);
                            if (STATS == 2) { u0 = fmaxf(u0, 0.01f * u0); u1 = fmaxf(u1, 0.01f * u1); }      // LeakyReLU(u) == max(u, 0.01 u): mul + max instead of mul + cmp + select
                            else if (STATS == 3) { u0 = act_fwd(stat_pre, u0); u1 = act_fwd(stat_pre, u1); }
                            ss[2 * k] += u0; sq[2 * k] += u0 * u0; ss[2 * k + 1] += u1; sq[2 * k + 1] += u1 * u1;
                        }
                    }
                }
            }
        };
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint2 o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0 = acc[t][4 * q], v1 = acc[t][4 * q + 1], v2 = acc[t][4 * q + 2], v3 = acc[t][4 * q + 3];
                if (STATS == 4) {
                    const float4 aq = *reinterpret_cast<const float4*>(sB + 32 + 8 * q + 4 * hh);
                    const float4 cq = *reinterpret_cast<const float4*>(sB + 64 + 8 * q + 4 * hh);
                    float v4[4] = {v0, v1, v2, v3};
                    const float a4[4] = {aq.x, aq.y, aq.z, aq.w}, b4[4] = {cq.x, cq.y, cq.z, cq.w};
                    affine4(v4, a4, b4, stat_pre, aff_post);
                    v0 = v4[0]; v1 = v4[1]; v2 = v4[2]; v3 = v4[3];
                }
                o[q].x = pack_bf16x2(v0, v1);
                o[q].y = pack_bf16x2(v2, v3);
            }
#pragma unroll
            for (int h2 = 0; h2 < RND; ++h2) {
                __builtin_amdgcn_sched_barrier(0);          // pack arithmetic above (behind the previous round's reads), stores below
                if (t > 0 || h2 > 0) flush(h2 > 0 ? t : t - 1, h2 > 0 ? h2 - 1 : RND - 1);    // the previous round: its reads are long done
                // one wave writes and reads its own scratch: LDS operations of a wave complete in order (the fences are for the compiler)
                if (WIDE) {
                    const int f = (r >> 1) & 3;
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(sc + r * 64 + ((q ^ f) << 4) + hh * 8) = o[q];
                } else if ((r >> 4) == h2) {
                    const int rr = r & 15, f = (rr >> 1) & 3;
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(sc + rr * 64 + ((q ^ f) << 4) + hh * 8) = o[q];
                }
                wave_lds_fence();
                const int p16 = lane >> 2, cch = lane & 3;
#pragma unroll
                for (int u = 0; u < NPEND; ++u)
                    pend[u] = *reinterpret_cast<const u32x4*>(sc + (16 * u + p16) * 64 + ((cch ^ ((p16 >> 1) & 3)) << 4));
                wave_lds_fence();       // the next round's writes come after every lane's read of this round
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        flush(3, RND - 1);
        __syncthreads();                                    // the staged image of the next tile is complete
    }
    if (STATS >= 1 && STATS <= 3) {
        // lanes with equal (lane & 3) hold different pixels of the same 8 channels: butterfly over lane bits 2..5, LDS, fp64 atomics
        // (once per block: LDS float atomics are slow, a per-tile flush of the partials cost +70 % kernel time)
        __syncthreads();
        float* red = reinterpret_cast<float*>(sX);          // [4 waves][64]
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float a = ss[k], b = sq[k];
#pragma unroll
            for (int o = 32; o > 2; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if (lane < 4) {             // per-wave slots, summed below: no LDS atomics
                red[wave * 64 + 8 * lane + k] = a;
                red[wave * 64 + 32 + 8 * lane + k] = b;
            }
        }
        __syncthreads();
        if (tid < 64) atomicAdd(&stats[tid < 32 ? tid : stats_sq_off + tid - 32], (double)red[tid] + (double)red[64 + tid] + (double)red[128 + tid] + (double)red[192 + tid]);
    }
}

/* x, y: bf16 NHWC [N,H,W,32]; wp: packed bf16 [KH*KW][32][32] from tcct_conv32_pack_weights; stride 1; output size == input
 * size requires PH = (KH-1)/2 etc. but any PH <= KH-1, PW <= KW-1 with "same" output extent H x W is accepted. */
static bool conv32_fwd33_stream_launch(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, double* stats, int stat_code, int stats_sq_off,
                                       bool force, hipStream_t st, const float* aff = nullptr);
static bool conv32_fwd1k_stream_launch(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int K, bool force, hipStream_t st);
static bool conv32_fwdk1_stream_launch(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int K, bool force, hipStream_t st);
static int g_fwd_mode = -1;
/* 0 (default): plain 3x3 convolutions whose waves get >= FS_MIN_RUN rows take the row-stream kernel; 1: the tiled kernel for every
 * shape; 2: the row-stream kernel for every plain 3x3 (the comparison arms of the bit-identity test).  Returns the previous mode; mode < 0 only queries. */
extern "C" int64_t tcct_conv32_fwd_mode(int mode) {
    if (g_fwd_mode < 0) g_fwd_mode = 0;
    const int prev = g_fwd_mode;
    if (mode >= 0 && mode <= 2) g_fwd_mode = mode;
    return prev;
}
static int conv32_fwd_impl(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW,
                           int PH, int PW, int xs, int xo, int ys, int yo, int accum, double* stats, int stat_pre, tcct_stream_t stream,
                           const float* aff = nullptr, int aff_post = 0, bool affine = false, const void* yadd = nullptr, int stats_sq_off = 32);
extern "C" int tcct_conv32_fwd(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW,
                               int PH, int PW, tcct_stream_t stream) {
    return conv32_fwd_impl(x, wp, bias, y, N, H, W, KH, KW, PH, PW, 32, 0, 32, 0, 0, nullptr, 0, stream);
}
/* forward + fused BatchNorm statistics of the consumer: stats[64] (fp64, must be zero on entry) += {sum, sum of squares} per
 * output channel of pre_act(y) with y as stored (bf16) -- replaces a separate tcct_bn_stats pass over y */
extern "C" int tcct_conv32_fwd_bnstats(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW,
                                       int PH, int PW, double* stats, int pre_act, tcct_stream_t stream) {
    return conv32_fwd_impl(x, wp, bias, y, N, H, W, KH, KW, PH, PW, 32, 0, 32, 0, 0, stats, pre_act, stream);
}
/* inference: y = post_act(a[c] * pre_act(conv(x) + bias[c]) + b[c]) with ab = {a[32], b[32]} (eval-mode BatchNorm folded into
 * the epilogue, tcct_bn_eval_ab; ab == NULL: a = 1, b = 0, i.e. only the activations) -- no separate normalisation pass */
extern "C" int tcct_conv32_fwd_affine(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW,
                                      int PH, int PW, const float* ab, int pre_act, int post_act, tcct_stream_t stream) {
    return conv32_fwd_impl(x, wp, bias, y, N, H, W, KH, KW, PH, PW, 32, 0, 32, 0, 0, nullptr, pre_act, stream, ab, post_act, true);
}
/* y = conv(x) + bias + res (res: bf16 NHWC [N,H,W,32], may not overlap y): the input gradient of a convolution whose input has a
 * second consumer -- that consumer's gradient is added in this epilogue instead of in a separate accumulation pass */
extern "C" int tcct_conv32_fwd_add(const void* x, const void* wp, const float* bias, const void* res, void* y, int N, int H, int W,
                                   int KH, int KW, int PH, int PW, tcct_stream_t stream) {
    TCCT_CHECK(res != nullptr && res != y, "conv32_fwd_add: res must be a separate tensor");
    return conv32_fwd_impl(x, wp, bias, y, N, H, W, KH, KW, PH, PW, 32, 0, 32, 0, 1, nullptr, 0, stream, nullptr, 0, false, res);
}
/* same kernel on 32-channel slabs of wider tensors: x has xs channels/pixel (slab at xo), y has ys (slab at yo); accumulate=1
 * adds into y.  Used to run 32->64 / 64->32 convolutions (MPViT stem[1], nets/tcct.py:682-689) as 32x32 sub-GEMMs. */
extern "C" int tcct_conv32_fwd_strided(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH,
                                       int KW, int PH, int PW, int xs, int xo, int ys, int yo, int accumulate,
                                       tcct_stream_t stream) {
    TCCT_CHECK(xs % 8 == 0 && xo % 8 == 0 && ys % 4 == 0 && yo % 4 == 0 && xo + 32 <= xs && yo + 32 <= ys, "conv32_fwd_strided: bad slab");
    return conv32_fwd_impl(x, wp, bias, y, N, H, W, KH, KW, PH, PW, xs, xo, ys, yo, accumulate, nullptr, 0, stream);
}
/* tcct_conv32_fwd_strided + the statistics of the train-mode BatchNorm behind the WIDE convolution (MPViT stem[1]: 32 -> 64, 3x3, reference
 * nets/tcct.py:682-689 / :80): call it for the LAST input slab of an output slab (the accumulated value is what gets counted); stats fp64
 * {sum[ys], sum of squares[ys]} of the whole ys-channel tensor, zero before the first slab */
extern "C" int tcct_conv32_fwd_strided_bnstats(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH,
                                               int KW, int PH, int PW, int xs, int xo, int ys, int yo, int accumulate, double* stats,
                                               int pre_act, tcct_stream_t stream) {
    TCCT_CHECK(xs % 8 == 0 && xo % 8 == 0 && ys % 4 == 0 && yo % 4 == 0 && xo + 32 <= xs && yo + 32 <= ys && stats, "conv32_fwd_strided_bnstats: bad slab");
    return conv32_fwd_impl(x, wp, bias, y, N, H, W, KH, KW, PH, PW, xs, xo, ys, yo, accumulate, stats + yo, pre_act, stream, nullptr, 0, false, nullptr, ys);
}
/* inference: one 32-channel OUTPUT slab of a wider convolution with a single 32-channel input slab (MPViT stem[1]: 32 -> 64, 3x3, nets/tcct.py:682-689) with the
 * eval-mode BatchNorm + activation of those 32 channels in the epilogue: y[..., yo:yo+32] = post_act(a[c] * pre_act(conv(x[..., xo:xo+32]) + bias[c]) + b[c]),
 * ab = {a[32], b[32]} of the slab's channels (tcct_bn_eval_ab on the 32-channel slices of the BatchNorm's tensors).  Replaces tcct_conv32_fwd_strided + tcct_bn_apply. */
extern "C" int tcct_conv32_fwd_strided_affine(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW, int PH, int PW, int xs,
                                              int xo, int ys, int yo, const float* ab, int pre_act, int post_act, tcct_stream_t stream) {
    TCCT_CHECK(xs % 8 == 0 && xo % 8 == 0 && ys % 4 == 0 && yo % 4 == 0 && xo + 32 <= xs && yo + 32 <= ys, "conv32_fwd_strided_affine: bad slab");
    return conv32_fwd_impl(x, wp, bias, y, N, H, W, KH, KW, PH, PW, xs, xo, ys, yo, 0, nullptr, pre_act, stream, ab, post_act, true);
}
static int conv32_fwd_impl(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW,
                           int PH, int PW, int xs, int xo, int ys, int yo, int accum, double* stats, int stat_pre, tcct_stream_t stream,
                           const float* aff, int aff_post, bool affine, const void* yadd, int stats_sq_off) {
    TCCT_CHECK(KH >= 1 && KW >= 1 && KH * KW <= 13 * 13, "conv32_fwd: bad kernel %dx%d", KH, KW);
    TCCT_CHECK(2 * PH == KH - 1 && 2 * PW == KW - 1, "conv32_fwd: only 'same' padding (got pad %d,%d for %dx%d)", PH, PW, KH, KW);
    const bool vert = (KW == 1 && KH > 1);
    const bool sq = !vert && KH == 3 && KW == 3;
    const int segs = sq ? 1 : 2;
    const int TH = vert ? 32 * segs : 16 / segs, TW = vert ? 16 / segs : 32 * segs;
    const int LH = TH + KH - 1, LW = TW + KW - 1;
    const bool wide = sq || ((KH == 1 || KW == 1) && (KH * KW == 1 || KH * KW == 11));        // == the kernel's WIDE for the compile-time shapes
    size_t lds = (size_t)KH * KW * 32 * 64 + (size_t)LH * LW * IPS + 384 + (wide ? 8192 : 4096);
    TCCT_CHECK(lds <= 80 * 1024, "conv32_fwd: %dx%d needs %zu B of LDS (> 80 KiB for 2 blocks/CU)", KH, KW, lds);
    TCCT_CHECK(LH * LW * 4 <= MAXL * MB, "conv32_fwd: %dx%d tile image exceeds the staging slots", KH, KW);
    int tilesH = (H + TH - 1) / TH, tilesW = (W + TW - 1) / TW;
    int64_t nt = (int64_t)N * tilesH * tilesW;
    TCCT_CHECK(nt > 0 && nt < (1LL << 31), "conv32_fwd: bad tile count");
    TCCT_CHECK(tilesH < 1024 && tilesW < 1024 && N < 2048, "conv32_fwd: %d images of %d x %d tiles exceed the packed tile id (2047 images, 1023 x 1023 tiles)", N, tilesH, tilesW);
    TCCT_CHECK((int64_t)H * W * xs * 2 < (1LL << 31) && (int64_t)H * W * ys * 2 < (1LL << 31),
               "conv32_fwd: one image of %d x %d x %d channels exceeds the 2 GiB buffer-descriptor range", H, W, xs > ys ? xs : ys);
    int grid = (int)(nt < 512 ? nt : 512);  // (256 / 384 / 768 blocks: 0.323 / 0.294 / 0.292 ms against 0.236 with 512 = two resident blocks per CU)
    hipStream_t st = (hipStream_t)stream;
    if (g_fwd_mode < 0) (void)tcct_conv32_fwd_mode(-1);
    const int aff_code = !affine ? -1 : (stat_pre == TCCT_ACT_NONE && aff_post == TCCT_ACT_NONE) ? 4 : (stat_pre == TCCT_ACT_NONE && aff_post == TCCT_ACT_LRELU) ? 5 :
                         (stat_pre == TCCT_ACT_LRELU && aff_post == TCCT_ACT_NONE) ? 6 : 0;      // 0: an epilogue the row-stream kernel has no instance for
    if (g_fwd_mode != 1 && sq && xs == 32 && xo == 0 && ys == 32 && yo == 0 && !accum && !yadd &&
        (affine ? aff_code > 0 : (!stats || stat_pre == TCCT_ACT_NONE || stat_pre == TCCT_ACT_LRELU))) {
        if (conv32_fwd33_stream_launch(x, wp, bias, y, N, H, W, affine ? nullptr : stats, affine ? aff_code : (!stats ? 0 : (stat_pre == TCCT_ACT_NONE ? 1 : 2)), stats_sq_off,
                                       g_fwd_mode == 2, st, aff)) TCCT_LAUNCH_OK();
    }
    if (g_fwd_mode != 1 && KH == 1 && (KW == 13 || KW == 11 || KW == 9) && xs == 32 && xo == 0 && ys == 32 && yo == 0 && !accum && !affine && !yadd && !stats) {
        if (conv32_fwd1k_stream_launch(x, wp, bias, y, N, H, W, KW, g_fwd_mode == 2, st)) TCCT_LAUNCH_OK();
    }
    if (g_fwd_mode != 1 && KW == 1 && (KH == 13 || KH == 11 || KH == 9) && xs == 32 && xo == 0 && ys == 32 && yo == 0 && !accum && !affine && !yadd && !stats) {
        if (conv32_fwdk1_stream_launch(x, wp, bias, y, N, H, W, KH, g_fwd_mode == 2, st)) TCCT_LAUNCH_OK();
    }
#define CF_LAUNCH(V, S, KHT, KWT)                                                                                           \
    do {                                                                                                                    \
        static bool attr = false;                                                                                           \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_mfma<V, S, KHT, KWT>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_conv32_mfma<V, S, KHT, KWT>), dim3(grid), dim3(MB), lds, st, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, N, H, W, \
                           KH, KW, PH, PW, tilesH, tilesW, (int)nt, xs, xo, ys, yo, accum, stats, stat_pre, aff, aff_post, (const bf16*)yadd, stats_sq_off);                \
    } while (0)
#define CF_S(V, KHT, KWT)                                                                                   \
    do {                                                                                                    \
        if (affine) CF_LAUNCH(V, 4, KHT, KWT);                                                              \
        else if (!stats) CF_LAUNCH(V, 0, KHT, KWT);                                                           \
        else if (stat_pre == TCCT_ACT_NONE) CF_LAUNCH(V, 1, KHT, KWT);                                      \
        else if (stat_pre == TCCT_ACT_LRELU) CF_LAUNCH(V, 2, KHT, KWT);                                     \
        else CF_LAUNCH(V, 3, KHT, KWT);                                                                     \
    } while (0)
    // compile-time taps for the shapes that carry the time (3x3 everywhere; the level-0/1 cross convolutions); generic otherwise
    if (KH == 3 && KW == 3) CF_S(false, 3, 3);
    else if (KH == 1 && KW == 13) CF_S(false, 1, 13);
    else if (KH == 13 && KW == 1) CF_S(true, 13, 1);
    else if (KH == 1 && KW == 11) CF_S(false, 1, 11);
    else if (KH == 11 && KW == 1) CF_S(true, 11, 1);
    else if (KH == 1 && KW == 1) CF_S(false, 1, 1);
    else if (vert) CF_S(true, 0, 0);
    else CF_S(false, 0, 0);
#undef CF_S
#undef CF_LAUNCH
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[tap][co][ci] = sum_pixels dy[p][co] * x[p@tap][ci]  ->  per tap one 32x32 MFMA accumulator, K = pixels.
// Both operands need "8 consecutive pixels of one channel" per lane, i.e. the TRANSPOSE of the pixel-major NHWC rows:
// gfx950's ds_read_b64_tr_b16 does that transpose on the LDS read (4 pixels x 16 channels per 16-lane group), so the
// tiles are staged exactly like in the forward kernel (64 B per pixel, XOR-swizzled chunks) and any tap shift keeps the
// 8-byte alignment the instruction needs.  4 consecutive pixels = 256 contiguous bytes -> every bank once: conflict-free.
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

// Both wgrad images are read ONLY through transposing reads, which need no swizzle (4 consecutive pixels = 256 contiguous bytes
// hit every bank once): pixel rows stay linear (64 B), so a read address is `lane base + pixel*64` and the second half of the
// fragment (+4 pixels) is an immediate offset.  (PMC of the first version: 1690 VALU instructions per tile and wave, mostly
// swizzle arithmetic in front of the reads.)
__device__ __forceinline__ const unsigned char* tr_lane_base(const unsigned char* img, int lane) {
    const int i = lane & 15, q = i >> 2, pp = i & 3, g = lane >> 4;
    const int hh = g >> 1, cb = g & 1;
    return img + (8 * hh + q) * 64 + (16 * cb + 4 * pp) * 2;
}
// acc + v[2j] + v[2j+1] in one instruction (v_dot2c_f32_bf16 with a packed (1, 1) operand; exact products, fp32 accumulation)
typedef __bf16 bf16x2_v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2_ones(const bf16x8& v, int j, float acc) {
    const bf16x2_v p = {v[2 * j], v[2 * j + 1]};
    const bf16x2_v one = {(__bf16)1.0f, (__bf16)1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(p, one, acc, false);
}
__device__ __forceinline__ bf16x8 tr_load8p(const unsigned char* p) {
    // p = tr_lane_base(img) + P*64 for 16 consecutive LDS pixels P..P+15; returns pixels P+8*(lane>>5)+j (j=0..7) of channel lane&31
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 256));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

template <int TPW, bool VERT, bool SQ = false>      // SQ: 16 x 32 tiles (3x3; smaller x halo, fewer wasted pixel slots), else 8 x 64 / 64 x 8
__global__ void __launch_bounds__(MB, 2)
k_conv32_wgrad(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
               int N, int H, int W, int KH, int KW, int PH, int PW, int TG, int tilesH, int tilesW, int ntiles, int xs, int xo,
               int ds, int dof, int ldi, int o_off, int i_off) {
    constexpr int TH = VERT ? 64 : (SQ ? 16 : 8), TW = VERT ? 8 : (SQ ? 32 : 64);
    constexpr int CPR = (VERT ? TH : TW) / 16;     // 16-pixel chunks per tile row (HORZ) / column (VERT)
    constexpr int DSL = TH * TW * 4 / MB;          // dy slots per thread (8)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LH = TH + KH - 1, LW = TW + KW - 1;
    const int TAPS = KH * KW;
    unsigned char* sX = smem;
    unsigned char* sD = smem + LH * LW * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int WPG = 4 / TG;
    const int tg = wave / WPG, wi = wave % WPG;
    const int tap0 = tg * TPW;
    int poff[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        int tap = tap0 + t;
        int dy_ = tap / KW, dx_ = tap - dy_ * KW;
        poff[t] = (tap < KH * KW) ? (VERT ? dx_ * LH + dy_ : dy_ * LW + dx_) * 64 : 0;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float bsum = 0.f;
    const unsigned char* lbX = tr_lane_base(sX, lane);
    const unsigned char* lbD = tr_lane_base(sD, lane);

    // staging slots (see k_conv32_mfma): x image slots and dy image slots, geometry fixed per thread
    const int c = tid & 3;
    const int npix = LH * LW;
    int s_rc[MAXL];
#pragma unroll
    for (int j = 0; j < MAXL; ++j) {
        int pl = (tid >> 2) + j * (MB / 4);
        bool in = pl < npix;
        int lr = in ? pl / LW : 0x3fff, lc = in ? pl - (pl / LW) * LW : 0;
        s_rc[j] = (lr << 16) | lc;
    }
    // Global loads go through buffer descriptors like in k_conv32_mfma: an out-of-image pixel is an offset beyond the descriptor's range
    // (the hardware returns zeros), so the staging code has no branches.  The first version tested every slot (`if (in image) load`): two
    // branches and ~30 instructions of 64-bit address arithmetic per load, ~570 instructions per tile and wave in front of an MFMA phase of
    // 80 MFMAs -- as long as the phase itself.  Interior tiles (block-uniform test) now cost one add per slot.
    const uint32_t ximg_bytes = (uint32_t)H * (uint32_t)W * (uint32_t)xs * 2u - (uint32_t)xo * 2u;
    const uint32_t dimg_bytes = (uint32_t)H * (uint32_t)W * (uint32_t)ds * 2u - (uint32_t)dof * 2u;
    u32x4 prex[MAXL], pred[DSL];
    auto prefetch = [&](int tile) {
        const int tw = tile & 1023, th = (tile >> 10) & 1023, n = tile >> 20;         // TileSeq::at packing
        const int h0 = th * TH, w0 = tw * TW;
        const int hb = h0 - PH, wb = w0 - PW;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)n * H * W * xs + xo), 0, ximg_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)n * H * W * ds + dof), 0, dimg_bytes, 0x00020000);
        const uint32_t xs2 = (uint32_t)xs * 2u, ds2 = (uint32_t)ds * 2u;
        if (hb >= 0 && wb >= 0 && hb + LH <= H && wb + LW <= W) {       // interior tile: x halo and dy tile lie inside the image
            const uint32_t xbase = (uint32_t)((hb * W + wb) * xs * 2 + c * 16);
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                int rc = s_rc[j];
                asm volatile("" : "+v"(rc));        // opaque: recompute the offsets per tile instead of keeping them live across the MFMA phase
                const uint32_t lr = (uint32_t)(rc >> 16), lc = (uint32_t)(rc & 0xffff);
                prex[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, lr == 0x3fffu ? OOB_OFF : xbase + (lr * (uint32_t)W + lc) * xs2, 0, 0);
            }
            const uint32_t dbase = (uint32_t)((h0 * W + w0) * ds * 2 + c * 16);
#pragma unroll
            for (int j = 0; j < DSL; ++j) {
                const uint32_t pl = (uint32_t)((tid >> 2) + j * (MB / 4));           // TW is a power of two: shifts, no tables
                pred[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, dbase + ((pl / TW) * (uint32_t)W + (pl & (TW - 1))) * ds2, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < MAXL; ++j) {
                const int hi = hb + (s_rc[j] >> 16), wi_ = wb + (s_rc[j] & 0xffff);
                const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi_ < (unsigned)W;      // unused slots: lr = 0x3fff fails this
                prex[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (uint32_t)((hi * W + wi_) * xs * 2 + c * 16) : OOB_OFF, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < DSL; ++j) {
                const int pl = (tid >> 2) + j * (MB / 4);
                const int ho = h0 + pl / TW, wo = w0 + (pl & (TW - 1));
                pred[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, (ho < H && wo < W) ? (uint32_t)((ho * W + wo) * ds * 2 + c * 16) : OOB_OFF, 0, 0);
            }
        }
    };
    const TileSeq<VERT> seq(ntiles, tilesH, tilesW);
    int tile = seq.at(0), tile1 = -1;
    if (tile >= 0) prefetch(tile);
    for (int kt = 0; tile >= 0; tile = tile1, ++kt) {
        tile1 = seq.at(kt + 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            int rc = s_rc[j];
            asm volatile("" : "+v"(rc));            // opaque: the LDS address is recomputed per tile, not kept live (11 VGPRs)
            const int lr = rc >> 16, lc = rc & 0xffff;
            if (lr != 0x3fff) *reinterpret_cast<u32x4*>(sX + (VERT ? lc * LH + lr : lr * LW + lc) * 64 + c * 16) = prex[j];
        }
#pragma unroll
        for (int j = 0; j < DSL; ++j) {
            const int pl = (tid >> 2) + j * (MB / 4);
            const int p = VERT ? (pl & (TW - 1)) * TH + pl / TW : pl;
            *reinterpret_cast<u32x4*>(sD + p * 64 + c * 16) = pred[j];
        }
        __syncthreads();
        if (tile1 >= 0) prefetch(tile1);
        // chunks of 16 pixels; fragments of chunk i+1 are read while the MFMAs of chunk i run (sched_barrier pins the order)
        struct WFrag { bf16x8 a, b[TPW]; };
        auto load_chunk = [&](WFrag& f, int ch) {
            const int a_ = ch / CPR, s16 = (ch % CPR) * 16;   // HORZ: row / col offset; VERT: col / row offset
            const int Pd = VERT ? a_ * TH + s16 : a_ * TW + s16;
            const int Px = VERT ? a_ * LH + s16 : a_ * LW + s16;
            f.a = tr_load8p(lbD + Pd * 64);
            const unsigned char* px = lbX + Px * 64;
#pragma unroll
            for (int t = 0; t < TPW; ++t) f.b[t] = tr_load8p(px + poff[t]);
        };
        auto mma_chunk = [&](const WFrag& f) {
            if (tg == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bsum = dot2_ones(f.a, j, bsum);      // 4 x v_dot2c_f32_bf16 instead of 8 x (unpack + add): the bias sum was 12 % of the compute side
            }
#pragma unroll
            for (int t = 0; t < TPW; ++t)
                if (tap0 + t < TAPS) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a, f.b[t], acc[t], 0, 0, 0);
        };
        if (TPW <= 4) {
            WFrag f0, f1;                   // named buffers: a runtime-indexed array of fragments would live in scratch
            load_chunk(f0, wi);
            for (int ch = wi; ch < 32; ch += 2 * WPG) {           // 32 / WPG is even
                load_chunk(f1, ch + WPG);
                __builtin_amdgcn_sched_barrier(0);
                mma_chunk(f0);
                __builtin_amdgcn_sched_barrier(0);
                if (ch + 2 * WPG < 32) load_chunk(f0, ch + 2 * WPG);
                __builtin_amdgcn_sched_barrier(0);
                mma_chunk(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {                            // 5 accumulators: a second fragment set would spill; read all 12 fragments, then 5 MFMAs
            // (round 3: a rotating-register form -- the x fragment of tap t reloaded for the next chunk right behind the MFMA that consumed it --
            // hid the LDS round trip per chunk and changed nothing: 0.176 ms with the loads ablated either way, 0.25-0.27 in full)
            WFrag f0;
            for (int ch = wi; ch < 32; ch += WPG) {
                load_chunk(f0, ch);
                __builtin_amdgcn_sched_barrier(0);
                mma_chunk(f0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // block-level reduction of the WPG = 4/TG partial accumulators per tap in LDS: the waves of a tap group take turns (plain
    // stores / read-add-write, one barrier per turn).  LDS float atomics did this before and cost ~25 us per BLOCK (a one-tile
    // launch took 37 us against 9 us for the forward kernel), i.e. most of the time of every call below level 1.
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    for (int turn = 0; turn < WPG; ++turn) {
        if (wi == turn) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tap = tap0 + t;
                if (tap < TAPS) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int co = (k & 3) + 8 * (k >> 2) + 4 * hh;
                        float* dst = &red[tap * 1024 + co * 32 + r];
                        *dst = turn == 0 ? acc[t][k] : *dst + acc[t][k];
                    }
                }
            }
        }
        __syncthreads();
    }
    // OIHW-linear order: consecutive lanes add to consecutive addresses (256 contiguous bytes per wave instruction; a
    // tap-major walk would scatter every lane into its own 64-byte segment: 8x the atomic traffic, PMC WRITE_SIZE 151 MB)
    for (int i = tid; i < TAPS * 1024; i += MB) {
        const int tap = i % TAPS, cc = i / TAPS;          // cc = co*32 + ci
        const int co = cc >> 5, ci = cc & 31;
        atomicAdd(&dw[((int64_t)(o_off + co) * ldi + i_off + ci) * TAPS + tap], red[tap * 1024 + cc]);
    }
    if (dbias && tg == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 32) atomicAdd(&dbias[o_off + r], bsum);
    }
}

// (Rounds 2-3 carried three more forms of this kernel -- tiles DMA'd global -> LDS into two buffers of one 8-wave block per CU, and two 3x3 forms that
// derived the dx = 1, 2 operands from one fragment by funnel shifts + v_permlane32_swap.  All bit-compatible, none faster inside the step; round 4
// deleted them when the rolling-row form below became the default.  What they taught is in DESIGN.md 3: the DMA path needs ~400 bookkeeping
// instructions per tile and wave, and cutting LDS reads alone does not move a kernel that waits on one tile in flight per block.)
// ------------------------------------------------------------------------------------------------ 3x3 weight gradient, rolling rows (round 4)
// The register-staged kernel above reads every x pixel from LDS once per TAP and every dy pixel once per tap group: 192 transposing reads per
// tile and wave, 393 KB per 72 KB tile and block -- at two blocks per CU the LDS array is ~80 % busy at the rate HBM could feed the kernel
// (its loads alone take 0.171 ms, the whole kernel 0.234).  The LDS-DMA forms cut the reads with funnel shifts across the dx taps but paid for it
// in per-tile bookkeeping.  This form cuts them with NO extra arithmetic, along the other axis: a wave owns ONE tap column dx and one 16-pixel
// column chunk of the 16 x 32 tile and walks down the 18 halo rows; the x fragment of halo row a is the operand of tap dy = 0 for output row a,
// of dy = 1 for row a - 1 and of dy = 2 for row a - 2, so it is read ONCE and multiplied with the three dy fragments of a rolling window that
// lives in registers.  Per tile and wave: 18 x + 16 dy fragments (68 transposing reads instead of 192), 48 MFMAs, three accumulators.  Six waves
// (3 tap columns x 2 column chunks) per block, two blocks per CU: 204 fragment reads per tile and block instead of 384, the matrix pipes see
// the same 288 MFMAs.  Staging: the register-prefetch path of k_conv32_wgrad with 96 pixels per slot round.  Results are bit-compatible
// with the other forms up to the order of the fp32 atomics.
#define WR_T 384
#define WR_XL 7                 // x staging slots per thread: 18 x 34 = 612 halo pixels / 96
#define WR_DL 6                 // dy staging slots: 512 / 96 (the last one partial)
__global__ void __launch_bounds__(WR_T, 3)      // HIP: second argument = waves per SIMD (2 blocks x 6 waves = 3 per SIMD: <= 168 VGPRs)
k_conv32_wgrad33_roll(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
                      int N, int H, int W, int tilesH, int tilesW, int ntiles) {
    constexpr int TH = 16, TW = 32, LH = 18, LW = 34, NPX = LH * LW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sD = smem + NPX * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int g = wave >> 1, wi = wave & 1;             // tap column dx = g, column chunk wi
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float bsum = 0.f;
    const unsigned char* xcol = tr_lane_base(sX, lane) + (wi * 16 + g) * 64;
    const unsigned char* dcol = tr_lane_base(sD, lane) + wi * 16 * 64;
    const int c = tid & 3;
    // halo pixel of staging slot j: pl = tid / 4 + 96 j -> (row, column) by a multiply-shift division (pl < 1024: pl * 241 >> 13 == pl / 34), recomputed
    // per use: a table of 7 packed coordinates per thread pushed the kernel over the 168-VGPR budget of three waves per SIMD, and the 8-byte
    // scratch reload that followed sat in the MIDDLE of the prefetch burst with an s_waitcnt vmcnt(0) in front of the remaining loads
    auto slot_rc = [&](int j, uint32_t& lr, uint32_t& lc) {
        uint32_t pl = (uint32_t)(tid >> 2) + (uint32_t)(j * (WR_T / 4));
        asm volatile("" : "+v"(pl));
        lr = (pl * 241u) >> 13;
        lc = pl - lr * (uint32_t)LW;
        return pl < (uint32_t)NPX;
    };
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    u32x4 prex[WR_XL], pred[WR_DL];
    auto prefetch = [&](int tile) {
        const int tw = tile & 1023, th = (tile >> 10) & 1023, n = tile >> 20;
        const int h0 = th * TH, w0 = tw * TW;
        const int hb = h0 - 1, wb = w0 - 1;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        if (hb >= 0 && wb >= 0 && hb + LH <= H && wb + LW <= W) {       // interior tile
            const uint32_t xbase = (uint32_t)((hb * W + wb) * 64 + c * 16);
#pragma unroll
            for (int j = 0; j < WR_XL; ++j) {
                uint32_t lr, lc;
                const bool in = slot_rc(j, lr, lc);
                prex[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, in ? xbase + (lr * (uint32_t)W + lc) * 64u : OOB_OFF, 0, 0);
            }
            const uint32_t dbase = (uint32_t)((h0 * W + w0) * 64 + c * 16);
#pragma unroll
            for (int j = 0; j < WR_DL; ++j) {
                const uint32_t pl = (uint32_t)((tid >> 2) + j * (WR_T / 4));
                pred[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, pl < (uint32_t)(TH * TW) ? dbase + ((pl >> 5) * (uint32_t)W + (pl & 31u)) * 64u : OOB_OFF, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < WR_XL; ++j) {
                uint32_t lr, lc;
                const bool in = slot_rc(j, lr, lc);
                const int hi = hb + (int)lr, wi_ = wb + (int)lc;
                const bool ok = in && (unsigned)hi < (unsigned)H && (unsigned)wi_ < (unsigned)W;
                prex[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (uint32_t)((hi * W + wi_) * 64 + c * 16) : OOB_OFF, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < WR_DL; ++j) {
                const int pl = (tid >> 2) + j * (WR_T / 4);
                const int ho = h0 + (pl >> 5), wo = w0 + (pl & 31);
                pred[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, (pl < TH * TW && ho < H && wo < W) ? (uint32_t)((ho * W + wo) * 64 + c * 16) : OOB_OFF, 0, 0);
            }
        }
    };
    const TileSeq<false> seq(ntiles, tilesH, tilesW);
    int tile = seq.at(0), tile1 = -1;
    if (tile >= 0) prefetch(tile);
    for (int kt = 0; tile >= 0; tile = tile1, ++kt) {
        tile1 = seq.at(kt + 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WR_XL; ++j) {       // halo pixels are stored in (row, column) order: LDS pixel index == pl
            const int pl = (tid >> 2) + j * (WR_T / 4);
            if (pl < NPX) *reinterpret_cast<u32x4*>(sX + pl * 64 + c * 16) = prex[j];
        }
#pragma unroll
        for (int j = 0; j < WR_DL; ++j) {
            const int pl = (tid >> 2) + j * (WR_T / 4);
            if (pl < TH * TW) *reinterpret_cast<u32x4*>(sD + pl * 64 + c * 16) = pred[j];
        }
        __syncthreads();
        if (tile1 >= 0) prefetch(tile1);
        // walk down the 18 halo rows: X[a] meets D[a] (tap dy 0), D[a-1] (dy 1), D[a-2] (dy 2); fragments of row a + 1 are requested before the MFMAs of row a
        bf16x8 D[3], Xc, Xn, Dn;
        Xc = tr_load8p(xcol);
        D[0] = tr_load8p(dcol);
#pragma unroll
        for (int a = 0; a < LH; ++a) {
            if (a + 1 < LH) Xn = tr_load8p(xcol + (a + 1) * LW * 64);
            if (a + 1 < TH) Dn = tr_load8p(dcol + (a + 1) * TW * 64);
            __builtin_amdgcn_sched_barrier(0);
            if (a < TH) {
                if (g == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bsum = dot2_ones(D[a % 3], j, bsum);
                }
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(D[a % 3], Xc, acc[0], 0, 0, 0);
            }
            if (a >= 1 && a - 1 < TH) acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(D[(a + 2) % 3], Xc, acc[1], 0, 0, 0);
            if (a >= 2) acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(D[(a + 1) % 3], Xc, acc[2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            Xc = Xn;
            if (a + 1 < TH) D[(a + 1) % 3] = Dn;
        }
    }
    // the two column-chunk waves of a tap column hold partial sums of the same three taps: they take turns in LDS (no LDS float atomics)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    for (int turn = 0; turn < 2; ++turn) {
        if (wi == turn) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const int tap = t * 3 + g;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int co = (k & 3) + 8 * (k >> 2) + 4 * hh;
                    float* dst = &red[tap * 1024 + co * 32 + r];
                    *dst = turn == 0 ? acc[t][k] : *dst + acc[t][k];
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < 9 * 1024; i += WR_T) {
        const int tap = i % 9, cc = i / 9;          // cc = co*32 + ci: OIHW-linear order, 256 contiguous bytes per wave instruction
        atomicAdd(&dw[(int64_t)cc * 9 + tap], red[tap * 1024 + cc]);
    }
    if (dbias && g == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        __syncthreads();
        if (lane < 32) red[wi * 32 + r] = bsum;
    } else __syncthreads();
    __syncthreads();
    if (dbias && tid < 32) atomicAdd(&dbias[tid], red[tid] + red[32 + tid]);
}

// ------------------------------------------------------------------------------------------------ 3x3 weight gradient, wave-private row streams (round 4)
// Every tiled form above keeps ONE tile per block in flight and hands it over through block barriers: the bytes in flight per CU swing between a full
// tile set and nothing, and the three forms tie at ~0.23 ms whatever their LDS / instruction budget is.  Here nothing is shared between waves: a wave
// owns a 16-pixel-wide column strip of one image and walks down its rows.  Row a of the strip (18 x pixels with the two halo columns + 16 dy pixels,
// 2176 B) travels global -> LDS by three `buffer_load_dwordx4 ... lds` pieces into a wave-private ring of WS_R rows; WS_P rows are always in flight
// (issued one per row consumed: `s_waitcnt vmcnt(3 (WS_P - 1))` retires exactly the next row), so the memory pipe sees a steady stream instead of tile
// bursts.  Per row: the three dx operands are three transposing reads of the same LDS row at pixel offsets 0, 1, 2, the dy fragments of rows a, a - 1,
// a - 2 roll through registers, nine MFMAs into nine accumulators (all taps in one wave: 144 accumulator registers, two waves per SIMD).  No block
// barrier until the final reduction.  Out-of-image rows / columns and the dy rows behind a segment's end are buffer-range misses (the DMA writes zeros).
// Work: the strips of all images form one sequence of N * strips * H rows, cut into equal runs of rows per wave (a run may continue into the next strip).
#define WS_T 256
#define WS_R 9                  // ring rows per wave (a multiple of 3: ring slot and dy-window position are compile-time in the 9-fold unrolled body)
#define WS_P 7                  // rows in flight
#define WS_ROWB 2176            // 18 x pixels (1152 B) + 16 dy pixels (1024 B)
#define WS_MIN_RUN 32            // rows per wave below which the pipeline fill (7 rows per segment) costs more than the tiles' barriers: levels 0-1 stream, 2-4 roll
__device__ __forceinline__ void lds_dma16(const u32x4& rsrc, uint32_t voff, uint32_t lds_addr) {
    // 64 lanes x 16 B from per-lane buffer offsets to LDS bytes [lds_addr + 16 lane ..): inline asm on purpose -- the builtin form makes hipcc treat the DMA as a
    // pending LDS store and drain it (vmcnt(0)) in front of the next ds_read.  M0 (the LDS base) is compiler-reserved: saved and restored.
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ u32x4 make_rsrc_words(const void* base, uint32_t bytes) {
    const uint64_t b = (uint64_t)base;
    u32x4 d;
    d[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    d[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    d[2] = __builtin_amdgcn_readfirstlane(bytes);
    d[3] = 0x00020000u;
    return d;
}
__global__ void __launch_bounds__(WS_T, 2)
k_conv32_wgrad33_stream(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
                        int N, int H, int W, int strips, int run) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    unsigned char* ring = smem + wave * (WS_R * WS_ROWB);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
    const unsigned char* xb = tr_lane_base(ring, lane);
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float bsum = 0.f;
    // the four waves of a block walk four ADJACENT strips over the same rows (4 KB of every x / dy row between them at about the same time)
    const int sgroups = (strips + 3) >> 2;
    const int64_t total = (int64_t)N * sgroups * H;
    int64_t cur = (int64_t)blockIdx.x * run;        // (a row-major block order -- the strip groups of one run of image rows on consecutive blocks -- pays for the
    const int64_t end = cur + run < total ? cur + run : total;      //  forward kernel below, not here: 0.176 ms either way)
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    const uint32_t rowb = (uint32_t)W * 64u;
    const int px = lane >> 2, c = lane & 3;
    while (cur < end) {
        const int sidx = (int)(cur / H), r0 = (int)(cur - (int64_t)sidx * H);
        const int n = sidx / sgroups, s = (sidx - n * sgroups) * 4 + wave;
        const int left = (int)(end - cur);
        const int L = __builtin_amdgcn_readfirstlane(H - r0 < left ? H - r0 : left);      // dy rows r0 .. r0 + L - 1; x rows r0 - 1 .. r0 + L
        if (s >= strips) { cur += L; continue; }
        const int w0 = s * 16;
        const u32x4 rx = make_rsrc_words(x + (int64_t)n * H * W * 32, img_bytes);
        const u32x4 rd = make_rsrc_words(dy + (int64_t)n * H * W * 32, img_bytes);
        const int cx1 = w0 - 1 + px, cx2 = w0 + 15 + px, cd = w0 + px;
        const uint32_t ox1 = (cx1 >= 0 && cx1 < W) ? (uint32_t)(cx1 * 64 + c * 16) : OOB_OFF;
        const uint32_t ox2 = (cx2 < W) ? (uint32_t)(cx2 * 64 + c * 16) : OOB_OFF;
        const uint32_t od = (cd < W) ? (uint32_t)(cd * 64 + c * 16) : OOB_OFF;
        const uint32_t row0 = (uint32_t)(r0 - 1) * rowb;            // r0 = 0: wraps far beyond the descriptor range -> zeros (the padding row)
        // halo row a of the segment -> ring slot `slot`; rows behind the segment (a > L + 1) and dy rows a >= L are all-miss pieces (zeros, no HBM traffic)
        auto issue = [&](int a, int slot) {
            const uint32_t ro = row0 + (uint32_t)a * rowb;
            const bool xin = a <= L + 1, din = a < L;
            const uint32_t base = ring_lds + (uint32_t)(slot * WS_ROWB);
            lds_dma16(rx, xin ? ox1 + ro : OOB_OFF, base);
            if (lane < 8) lds_dma16(rx, xin ? ox2 + ro : OOB_OFF, base + 1024u);
            lds_dma16(rd, din ? od + ro + rowb : OOB_OFF, base + 1152u);
        };
#pragma unroll
        for (int a = 0; a < WS_P; ++a) issue(a, a);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (WS_P - 1)) : "memory");
        bf16x8 X[3], D[3], Xn[3], Dn;
#pragma unroll
        for (int d = 0; d < 3; ++d) X[d] = tr_load8p(xb + d * 64);
        D[0] = tr_load8p(xb + 1152);
#pragma unroll
        for (int k = 0; k < 8; ++k) { D[1][k] = (__bf16)0.f; D[2][k] = (__bf16)0.f; }
        const int groups = (L + 2 + WS_R - 1) / WS_R;
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int j = 0; j < WS_R; ++j) {
                const int a = g * WS_R + j;
                issue(a + WS_P, (j + WS_P) % WS_R);
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (WS_P - 1)) : "memory");         // row a + 1 has landed
                const unsigned char* nx = xb + ((j + 1) % WS_R) * WS_ROWB;
#pragma unroll
                for (int d = 0; d < 3; ++d) Xn[d] = tr_load8p(nx + d * 64);
                Dn = tr_load8p(nx + 1152);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) bsum = dot2_ones(D[j % 3], q, bsum);
#pragma unroll
                for (int t = 0; t < 3; ++t)             // tap row t: x halo row a against dy row a - t
#pragma unroll
                    for (int d = 0; d < 3; ++d) acc[t * 3 + d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(D[(j + 3 - t) % 3], X[d], acc[t * 3 + d], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 3; ++d) X[d] = Xn[d];
                D[(j + 1) % 3] = Dn;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the all-miss pieces behind the segment: the next segment reuses their slots
        cur += L;
    }
    // the four waves of a block take turns adding their nine 32 x 32 partial sums into one LDS image (no LDS float atomics), then one atomic per element
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    for (int turn = 0; turn < WS_T / 64; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int co = (k & 3) + 8 * (k >> 2) + 4 * hh;
                    float* dst = &red[t * 1024 + co * 32 + r];
                    *dst = turn == 0 ? acc[t][k] : *dst + acc[t][k];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < 9 * 1024; i += WS_T) {
        const int tap = i % 9, cc = i / 9;          // OIHW-linear order: 256 contiguous bytes per wave instruction
        atomicAdd(&dw[(int64_t)cc * 9 + tap], red[tap * 1024 + cc]);
    }
    __syncthreads();
    if (dbias) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 32) red[wave * 32 + r] = bsum;
        __syncthreads();
        if (tid < 32) atomicAdd(&dbias[tid], red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid]);
    }
}

// ------------------------------------------------------------------------------------------------ 3x3 forward / input gradient, wave-private row streams (round 4)
// The row-stream structure of k_conv32_wgrad33_stream for the convolution itself (same arithmetic as k_conv32_mfma<false, STATS, 3, 3> on plain 32-channel
// tensors, results BIT-IDENTICAL: every output pixel sums bias, then taps in (dy, dx, half) order, on the same MFMA).  A wave owns a 32-pixel-wide strip and
// walks down it; halo row a (34 pixels, 2176 B, three LDS-DMA pieces) feeds the three output rows a - 2, a - 1, a, whose accumulators roll through
// registers: six B fragments (3 dx x 2 halves of the input channels) are read ONCE per halo row instead of once per output row, the 18 weight fragments
// live in registers for the whole kernel (no weight image in LDS), output row a - 2 is complete after the row's MFMAs and leaves through the per-wave
// transpose (in the ring slot that is free at that moment) as two 1 KB stores.  The DMA pieces put chunk c of ring pixel P at position c ^ ((P >> 2) & 3)
// (the permutation is applied to the SOURCE offsets): conflict-free ds_read_b128 at any dx on linear 64-byte pixel rows.  Every row issues
// exactly 3 DMA pieces; `s_waitcnt vmcnt(3 (FS_P - 1))` retires the next row whatever the stores in between do.
#define FS_T 256
#define FS_R 9
#define FS_P 7
#define FS_ROWB 2176            // 34 halo pixels x 64 B
#define FS_MIN_RUN 24
#ifndef FS_SYNC
#define FS_SYNC 1
#endif
template <int STATS>            // 0: none; 1: statistics of y; 2: of LeakyReLU(y) (y as stored) -> stats[0..31], stats[stats_sq_off ..+32) (fp64 atomics);
                                // 4 / 5 / 6: inference epilogue y = post(a[c] * pre(conv + bias) + b[c]) (eval-mode BatchNorm folded in, as k_conv32_mfma<.., 4, ..>) with
                                // (pre, post) = (none, none) / (none, LeakyReLU) / (LeakyReLU, none) -- compile-time: the run-time switch of affine4(), unrolled 9 x 4
                                // times, made an 18 000-line kernel with scratch that ran SLOWER than the tiled one
__global__ void __launch_bounds__(FS_T, 2)      // two waves per SIMD: <= 256 VGPRs
k_conv32_fwd33_stream(const bf16* __restrict__ x, const bf16* __restrict__ wp, const float* __restrict__ bias, bf16* __restrict__ y,
                      int N, int H, int W, int strips, int run, int rpi, double* __restrict__ stats, int stats_sq_off,
                      const float* __restrict__ aff) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nw = blockDim.x >> 6;             // waves = adjacent strips per block (4)
    unsigned char* ring = smem + wave * (FS_R * FS_ROWB);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
    float* sB = reinterpret_cast<float*>(smem + nw * FS_R * FS_ROWB);          // bias[32], then the statistics partials [nw][64]
    bf16x8 Wf[9][2];            // A operands: row co = r, input channels 8 hh + 16 half ..+7 of tap t
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) Wf[t][kc] = *reinterpret_cast<const bf16x8*>(wp + (t * 32 + r) * 32 + (hh + 2 * kc) * 8);
    if (tid < 32) sB[tid] = bias ? bias[tid] : 0.f;
    if (STATS >= 4 && tid < 64) sB[32 + tid] = aff ? aff[tid] : (tid < 32 ? 1.f : 0.f);     // a[32], b[32] of the folded BatchNorm (the statistics partials' place)
    __syncthreads();
    const unsigned char* xB[3][2];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) { const int P = r + d; xB[d][kc] = ring + P * 64 + (((hh + 2 * kc) ^ ((P >> 2) & 3)) << 4); }
    constexpr bool ST = STATS >= 1 && STATS <= 2;
    float ss[ST ? 8 : 1], sq[ST ? 8 : 1];
#pragma unroll
    for (int k = 0; k < (ST ? 8 : 1); ++k) ss[k] = sq[k] = 0.f;
    // the waves of a block walk ADJACENT strips over the same rows (nw x 2 KB of every image row between them, at about the same time)
    const int sgroups = (strips + nw - 1) / nw;
    // Row-major block order: consecutive blocks take the strip groups of ONE run of image rows, so whole image rows are in flight together (0.1838 ms at
    // level 0 against 0.1893 for runs cut from the strip-major sequence of rows).  rpi = runs per image; a block has one run of one strip group.
    int64_t cur, end;
    {
        const int q = blockIdx.x / sgroups, grp = blockIdx.x - q * sgroups;
        const int n_ = q / rpi, r_ = (q - n_ * rpi) * run;
        cur = ((int64_t)n_ * sgroups + grp) * H + r_;
        end = cur + (H - r_ < run ? H - r_ : run);
    }
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    const uint32_t rowb = (uint32_t)W * 64u;
    const int pq = lane >> 2, cs = (lane & 3) ^ ((lane >> 4) & 3);          // LDS position lane & 3 of ring pixel 16 piece + pq holds chunk cs
    const int p16 = lane >> 2, cch = lane & 3;
    while (cur < end) {
        const int sidx = (int)(cur / H), r0 = (int)(cur - (int64_t)sidx * H);
        const int n = sidx / sgroups, s = (sidx - n * sgroups) * nw + wave;
        const int left = (int)(end - cur);
        const int L = __builtin_amdgcn_readfirstlane(H - r0 < left ? H - r0 : left);      // output rows r0 .. r0 + L - 1; halo rows r0 - 1 .. r0 + L
        if (s >= strips) {
#if FS_SYNC
            for (int g = 0; g < (L + 2 + FS_R - 1) / FS_R; ++g) __builtin_amdgcn_s_barrier();
#endif
            cur += L; continue;
        }
        const int w0 = s * 32;
        const u32x4 rx = make_rsrc_words(x + (int64_t)n * H * W * 32, img_bytes);
        const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        const int c0 = w0 - 1 + pq, c1 = w0 + 15 + pq, c2 = w0 + 31 + pq;
        const uint32_t o0 = (c0 >= 0 && c0 < W) ? (uint32_t)(c0 * 64 + cs * 16) : OOB_OFF;
        const uint32_t o1 = (c1 < W) ? (uint32_t)(c1 * 64 + cs * 16) : OOB_OFF;
        const uint32_t o2 = (c2 < W) ? (uint32_t)(c2 * 64 + cs * 16) : OOB_OFF;
        uint32_t so[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int wo = w0 + 16 * u + p16; so[u] = wo < W ? (uint32_t)(wo * 64 + cch * 16) : OOB_OFF; }
        const uint32_t row0 = (uint32_t)(r0 - 1) * rowb;
        auto issue = [&](int a, int slot) {
            const uint32_t ro = row0 + (uint32_t)a * rowb;
            const bool xin = a <= L + 1;
            const uint32_t base = ring_lds + (uint32_t)(slot * FS_ROWB);
            lds_dma16(rx, xin ? o0 + ro : OOB_OFF, base);
            lds_dma16(rx, xin ? o1 + ro : OOB_OFF, base + 1024u);
            if (lane < 8) lds_dma16(rx, xin ? o2 + ro : OOB_OFF, base + 2048u);
        };
#pragma unroll
        for (int a = 0; a < FS_P; ++a) issue(a, a);
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (FS_P - 1)) : "memory");
        bf16x8 X[3][2], Xn[3][2];
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) X[d][kc] = *reinterpret_cast<const bf16x8*>(xB[d][kc]);
        f32x16 acc[3];
        const int groups = (L + 2 + FS_R - 1) / FS_R;
        for (int g = 0; g < groups; ++g) {
#if FS_SYNC
            __builtin_amdgcn_s_barrier();           // keeps the four strips of a block on the same rows (no data is shared)
#endif
#pragma unroll
            for (int j = 0; j < FS_R; ++j) {
                const int a = g * FS_R + j;
                issue(a + FS_P, (j + FS_P) % FS_R);
                // Halo row a + 1 has landed once at most the 3 (FS_P - 1) younger DMA pieces are outstanding.  The 2 (FS_P - 1) younger stores are NOT added to the
                // count: loads retire in order among loads, but a store whose lanes are all out of range (rows behind the segment, the upper half of a narrow last
                // strip) is dropped and acknowledged at once -- counting it let the fragment reads of the first rows of a segment overtake their DMA (full-size test).
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (FS_P - 1)) : "memory");
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) Xn[d][kc] = *reinterpret_cast<const bf16x8*>(xB[d][kc] + ((j + 1) % FS_R) * FS_ROWB);
                {           // output row a starts at the bias of this lane's 16 channels
                    float4 bq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(sB + 8 * q + 4 * hh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[j % 3][4 * q] = bq[q].x; acc[j % 3][4 * q + 1] = bq[q].y; acc[j % 3][4 * q + 2] = bq[q].z; acc[j % 3][4 * q + 3] = bq[q].w; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)          // halo row a is tap row dy of output row a - dy
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int kc = 0; kc < 2; ++kc)
                            acc[(j + 3 - dy) % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[dy * 3 + d][kc], X[d][kc], acc[(j + 3 - dy) % 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // output row a - 2 is complete: pack, transpose through the free ring slot (the one halo row a + 8 will land in), two 1 KB stores
                const int orow = a - 2;
                const bool ovalid = orow >= 0 && orow < L;
                const f32x16& A = acc[(j + 1) % 3];
                unsigned char* sc = ring + ((j + 8) % FS_R) * FS_ROWB;
                uint2 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v4[4] = {A[4 * q], A[4 * q + 1], A[4 * q + 2], A[4 * q + 3]};
                    if (STATS >= 4) {
                        const float4 aq = *reinterpret_cast<const float4*>(sB + 32 + 8 * q + 4 * hh);
                        const float4 cq = *reinterpret_cast<const float4*>(sB + 64 + 8 * q + 4 * hh);
                        const float a4[4] = {aq.x, aq.y, aq.z, aq.w}, b4[4] = {cq.x, cq.y, cq.z, cq.w};
                        affine4_c<(STATS == 6 ? TCCT_ACT_LRELU : TCCT_ACT_NONE), (STATS == 5 ? TCCT_ACT_LRELU : TCCT_ACT_NONE)>(v4, a4, b4);
                    }
                    o[q].x = pack_bf16x2(v4[0], v4[1]); o[q].y = pack_bf16x2(v4[2], v4[3]);
                }
                const int f = (r >> 1) & 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(sc + r * 64 + ((q ^ f) << 4) + hh * 8) = o[q];
                wave_lds_fence();
                u32x4 pend[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) pend[u] = *reinterpret_cast<const u32x4*>(sc + (16 * u + p16) * 64 + ((cch ^ ((p16 >> 1) & 3)) << 4));
                wave_lds_fence();
                const uint32_t oro = (uint32_t)(r0 + orow) * rowb;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool inb = ovalid && so[u] != OOB_OFF;
                    __builtin_amdgcn_raw_buffer_store_b128(pend[u], ws, inb ? so[u] + oro : OOB_OFF, 0, 0);
                    if (ST) {
                        if (inb) {
                            const uint32_t wv[4] = {pend[u][0], pend[u][1], pend[u][2], pend[u][3]};
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                float u0 = __uint_as_float(wv[k] << 16), u1 = __uint_as_float(wv[k] & 0xffff0000u);
                                if (STATS == 2) { u0 = fmaxf(u0, 0.01f * u0); u1 = fmaxf(u1, 0.01f * u1); }
                                ss[2 * k] += u0; sq[2 * k] += u0 * u0; ss[2 * k + 1] += u1; sq[2 * k + 1] += u1 * u1;
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int d = 0; d < 3; ++d)
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) X[d][kc] = Xn[d][kc];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur += L;
    }
    if (ST) {
        // a lane owns channels 8 (lane & 3) ..+7 of the pixels it stored: butterfly over lane bits 2..5, per-wave LDS slots, fp64 atomics (as k_conv32_mfma)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* red = sB + 32;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float a = ss[k], b = sq[k];
#pragma unroll
            for (int o = 32; o > 2; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if (lane < 4) {
                red[wave * 64 + 8 * lane + k] = a;
                red[wave * 64 + 32 + 8 * lane + k] = b;
            }
        }
        __syncthreads();
        if (tid < 64) {
            double t = 0.0;
            for (int w = 0; w < nw; ++w) t += (double)red[w * 64 + tid];
            atomicAdd(&stats[tid < 32 ? tid : stats_sq_off + tid - 32], t);
        }
    }
}
/* plain 32-channel 3x3 (no slabs, no accumulate, STATS 0-2): true when the row-stream kernel was launched */
static bool conv32_fwd33_stream_launch(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, double* stats, int stat_code, int stats_sq_off,
                                       bool force, hipStream_t st, const float* aff) {
    const int strips = (W + 31) / 32;
    const int nw = 4;           // 4 waves x 2 blocks per CU (7 or 8 waves in one block per CU: 0.197 ms against 0.190 at level 0)
    const int sg = (strips + nw - 1) / nw;
    // runs per image: as many as fill the 512 block slots, each at least FS_MIN_RUN rows long (the pipeline fill is FS_P rows per run)
    int rpi = 512 / (N * sg);
    if (rpi > H / FS_MIN_RUN) rpi = H / FS_MIN_RUN;
    if (rpi < 1) rpi = 1;
    const int run = (H + rpi - 1) / rpi;
    rpi = (H + run - 1) / run;
    const int blocks = N * rpi * sg;
    if (!force && (H < FS_MIN_RUN || blocks < 384)) return false;       // small maps: the tiled kernel
    const size_t lds = (size_t)nw * FS_R * FS_ROWB + 128 + (size_t)nw * 256;
#define FS_LAUNCH(S)                                                                                                                                          \
    do {                                                                                                                                                      \
        static bool attr = false;                                                                                                                             \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_fwd33_stream<S>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_conv32_fwd33_stream<S>), dim3((unsigned)blocks), dim3(64 * nw), lds, st, (const bf16*)x, (const bf16*)wp, bias, (bf16*)y, N, H, W, strips, \
                           run, rpi, stats, stats_sq_off, aff);                                                                                                  \
    } while (0)
    if (stat_code == 0) FS_LAUNCH(0);
    else if (stat_code == 1) FS_LAUNCH(1);
    else if (stat_code == 2) FS_LAUNCH(2);
    else if (stat_code == 4) FS_LAUNCH(4);
    else if (stat_code == 5) FS_LAUNCH(5);
    else FS_LAUNCH(6);
#undef FS_LAUNCH
    tcct_census_hit(TCCT_CENSUS_FWD33_STREAM);
    return true;
}

// ------------------------------------------------------------------------------------------------ 1 x K forward / input gradient, wave-private row streams (round 4)
// The horizontal cross convolutions (reference nets/tcct.py:814-818; K = 13, 11, 9 at levels 0, 1, 2) in the row-stream structure of k_conv32_fwd33_stream.  A row is
// self-contained here (no rolling window): halo row = 32 + K - 1 pixels, one accumulator, 2 K MFMAs per row in the tiled kernel's (dx, half) order on the same
// MFMA => BIT-IDENTICAL to k_conv32_mfma<false, 0, 1, K>.  The 2 K weight fragments and the 2 K per-lane fragment addresses live in registers; ring of 7 rows, 5 in flight.
#define F1_R 7
#define F1_P 5
template <int K>
__global__ void __launch_bounds__(FS_T, 2)
k_conv32_fwd1k_stream(const bf16* __restrict__ x, const bf16* __restrict__ wp, const float* __restrict__ bias, bf16* __restrict__ y,
                      int N, int H, int W, int strips, int run, int rpi) {
    constexpr int LWP = 32 + K - 1, ROWB = LWP * 64, TAIL = LWP - 32;         // halo pixels per row, ring bytes per row, pixels of the third DMA piece
    static_assert(TAIL >= 1 && TAIL <= 16 && ROWB >= 2048, "1 x K row streams: 3 <= K <= 17");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nw = blockDim.x >> 6;
    unsigned char* ring = smem + wave * (F1_R * ROWB);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
    float* sB = reinterpret_cast<float*>(smem + nw * F1_R * ROWB);
    bf16x8 Wf[K][2];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) Wf[t][kc] = *reinterpret_cast<const bf16x8*>(wp + (t * 32 + r) * 32 + (hh + 2 * kc) * 8);
    if (tid < 32) sB[tid] = bias ? bias[tid] : 0.f;
    __syncthreads();
    uint32_t xo_[K][2];          // byte offset in a ring row of this lane's fragment of tap dx, half kc (chunk c of ring pixel P sits at position c ^ ((P >> 2) & 3))
#pragma unroll
    for (int d = 0; d < K; ++d)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) { const int P = r + d; xo_[d][kc] = (uint32_t)(P * 64 + (((hh + 2 * kc) ^ ((P >> 2) & 3)) << 4)); }
    const int sgroups = (strips + nw - 1) / nw;
    int64_t cur, end;
    {
        const int q = blockIdx.x / sgroups, grp = blockIdx.x - q * sgroups;
        const int n_ = q / rpi, r_ = (q - n_ * rpi) * run;
        cur = ((int64_t)n_ * sgroups + grp) * H + r_;
        end = cur + (H - r_ < run ? H - r_ : run);
    }
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    const uint32_t rowb = (uint32_t)W * 64u;
    const int pq = lane >> 2, cs = (lane & 3) ^ ((lane >> 4) & 3);
    const int p16 = lane >> 2, cch = lane & 3;
    constexpr int PADL = (K - 1) / 2;
    while (cur < end) {
        const int sidx = (int)(cur / H), r0 = (int)(cur - (int64_t)sidx * H);
        const int n = sidx / sgroups, s = (sidx - n * sgroups) * nw + wave;
        const int left = (int)(end - cur);
        const int L = __builtin_amdgcn_readfirstlane(H - r0 < left ? H - r0 : left);      // rows r0 .. r0 + L - 1
        if (s >= strips) { cur += L; continue; }
        const int w0 = s * 32;
        const u32x4 rx = make_rsrc_words(x + (int64_t)n * H * W * 32, img_bytes);
        const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        const int c0 = w0 - PADL + pq, c1 = c0 + 16, c2 = c0 + 32;
        const uint32_t o0 = (c0 >= 0 && c0 < W) ? (uint32_t)(c0 * 64 + cs * 16) : OOB_OFF;
        const uint32_t o1 = (c1 >= 0 && c1 < W) ? (uint32_t)(c1 * 64 + cs * 16) : OOB_OFF;
        const uint32_t o2 = (c2 < W) ? (uint32_t)(c2 * 64 + cs * 16) : OOB_OFF;
        uint32_t so[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int wo = w0 + 16 * u + p16; so[u] = wo < W ? (uint32_t)(wo * 64 + cch * 16) : OOB_OFF; }
        const uint32_t row0 = (uint32_t)r0 * rowb;
        auto issue = [&](int a, int slot) {
            const uint32_t ro = row0 + (uint32_t)a * rowb;
            const bool xin = a < L;
            const uint32_t base = ring_lds + (uint32_t)(slot * ROWB);
            lds_dma16(rx, xin ? o0 + ro : OOB_OFF, base);
            lds_dma16(rx, xin ? o1 + ro : OOB_OFF, base + 1024u);
            if (lane < 4 * TAIL) lds_dma16(rx, xin ? o2 + ro : OOB_OFF, base + 2048u);
        };
#pragma unroll
        for (int a = 0; a < F1_P; ++a) issue(a, a);
        const int groups = (L + F1_R - 1) / F1_R;
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int j = 0; j < F1_R; ++j) {
                const int a = g * F1_R + j;
                // row a has landed once only the 3 (F1_P - 1) pieces of rows a + 1 .. a + F1_P - 1 are outstanding (stores are not counted: k_conv32_fwd33_stream)
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * (F1_P - 1)) : "memory");
                const unsigned char* rowp = ring + j * ROWB;
                f32x16 acc;
                {
                    float4 bq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(sB + 8 * q + 4 * hh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[4 * q] = bq[q].x; acc[4 * q + 1] = bq[q].y; acc[4 * q + 2] = bq[q].z; acc[4 * q + 3] = bq[q].w; }
                }
                bf16x8 F[2][2];
                F[0][0] = *reinterpret_cast<const bf16x8*>(rowp + xo_[0][0]);
                F[0][1] = *reinterpret_cast<const bf16x8*>(rowp + xo_[0][1]);
#pragma unroll
                for (int d = 0; d < K; ++d) {
                    if (d + 1 < K) {
                        F[(d + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(rowp + xo_[d + 1][0]);
                        F[(d + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(rowp + xo_[d + 1][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[d][0], F[d & 1][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[d][1], F[d & 1][1], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the DMA of row a + F1_P goes to slot (j + F1_P) % F1_R; slot (j + F1_R - 1) % F1_R (row a - 1, consumed) is the transpose scratch of this row
                issue(a + F1_P, (j + F1_P) % F1_R);
                const bool ovalid = a < L;
                unsigned char* sc = ring + ((j + F1_R - 1) % F1_R) * ROWB;
                uint2 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { o[q].x = pack_bf16x2(acc[4 * q], acc[4 * q + 1]); o[q].y = pack_bf16x2(acc[4 * q + 2], acc[4 * q + 3]); }
                const int f = (r >> 1) & 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(sc + r * 64 + ((q ^ f) << 4) + hh * 8) = o[q];
                wave_lds_fence();
                u32x4 pend[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) pend[u] = *reinterpret_cast<const u32x4*>(sc + (16 * u + p16) * 64 + ((cch ^ ((p16 >> 1) & 3)) << 4));
                wave_lds_fence();
                const uint32_t oro = (uint32_t)(r0 + a) * rowb;
#pragma unroll
                for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_buffer_store_b128(pend[u], ws, (ovalid && so[u] != OOB_OFF) ? so[u] + oro : OOB_OFF, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur += L;
    }
}
/* plain 32-channel 1 x K (K = 13, 11, 9; no slabs, no accumulate, no statistics): true when the row-stream kernel was launched */
static bool conv32_fwd1k_stream_launch(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int K, bool force, hipStream_t st) {
    const int strips = (W + 31) / 32;
    const int nw = 4;
    const int sg = (strips + nw - 1) / nw;
    int rpi = 512 / (N * sg);
    if (rpi > H / FS_MIN_RUN) rpi = H / FS_MIN_RUN;
    if (rpi < 1) rpi = 1;
    const int run = (H + rpi - 1) / rpi;
    rpi = (H + run - 1) / run;
    const int blocks = N * rpi * sg;
    if (!force && (H < FS_MIN_RUN || blocks < 384)) return false;
#define F1_LAUNCH(KK)                                                                                                                                          \
    do {                                                                                                                                                      \
        static bool attr = false;                                                                                                                             \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_fwd1k_stream<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_conv32_fwd1k_stream<KK>), dim3((unsigned)blocks), dim3(64 * nw), (size_t)nw * F1_R * (32 + KK - 1) * 64 + 128, st, (const bf16*)x, (const bf16*)wp, \
                           bias, (bf16*)y, N, H, W, strips, run, rpi);                                                                                      \
    } while (0)
    if (K == 13) F1_LAUNCH(13);
    else if (K == 11) F1_LAUNCH(11);
    else if (K == 9) F1_LAUNCH(9);
    else return false;
#undef F1_LAUNCH
    return true;
}

// ------------------------------------------------------------------------------------------------ K x 1 forward / input gradient, wave-private row streams (round 4)
// The vertical cross convolutions.  A strip row is exactly 32 pixels (no column halo: two DMA pieces); output row o needs the K halo rows o .. o + K - 1, so the B
// fragments of the last K rows stay in a REGISTER window (K x 2 fragments, 104 VGPRs for K = 13; the unrolled body's position j = a mod K names the registers) and
// each halo row is read from LDS exactly once; the 2 K weight fragments come from a block-shared LDS image (the tiled kernel's swizzled rows) per output row.
// MFMAs in the tiled kernel's (dy, half) order => BIT-IDENTICAL to k_conv32_mfma<true, 0, K, 1>.  Ring of 6 rows per wave (runtime slot index), 4 in flight.
#define FV_R 6
#define FV_P 4
template <int K>
__global__ void __launch_bounds__(FS_T, 2)
k_conv32_fwdk1_stream(const bf16* __restrict__ x, const bf16* __restrict__ wp, const float* __restrict__ bias, bf16* __restrict__ y,
                      int N, int H, int W, int strips, int run, int rpi) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int nw = blockDim.x >> 6;
    unsigned char* sW = smem;                                   // K taps x 32 rows x 64 B, chunks XOR-swizzled by (row >> 2) & 3
    unsigned char* ring = smem + K * 2048 + wave * (FV_R * 2048);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
    float* sB = reinterpret_cast<float*>(smem + K * 2048 + nw * FV_R * 2048);
    for (int i = tid; i < K * 32 * 4; i += blockDim.x) {
        const int row = i >> 2, c = i & 3;
        *reinterpret_cast<uint4*>(sW + row * 64 + ((c ^ ((row >> 2) & 3)) << 4)) = reinterpret_cast<const uint4*>(wp)[i];
    }
    if (tid < 32) sB[tid] = bias ? bias[tid] : 0.f;
    __syncthreads();
    const int wsw = (r >> 2) & 3;
    const unsigned char* wA[2] = {sW + r * 64 + ((hh ^ wsw) << 4), sW + r * 64 + (((2 + hh) ^ wsw) << 4)};
    uint32_t xo_[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) xo_[kc] = (uint32_t)(r * 64 + (((hh + 2 * kc) ^ ((r >> 2) & 3)) << 4));
    const int sgroups = (strips + nw - 1) / nw;
    int64_t cur, end;
    {
        const int q = blockIdx.x / sgroups, grp = blockIdx.x - q * sgroups;
        const int n_ = q / rpi, r_ = (q - n_ * rpi) * run;
        cur = ((int64_t)n_ * sgroups + grp) * H + r_;
        end = cur + (H - r_ < run ? H - r_ : run);
    }
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    const uint32_t rowb = (uint32_t)W * 64u;
    const int pq = lane >> 2, cs = (lane & 3) ^ ((lane >> 4) & 3);
    const int p16 = lane >> 2, cch = lane & 3;
    constexpr int PADT = (K - 1) / 2;
    while (cur < end) {
        const int sidx = (int)(cur / H), r0 = (int)(cur - (int64_t)sidx * H);
        const int n = sidx / sgroups, s = (sidx - n * sgroups) * nw + wave;
        const int left = (int)(end - cur);
        const int L = __builtin_amdgcn_readfirstlane(H - r0 < left ? H - r0 : left);      // output rows r0 .. r0 + L - 1; halo rows r0 - PADT .. r0 + L - 1 + PADT
        if (s >= strips) { cur += L; continue; }
        const int w0 = s * 32;
        const u32x4 rx = make_rsrc_words(x + (int64_t)n * H * W * 32, img_bytes);
        const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        const int c0 = w0 + pq, c1 = c0 + 16;
        const uint32_t o0 = c0 < W ? (uint32_t)(c0 * 64 + cs * 16) : OOB_OFF;
        const uint32_t o1 = c1 < W ? (uint32_t)(c1 * 64 + cs * 16) : OOB_OFF;
        uint32_t so[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int wo = w0 + 16 * u + p16; so[u] = wo < W ? (uint32_t)(wo * 64 + cch * 16) : OOB_OFF; }
        const uint32_t row0 = (uint32_t)(r0 - PADT) * rowb;        // rows above the image wrap far beyond the descriptor range: zeros (the padding rows)
        const int nh = L + K - 1;                                   // halo rows of the run
        int slot_i = 0, slot_d = FV_P % FV_R;                       // ring slots of the row consumed / issued next (runtime: K and FV_R are coprime)
        auto issue = [&](int a, int slot) {
            const uint32_t ro = row0 + (uint32_t)a * rowb;
            const bool xin = a < nh;
            const uint32_t base = ring_lds + (uint32_t)(slot * 2048);
            lds_dma16(rx, xin ? o0 + ro : OOB_OFF, base);
            lds_dma16(rx, xin ? o1 + ro : OOB_OFF, base + 1024u);
        };
#pragma unroll
        for (int a = 0; a < FV_P; ++a) issue(a, a);
        bf16x8 Xw[K][2];
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
            for (int k = 0; k < 8; ++k) { Xw[t][0][k] = (__bf16)0.f; Xw[t][1][k] = (__bf16)0.f; }
        const int groups = (nh + K - 1) / K;
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int a = g * K + j;
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (FV_P - 1)) : "memory");         // halo row a has landed (DMA pieces only are counted)
                const unsigned char* rowp = ring + slot_i * 2048;
                Xw[j][0] = *reinterpret_cast<const bf16x8*>(rowp + xo_[0]);
                Xw[j][1] = *reinterpret_cast<const bf16x8*>(rowp + xo_[1]);
                f32x16 acc;
                {
                    float4 bq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(sB + 8 * q + 4 * hh);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc[4 * q] = bq[q].x; acc[4 * q + 1] = bq[q].y; acc[4 * q + 2] = bq[q].z; acc[4 * q + 3] = bq[q].w; }
                }
                // output row o = a - (K - 1): tap dy reads halo row o + dy = window position (j + 1 + dy) mod K
                bf16x8 Af[2][2];
                Af[0][0] = *reinterpret_cast<const bf16x8*>(wA[0]);
                Af[0][1] = *reinterpret_cast<const bf16x8*>(wA[1]);
#pragma unroll
                for (int dy = 0; dy < K; ++dy) {
                    if (dy + 1 < K) {
                        Af[(dy + 1) & 1][0] = *reinterpret_cast<const bf16x8*>(wA[0] + (dy + 1) * 2048);
                        Af[(dy + 1) & 1][1] = *reinterpret_cast<const bf16x8*>(wA[1] + (dy + 1) * 2048);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[dy & 1][0], Xw[(j + 1 + dy) % K][0], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[dy & 1][1], Xw[(j + 1 + dy) % K][1], acc, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                issue(a + FV_P, slot_d);
                const int orow = a - (K - 1);
                const bool ovalid = orow >= 0 && orow < L;
                int sc_i = slot_i - 1; if (sc_i < 0) sc_i += FV_R;                  // the slot of halo row a - 1: free until the DMA of row a + FV_R - 1
                unsigned char* sc = ring + sc_i * 2048;
                uint2 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { o[q].x = pack_bf16x2(acc[4 * q], acc[4 * q + 1]); o[q].y = pack_bf16x2(acc[4 * q + 2], acc[4 * q + 3]); }
                const int f = (r >> 1) & 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(sc + r * 64 + ((q ^ f) << 4) + hh * 8) = o[q];
                wave_lds_fence();
                u32x4 pend[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) pend[u] = *reinterpret_cast<const u32x4*>(sc + (16 * u + p16) * 64 + ((cch ^ ((p16 >> 1) & 3)) << 4));
                wave_lds_fence();
                const uint32_t oro = (uint32_t)(r0 + orow) * rowb;
#pragma unroll
                for (int u = 0; u < 2; ++u) __builtin_amdgcn_raw_buffer_store_b128(pend[u], ws, (ovalid && so[u] != OOB_OFF) ? so[u] + oro : OOB_OFF, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                slot_i = slot_i + 1 == FV_R ? 0 : slot_i + 1;
                slot_d = slot_d + 1 == FV_R ? 0 : slot_d + 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur += L;
    }
}
/* plain 32-channel K x 1 (K = 13, 11, 9): true when the row-stream kernel was launched */
static bool conv32_fwdk1_stream_launch(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int K, bool force, hipStream_t st) {
    const int strips = (W + 31) / 32;
    const int nw = 4;
    const int sg = (strips + nw - 1) / nw;
    int rpi = 512 / (N * sg);
    if (rpi > H / (2 * FS_MIN_RUN)) rpi = H / (2 * FS_MIN_RUN);         // a run re-reads K - 1 halo rows: at least 48 rows long
    if (rpi < 1) rpi = 1;
    const int run = (H + rpi - 1) / rpi;
    rpi = (H + run - 1) / run;
    const int blocks = N * rpi * sg;
    if (!force && (H < 2 * FS_MIN_RUN || blocks < 384)) return false;
#define FV_LAUNCH(KK)                                                                                                                                          \
    do {                                                                                                                                                      \
        static bool attr = false;                                                                                                                             \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_fwdk1_stream<KK>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_conv32_fwdk1_stream<KK>), dim3((unsigned)blocks), dim3(64 * nw), (size_t)KK * 2048 + (size_t)nw * FV_R * 2048 + 128, st, (const bf16*)x, \
                           (const bf16*)wp, bias, (bf16*)y, N, H, W, strips, run, rpi);                                                                     \
    } while (0)
    if (K == 13) FV_LAUNCH(13);
    else if (K == 11) FV_LAUNCH(11);
    else if (K == 9) FV_LAUNCH(9);
    else return false;
#undef FV_LAUNCH
    return true;
}

// ------------------------------------------------------------------------------------------------ 1 x K / K x 1 weight gradient, shifted lines (round 4)
// The generic kernel reads one x fragment per TAP and 16-pixel chunk: 13 taps = 1088 transposing reads per 8 x 64 tile and block, and the time of
// the 13-tap calls scales with exactly that (0.36 ms at level 0 against 0.22 for nine taps): the LDS array, not HBM, is the limit.  Along a line
// (a tile row for 1 x K, a tile column for K x 1; the staging puts either into consecutive LDS pixels) the x operand of tap t is the operand of
// tap 0 moved by t pixels, so all K operands of a chunk are windows of the SAME 32 pixels: two fragments (the second is the next chunk's first).
// A lane holds 8 consecutive pixels of one channel; its window of tap t is pixels [t, t + 8) of three octets -- its own, the one behind it (the other
// lane half's: one v_permlane32_swap + select per register) and the next fragment's -- cut out with v_alignbit_b32 (odd t) or by register choice
// (even t).  Per line: 5 x fragments + 4 dy fragments instead of 4 x (K + 1).  Taps are split over two wave groups (7 + 6 accumulators for K = 13),
// lines over the two waves of a group: 36 fragments = 72 transposing reads per wave and tile (288 per block instead of 1088), the matrix pipes see the same MFMAs.
// Results are bit-compatible with the generic kernel up to the order of the fp32 sums.
template <int S>
__device__ __forceinline__ bf16x8 line_window(const u32x4& o0, const u32x4& o1, const u32x4& o2) {
    static_assert(S >= 0 && S <= 15, "window offset");
    const uint32_t E[12] = {o0[0], o0[1], o0[2], o0[3], o1[0], o1[1], o1[2], o1[3], o2[0], o2[1], o2[2], o2[3]};
    constexpr int b = S / 2;
    u32x4 rr;
#pragma unroll
    for (int i = 0; i < 4; ++i) rr[i] = (S & 1) ? __builtin_amdgcn_alignbit(E[b + i + 1], E[b + i], 16) : E[b + i];
    return __builtin_bit_cast(bf16x8, rr);
}
__device__ __forceinline__ u32x4 tr_load8u(const unsigned char* p) { return __builtin_bit_cast(u32x4, tr_load8p(p)); }

template <int K, int T0, int NT>        // taps T0 .. T0 + NT - 1 of lines 4 lg .. 4 lg + 3
__device__ __forceinline__ void line_phase(f32x16 (&acc)[(K + 1) / 2], float& bsum, const unsigned char* xl0, const unsigned char* dl0, int hh) {
    constexpr int LL = 64 + K - 1;
    u32x4 X0 = tr_load8u(xl0), X1 = tr_load8u(xl0 + 16 * 64), D = tr_load8u(dl0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const int li = s >> 2, c = s & 3;
        const unsigned char* xl = xl0 + li * LL * 64;
        const unsigned char* dl = dl0 + li * 64 * 64;
        u32x4 Xa, Xb, Dn;               // next step's fragments, requested before this step's MFMAs
        if (c < 3) { Xa = tr_load8u(xl + (16 * c + 32) * 64); Dn = tr_load8u(dl + (16 * c + 16) * 64); }
        else if (li < 3) { Xa = tr_load8u(xl + LL * 64); Xb = tr_load8u(xl + LL * 64 + 16 * 64); Dn = tr_load8u(dl + 64 * 64); }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 o1;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t a_ = X0[d], b_ = X1[d];
            const auto sw = __builtin_amdgcn_permlane32_swap(a_, b_, false, false);
            const uint32_t s0 = sw[0], s1 = sw[1];          // (scalars first: hipcc 7.2 turns an indexed element of the pair inside a cast into element 0)
            o1[d] = hh ? s0 : s1;
        }
        const bf16x8 Df = __builtin_bit_cast(bf16x8, D);
        if (T0 == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bsum = dot2_ones(Df, j, bsum);
        }
        // (a lambda with a template parameter would do; spelled out so that every window offset is a literal)
#define LINE_MMA(T) if (T < NT) acc[T < NT ? T : 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Df, line_window<(T < NT ? T0 + T : 0)>(X0, o1, X1), acc[T < NT ? T : 0], 0, 0, 0)
        LINE_MMA(0); LINE_MMA(1); LINE_MMA(2); LINE_MMA(3); LINE_MMA(4); LINE_MMA(5); LINE_MMA(6);
#undef LINE_MMA
        __builtin_amdgcn_sched_barrier(0);
        if (c < 3) { X0 = X1; X1 = Xa; D = Dn; }
        else if (li < 3) { X0 = Xa; X1 = Xb; D = Dn; }
    }
}

#define WL_XL 10                // x staging slots per thread: 8 lines x (64 + K - 1 <= 76) pixels x 4 quarters / 256
template <int K, bool VERT>
__global__ void __launch_bounds__(MB, 2)
k_conv32_wgrad_line(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
                    int N, int H, int W, int tilesH, int tilesW, int ntiles) {
    constexpr int TPW = (K + 1) / 2, LL = 64 + K - 1, P = K / 2;
    constexpr int TH = VERT ? 64 : 8, TW = VERT ? 8 : 64;
    constexpr int LH = VERT ? LL : 8, LW = VERT ? 8 : LL, NPX = 8 * LL;
    static_assert(TPW <= 7 && NPX * 4 <= WL_XL * MB, "line kernel: K <= 13");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;
    unsigned char* sD = smem + (NPX + 16) * 64;          // the last line's fifth fragment reads up to 80 - LL pixels past its end (never multiplied: no window reaches them)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int tg = wave >> 1, lg = wave & 1;
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float bsum = 0.f;
    const unsigned char* xl0 = tr_lane_base(sX, lane) + lg * 4 * LL * 64;
    const unsigned char* dl0 = tr_lane_base(sD, lane) + lg * 4 * 64 * 64;
    const int c = tid & 3;
    // halo pixel of staging slot j in GLOBAL order (consecutive slots = consecutive pixels of an image row): (lr, lc); its LDS pixel = line * LL + position
    auto slot_rc = [&](int j, uint32_t& lr, uint32_t& lc) {
        uint32_t pl = (uint32_t)(tid >> 2) + (uint32_t)(j * (MB / 4));
        asm volatile("" : "+v"(pl));
        lr = pl / (uint32_t)LW;
        lc = pl - lr * (uint32_t)LW;
        return pl < (uint32_t)NPX;
    };
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    u32x4 prex[WL_XL], pred[8];
    auto prefetch = [&](int tile) {
        const int tw = tile & 1023, th = (tile >> 10) & 1023, n = tile >> 20;
        const int h0 = th * TH, w0 = tw * TW;
        const int hb = h0 - (VERT ? P : 0), wb = w0 - (VERT ? 0 : P);
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        if (hb >= 0 && wb >= 0 && hb + LH <= H && wb + LW <= W) {       // interior tile
            const uint32_t xbase = (uint32_t)((hb * W + wb) * 64 + c * 16);
#pragma unroll
            for (int j = 0; j < WL_XL; ++j) {
                uint32_t lr, lc;
                const bool in = slot_rc(j, lr, lc);
                prex[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, in ? xbase + (lr * (uint32_t)W + lc) * 64u : OOB_OFF, 0, 0);
            }
            const uint32_t dbase = (uint32_t)((h0 * W + w0) * 64 + c * 16);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t pl = (uint32_t)((tid >> 2) + j * (MB / 4));
                pred[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, dbase + ((pl / TW) * (uint32_t)W + (pl & (TW - 1))) * 64u, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < WL_XL; ++j) {
                uint32_t lr, lc;
                const bool in = slot_rc(j, lr, lc);
                const int hi = hb + (int)lr, wi_ = wb + (int)lc;
                const bool ok = in && (unsigned)hi < (unsigned)H && (unsigned)wi_ < (unsigned)W;
                prex[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? (uint32_t)((hi * W + wi_) * 64 + c * 16) : OOB_OFF, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int pl = (tid >> 2) + j * (MB / 4);
                const int ho = h0 + pl / TW, wo = w0 + (pl & (TW - 1));
                pred[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, (ho < H && wo < W) ? (uint32_t)((ho * W + wo) * 64 + c * 16) : OOB_OFF, 0, 0);
            }
        }
    };
    const TileSeq<VERT> seq(ntiles, tilesH, tilesW);
    int tile = seq.at(0), tile1 = -1;
    if (tile >= 0) prefetch(tile);
    for (int kt = 0; tile >= 0; tile = tile1, ++kt) {
        tile1 = seq.at(kt + 1);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < WL_XL; ++j) {
            uint32_t lr, lc;
            if (slot_rc(j, lr, lc)) *reinterpret_cast<u32x4*>(sX + (VERT ? lc * LL + lr : lr * LL + lc) * 64 + c * 16) = prex[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pl = (tid >> 2) + j * (MB / 4);
            const int p = VERT ? (pl & (TW - 1)) * TH + pl / TW : pl;
            *reinterpret_cast<u32x4*>(sD + p * 64 + c * 16) = pred[j];
        }
        __syncthreads();
        if (tile1 >= 0) prefetch(tile1);
        if (tg == 0) line_phase<K, 0, TPW>(acc, bsum, xl0, dl0, hh);
        else line_phase<K, TPW, K - TPW>(acc, bsum, xl0, dl0, hh);
    }
    // the two line-group waves of a tap group hold partial sums of the same taps: they take turns in LDS (as in k_conv32_wgrad)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
    for (int turn = 0; turn < 2; ++turn) {
        if (lg == turn) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int tap = tg * TPW + t;
                if (tap < K) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int co = (k & 3) + 8 * (k >> 2) + 4 * hh;
                        float* dst = &red[tap * 1024 + co * 32 + r];
                        *dst = turn == 0 ? acc[t][k] : *dst + acc[t][k];
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < K * 1024; i += MB) {
        const int tap = i % K, cc = i / K;          // cc = co*32 + ci: OIHW-linear order
        atomicAdd(&dw[(int64_t)cc * K + tap], red[tap * 1024 + cc]);
    }
    if (dbias && tg == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 32) atomicAdd(&dbias[r], bsum);
    }
}

// ------------------------------------------------------------------------------------------------ 1 x K / K x 1 weight gradient, wave-private row streams, ONE wave per SIMD (round 5)
// The row-stream structure of k_conv32_wgrad33_stream for the cross convolutions (reference nets/tcct.py:814-816).  All K accumulators (K x 16 registers: 208 for
// K = 13) stay in ONE wave, so the kernel runs one wave per SIMD (launch bounds (256, 1): the 512-register budget) -- the form DESIGN 3b (round 4) left untried
// after the wave-pair split (7 + 6 taps on two waves that stream the same strip twice) had lost to the shifted lines.  A wave owns a 16-pixel strip.
//   1 x K: ring row = 16 + K - 1 x pixels + 16 dy pixels (three DMA pieces); the x operand of tap t is ONE transposing read of the row at pixel offset t
//          (the row sits in LDS once; the shifted-line kernel's register surgery is not needed when nothing but this wave's reads competes for its LDS port);
//   K x 1: ring row = 16 x pixels + 16 dy pixels (two DMA pieces); x row b meets the dy fragments of rows b - t + (K - 1) / 2, kept in a register window
//          of the last K dy rows (K x 4 registers), body unrolled K-fold so that window positions are compile-time; ring slot by run-time index.
// WK_R = 8 ring rows per wave, WK_P = 6 of them in flight (one wave per SIMD has nobody to hide its latency behind: the depth does; 12 / 18 rows measured no faster,
// DESIGN 3b, and 150 KB of LDS per block would keep the other streams' kernels off the CU).  The counted `s_waitcnt vmcnt(NPIECE * (WK_P - 1))` and the slot
// arithmetic rest on these two constants; the launch sizes the ring from WK_R.  Same products and fp32 accumulation as the other weight-gradient kernels, summed in
// another order: bit-compatible up to that (tests/test_kernels_gpu.py).
#define WK_T 256
constexpr int WK_R = 8, WK_P = 6;           // ring rows per wave / rows in flight (64 / 88 KB of LDS per block)
static_assert(WK_P <= WK_R - 2, "k_conv32_wgradk_stream: two ring slots stay free (the row being read and the one being issued into)");
template <int K, bool VERT>
__global__ void __launch_bounds__(WK_T, 1)
k_conv32_wgradk_stream(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw, float* __restrict__ dbias,
                       int N, int H, int W, int strips, int run) {
    constexpr int PAD = (K - 1) / 2;
    constexpr int XP = VERT ? 16 : 16 + K - 1;                  // x pixels per ring row
    constexpr int ROWB = (XP + 16) * 64;
    constexpr int NPIECE = VERT ? 2 : 3;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    unsigned char* ring = smem + wave * (WK_R * ROWB);
    const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) unsigned char*)ring);
    const unsigned char* xb = tr_lane_base(ring, lane);
    f32x16 acc[K];
#pragma unroll
    for (int t = 0; t < K; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    float bsum = 0.f;
    const int sgroups = (strips + 3) >> 2;
    const int64_t total = (int64_t)N * sgroups * H;
    int64_t cur = (int64_t)blockIdx.x * run;
    const int64_t end = cur + run < total ? cur + run : total;
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    const uint32_t rowb = (uint32_t)W * 64u;
    const int px = lane >> 2, c = lane & 3;
    while (cur < end) {
        const int sidx = (int)(cur / H), r0 = (int)(cur - (int64_t)sidx * H);
        const int n = sidx / sgroups, s = (sidx - n * sgroups) * 4 + wave;
        const int left = (int)(end - cur);
        const int L = __builtin_amdgcn_readfirstlane(H - r0 < left ? H - r0 : left);      // dy rows r0 .. r0 + L - 1
        if (s >= strips) { cur += L; continue; }
        const int w0 = s * 16;
        const u32x4 rx = make_rsrc_words(x + (int64_t)n * H * W * 32, img_bytes);
        const u32x4 rd = make_rsrc_words(dy + (int64_t)n * H * W * 32, img_bytes);
        const int cd = w0 + px;
        const uint32_t od = (cd < W) ? (uint32_t)(cd * 64 + c * 16) : OOB_OFF;
        if constexpr (!VERT) {
            // ring x pixel i <-> image column w0 - PAD + i; tap t pairs dy pixel j with x pixel j + t
            const int cx1 = w0 - PAD + px, cx2 = cx1 + 16;
            const uint32_t ox1 = (cx1 >= 0 && cx1 < W) ? (uint32_t)(cx1 * 64 + c * 16) : OOB_OFF;
            const uint32_t ox2 = (px < XP - 16 && cx2 >= 0 && cx2 < W) ? (uint32_t)(cx2 * 64 + c * 16) : OOB_OFF;
            const uint32_t row0 = (uint32_t)r0 * rowb;
            auto issue = [&](int a, int slot) {
                const uint32_t ro = row0 + (uint32_t)a * rowb;
                const bool in = a < L;
                const uint32_t base = ring_lds + (uint32_t)(slot * ROWB);
                lds_dma16(rx, in ? ox1 + ro : OOB_OFF, base);
                if (lane < 4 * (XP - 16)) lds_dma16(rx, in ? ox2 + ro : OOB_OFF, base + 1024u);
                lds_dma16(rd, in ? od + ro : OOB_OFF, base + (uint32_t)(XP * 64));
            };
#pragma unroll
            for (int a = 0; a < WK_P; ++a) issue(a, a);
            const int groups = (L + WK_R - 1) / WK_R;
            for (int g = 0; g < groups; ++g) {
#pragma unroll
                for (int j = 0; j < WK_R; ++j) {
                    const int a = g * WK_R + j;
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPIECE * (WK_P - 1)) : "memory");         // row a has landed
                    const unsigned char* row = xb + j * ROWB;
                    const bf16x8 D = tr_load8p(row + XP * 64);
                    bf16x8 X0 = tr_load8p(row), X1 = tr_load8p(row + 64);
#pragma unroll
                    for (int q = 0; q < 4; ++q) bsum = dot2_ones(D, q, bsum);
#pragma unroll
                    for (int t = 0; t < K; ++t) {
                        bf16x8 X2 = X1;
                        if (t + 2 < K) X2 = tr_load8p(row + (t + 2) * 64);          // two operands ahead of the matrix pipe
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(D, X0, acc[t], 0, 0, 0);
                        X0 = X1; X1 = X2;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    issue(a + WK_P, (j + WK_P) % WK_R);                 // rows a + 1 .. a + WK_P - 1 are in flight in the slots behind this one; two slots stay free
                }
            }
        } else {
            // x halo row a of the segment = image row r0 - PAD + a (a = 0 .. L + K - 2); dy row a = image row r0 + a rides in the same ring row
            const int cx = w0 + px;
            const uint32_t ox = (cx < W) ? (uint32_t)(cx * 64 + c * 16) : OOB_OFF;
            const int nh = L + K - 1;
            int slot_i = 0, slot_d = WK_P % WK_R;
            auto issue = [&](int a, int slot) {
                const int xr = r0 - PAD + a;
                const bool xin = a < nh && xr >= 0 && xr < H, din = a < L;
                const uint32_t base = ring_lds + (uint32_t)(slot * ROWB);
                lds_dma16(rx, xin ? ox + (uint32_t)xr * rowb : OOB_OFF, base);
                lds_dma16(rd, din ? od + (uint32_t)(r0 + a) * rowb : OOB_OFF, base + 1024u);
            };
#pragma unroll
            for (int a = 0; a < WK_P; ++a) issue(a, a);
            bf16x8 Dw[K];
#pragma unroll
            for (int t = 0; t < K; ++t)
#pragma unroll
                for (int k = 0; k < 8; ++k) Dw[t][k] = (__bf16)0.f;
            const int groups = (nh + K - 1) / K;
            for (int g = 0; g < groups; ++g) {
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const int a = g * K + j;
                    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NPIECE * (WK_P - 1)) : "memory");
                    const unsigned char* row = xb + slot_i * ROWB;
                    const bf16x8 X = tr_load8p(row);
                    Dw[j] = tr_load8p(row + 1024);                      // dy row a (zeros behind the segment)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bsum = dot2_ones(Dw[j], q, bsum);
#pragma unroll
                    for (int t = 0; t < K; ++t)                         // x row a is tap row t of dy row a - t: window position (j - t) mod K
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Dw[(j + K - t) % K], X, acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    issue(a + WK_P, slot_d);
                    slot_i = slot_i + 1 == WK_R ? 0 : slot_i + 1;
                    slot_d = slot_d + 1 == WK_R ? 0 : slot_d + 1;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur += L;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);                    // [K][32 co][32 ci]
    for (int turn = 0; turn < WK_T / 64; ++turn) {
        if (wave == turn) {
#pragma unroll
            for (int t = 0; t < K; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int co = (k & 3) + 8 * (k >> 2) + 4 * hh;
                    float* dst = &red[t * 1024 + co * 32 + r];
                    *dst = turn == 0 ? acc[t][k] : *dst + acc[t][k];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < K * 1024; i += WK_T) {
        const int tap = i % K, cc = i / K;
        atomicAdd(&dw[(int64_t)cc * K + tap], red[tap * 1024 + cc]);
    }
    __syncthreads();
    if (dbias) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 32) red[wave * 32 + r] = bsum;
        __syncthreads();
        if (tid < 32) atomicAdd(&dbias[tid], red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid]);
    }
}

static int g_wgrad_mode = -1;
/* 0 (default): the fastest kernel per shape (3x3: row streams at levels 0-1, rolling rows below; 1 x K / K x 1 with 13 / 11 taps: one-wave-per-SIMD row streams
 * at levels 0-1, shifted lines below); 1: the generic register-staged kernel for every shape (the comparison arm of the bit-compatibility tests); 2: the 3x3 row
 * streams at every size; 3: the 1 x K / K x 1 row streams at every size; 4: the shifted-line kernel instead of them (A/B).  Returns the previous mode; mode < 0
 * only queries. */
extern "C" int64_t tcct_conv32_wgrad_mode(int mode) {
    if (g_wgrad_mode < 0) g_wgrad_mode = 0;
    const int prev = g_wgrad_mode;
    if (mode >= 0 && mode <= 4) g_wgrad_mode = mode;
    return prev;
}
/* dw OIHW fp32 [32,32,KH,KW] and dbias fp32 [32] (nullable) are overwritten. */
static int conv32_wgrad_impl(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH,
                             int PW, int xs, int xo, int ds, int dof, int ldi, int o_off, int i_off, int zero, tcct_stream_t stream);
extern "C" int tcct_conv32_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW,
                                 int PH, int PW, tcct_stream_t stream) {
    return conv32_wgrad_impl(x, dy, dw, dbias, N, H, W, KH, KW, PH, PW, 32, 0, 32, 0, 32, 0, 0, 1, stream);
}
/* 32x32 sub-block (o_off.., i_off..) of the gradient of an OIHW weight with cin_total input channels, from the x slab (xs, xo)
 * and the dy slab (ds, dof).  This entry ACCUMULATES with atomics and clears nothing: zero dw / dbias once before the sub-calls. */
extern "C" int tcct_conv32_wgrad_strided(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW,
                                         int PH, int PW, int xs, int xo, int ds, int dof, int cin_total, int o_off, int i_off,
                                         tcct_stream_t stream) {
    TCCT_CHECK(xs % 8 == 0 && xo % 8 == 0 && ds % 8 == 0 && dof % 8 == 0 && xo + 32 <= xs && dof + 32 <= ds, "conv32_wgrad_strided: bad slab");
    return conv32_wgrad_impl(x, dy, dw, dbias, N, H, W, KH, KW, PH, PW, xs, xo, ds, dof, cin_total, o_off, i_off, 0, stream);
}
static int conv32_wgrad_impl(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH,
                             int PW, int xs, int xo, int ds, int dof, int ldi, int o_off, int i_off, int zero, tcct_stream_t stream) {
    TCCT_CHECK(2 * PH == KH - 1 && 2 * PW == KW - 1, "conv32_wgrad: only 'same' padding");
    const int TAPS = KH * KW;
    TCCT_CHECK(TAPS >= 1 && TAPS <= 16, "conv32_wgrad: %dx%d unsupported (<= 16 taps)", KH, KW);
    const bool vert = (KW == 1 && KH > 1);
    const bool sq = !vert && KH == 3 && KW == 3 && xs == 32 && ds == 32;      // plain 3x3: 16 x 32 tiles
    const int TH = vert ? 64 : (sq ? 16 : 8), TW = vert ? 8 : (sq ? 32 : 64);
    const int LH = TH + KH - 1, LW = TW + KW - 1;
    size_t lds = (size_t)LH * LW * 64 + (size_t)TH * TW * 64;
    size_t red = (size_t)TAPS * 4096;
    if (red > lds) lds = red;
    TCCT_CHECK(lds <= 80 * 1024, "conv32_wgrad: %dx%d needs %zu B of LDS", KH, KW, lds);
    int tilesH = (H + TH - 1) / TH, tilesW = (W + TW - 1) / TW;
    int64_t nt = (int64_t)N * tilesH * tilesW;
    TCCT_CHECK(nt > 0 && nt < (1LL << 31), "conv32_wgrad: bad tile count");
    TCCT_CHECK(tilesH < 1024 && tilesW < 1024 && N < 2048, "conv32_wgrad: %d images of %d x %d tiles exceed the packed tile id (2047 images, 1023 x 1023 tiles)", N, tilesH, tilesW);
    TCCT_CHECK((int64_t)H * W * xs * 2 < (1LL << 31) && (int64_t)H * W * ds * 2 < (1LL << 31),
               "conv32_wgrad: one image of %d x %d x %d channels exceeds the 2 GiB buffer-descriptor range", H, W, xs > ds ? xs : ds);
    // every block of these kernels ends with TAPS x 1024 fp32 atomics on the same addresses: on small maps (levels 2-4: <= 1024 tiles) <= 128 blocks walk several tiles each
    // instead of 512 blocks spending more time in that tail than in their one or two tiles (round 6; the large maps keep 512)
    int grid = (int)(nt <= 1024 ? (nt < 128 ? nt : 128) : 512);
    hipStream_t st = (hipStream_t)stream;
    if (zero) {
        if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * TAPS * 1024, st) != hipSuccess) { tcct_set_error("conv32_wgrad: memset failed"); return -2; }
        if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * 32, st) != hipSuccess) { tcct_set_error("conv32_wgrad: memset failed"); return -2; }
    }
    if (g_wgrad_mode < 0) (void)tcct_conv32_wgrad_mode(-1);
    constexpr int stream_on = 1;
    const int m33 = g_wgrad_mode >= 3 ? 0 : g_wgrad_mode;           // modes 3 / 4 only concern the cross convolutions
    if ((m33 == 2 || (m33 == 0 && stream_on)) && sq && ldi == 32 && o_off == 0 && i_off == 0 && xo == 0 && dof == 0) {
        const int strips = (W + 15) / 16;
        const int64_t rows = (int64_t)N * ((strips + 3) / 4) * H;            // rows of strip groups (four adjacent strips, one per wave of a block)
        int blocks = 512;
        int64_t run = (rows + blocks - 1) / blocks;
        if (m33 == 2 || run >= WS_MIN_RUN) {
            if (run < 12) run = 12;                 // small maps: fewer, longer runs (the pipeline fill is 7 rows)
            blocks = (int)((rows + run - 1) / run);
            constexpr size_t ldss = (size_t)(WS_T / 64) * WS_R * WS_ROWB;
            static bool attrs = false;
            if (!attrs) { (void)hipFuncSetAttribute((const void*)k_conv32_wgrad33_stream, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attrs = true; }
            hipLaunchKernelGGL(k_conv32_wgrad33_stream, dim3((unsigned)blocks), dim3(WS_T), ldss, st, (const bf16*)x, (const bf16*)dy, dw, dbias, N, H, W, strips, (int)run);
            tcct_census_hit(TCCT_CENSUS_WGRAD33_STREAM);
            TCCT_LAUNCH_OK();
        }
    }
    if ((m33 == 0 || m33 == 2) && sq && ldi == 32 && o_off == 0 && i_off == 0 && xo == 0 && dof == 0) {      // plain 3x3: rolling rows, 6 waves x 2 blocks per CU
        constexpr size_t lds4 = (size_t)18 * 34 * 64 + 16 * 32 * 64;
        static bool attr4 = false;
        if (!attr4) { (void)hipFuncSetAttribute((const void*)k_conv32_wgrad33_roll, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr4 = true; }
        // (`grid`: <= 128 blocks on the small maps this kernel serves -- 96-936 tiles at levels 2-4 --: one block per tile spent more time in the 9 216 closing atomics than in
        // its tile, 4.7 M atomics at level 2)
        hipLaunchKernelGGL(k_conv32_wgrad33_roll, dim3((unsigned)grid), dim3(WR_T), lds4, st, (const bf16*)x, (const bf16*)dy, dw, dbias, N, H, W,
                           tilesH, tilesW, (int)nt);
        TCCT_LAUNCH_OK();
    }
    // 1 x K / K x 1 with K = 13, 11 on maps whose waves get long runs (levels 0-1): all taps in one wave per SIMD (mode 3 forces it, mode 4 keeps the shifted lines)
    if ((g_wgrad_mode == 0 || g_wgrad_mode == 3) && (KH == 1 || KW == 1) && (TAPS == 13 || TAPS == 11) && xs == 32 && ds == 32 && ldi == 32 && o_off == 0 && i_off == 0 &&
        xo == 0 && dof == 0) {
        const int strips = (W + 15) / 16;
        const int64_t rows = (int64_t)N * ((strips + 3) / 4) * H;
        int blocks = 256;                           // one block (four waves, one per SIMD) per CU
        int64_t run = (rows + blocks - 1) / blocks;
        if (g_wgrad_mode == 3 || run >= 96) {
            if (run < 16) run = 16;
            blocks = (int)((rows + run - 1) / run);
#define WK_LAUNCH(KK, V)                                                                                                     \
    do {                                                                                                                    \
        constexpr size_t ring = (size_t)(WK_T / 64) * WK_R * ((V ? 16 : 16 + KK - 1) + 16) * 64, redb = (size_t)KK * 4096;                                     \
        constexpr size_t ldsk = ring > redb ? ring : redb;                                                                                                     \
        static bool attr = false;                                                                                           \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_wgradk_stream<KK, V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_conv32_wgradk_stream<KK, V>), dim3((unsigned)blocks), dim3(WK_T), ldsk, st, (const bf16*)x, (const bf16*)dy, dw, dbias, N, H, W, strips, (int)run); \
    } while (0)
            if (TAPS == 13) { if (vert) WK_LAUNCH(13, true); else WK_LAUNCH(13, false); }
            else { if (vert) WK_LAUNCH(11, true); else WK_LAUNCH(11, false); }
#undef WK_LAUNCH
            tcct_census_hit(TCCT_CENSUS_WGRADK_STREAM);
            TCCT_LAUNCH_OK();
        }
    }
    constexpr bool line_on = true;
    if (line_on && g_wgrad_mode != 1 && (KH == 1 || KW == 1) && (TAPS == 13 || TAPS == 11 || TAPS == 9) && xs == 32 && ds == 32 && ldi == 32 && o_off == 0 && i_off == 0 &&
        xo == 0 && dof == 0) {      // 1 x K / K x 1 at levels 0-2: shifted lines
#define WL_LAUNCH(KK, V)                                                                                                     \
    do {                                                                                                                    \
        static bool attr = false;                                                                                           \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_wgrad_line<KK, V>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_conv32_wgrad_line<KK, V>), dim3(grid), dim3(MB), (size_t)(8 * (64 + KK - 1) + 16) * 64 + 512 * 64, st, (const bf16*)x, (const bf16*)dy, \
                           dw, dbias, N, H, W, tilesH, tilesW, (int)nt);                                                    \
    } while (0)
        if (TAPS == 13) { if (vert) WL_LAUNCH(13, true); else WL_LAUNCH(13, false); }
        else if (TAPS == 11) { if (vert) WL_LAUNCH(11, true); else WL_LAUNCH(11, false); }
        else { if (vert) WL_LAUNCH(9, true); else WL_LAUNCH(9, false); }
#undef WL_LAUNCH
        TCCT_LAUNCH_OK();
    }
    // taps are split over TG wave groups so that <= 5 accumulators (80 VGPRs) live next to the prefetch registers: no spills
    const int TG = TAPS > 10 ? 4 : (TAPS > 5 ? 2 : 1);
    const int tpw = (TAPS + TG - 1) / TG;
    TCCT_CHECK(TAPS <= 16, "conv32_wgrad: %d taps unsupported", TAPS);
    TCCT_CHECK(LH * LW * 4 <= MAXL * MB, "conv32_wgrad: %dx%d tile image exceeds the staging slots", KH, KW);
#define WG_LAUNCH(TPW, V, Q)                                                                                                 \
    do {                                                                                                                    \
        static bool attr = false;                                                                                           \
        if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_wgrad<TPW, V, Q>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } \
        hipLaunchKernelGGL((k_conv32_wgrad<TPW, V, Q>), dim3(grid), dim3(MB), lds, st, (const bf16*)x, (const bf16*)dy, dw, dbias, N, H, W, KH, \
                           KW, PH, PW, TG, tilesH, tilesW, (int)nt, xs, xo, ds, dof, ldi, o_off, i_off);                                                        \
    } while (0)
    if (sq) WG_LAUNCH(5, false, true);
    else if (tpw <= 4) { if (vert) WG_LAUNCH(4, true, false); else WG_LAUNCH(4, false, false); }
    else { if (vert) WG_LAUNCH(5, true, false); else WG_LAUNCH(5, false, false); }
#undef WG_LAUNCH
    TCCT_LAUNCH_OK();
}

// ------------------------------------------------------------------------------------------------ fused backward (3x3)
// Input gradient AND weight gradient of a dense 3x3 32->32 convolution in ONE pass: the two separate kernels each stage a dy tile
// (k_conv32_mfma on the flipped weights: dy with halo; k_conv32_wgrad: dy interior + x with halo), i.e. dy crosses HBM twice.
// Here a block of EIGHT waves stages the x halo tile (linear 64-byte pixel rows, transposing reads) and the dy halo tile (80-byte
// rows: conflict-free b128 fragment reads for the input gradient, transposing reads for the weight gradient) once per 16 x 32 tile;
// waves 0-3 run the input-gradient MFMA phase + epilogue of k_conv32_mfma, waves 4-7 the weight-gradient phase of k_conv32_wgrad
// (taps 0-4 / 5-8 on two wave pairs, 16 of the 32 pixel chunks each).  One block per CU (111 KB of LDS), the two roles share each
// SIMD's matrix pipe (72 + 80 MFMAs per tile and SIMD).  Algorithmic bytes per image pixel: x + dy + dx = 3 x 64 B (4 x 64 B before).
#define BWD_T 512
template <int DUMMY>
__global__ void __launch_bounds__(BWD_T, 1)
k_conv32_bwd33(const bf16* __restrict__ x, const bf16* __restrict__ dy, const bf16* __restrict__ wp, const bf16* __restrict__ dskip,
               bf16* __restrict__ dx, float* __restrict__ dw, float* __restrict__ dbias, int N, int H, int W, int tilesH, int tilesW,
               int ntiles) {
    constexpr int TH = 16, TW = 32, LH = 18, LW = 34, NPIX = LH * LW, SLOTS = (NPIX * 4 + BWD_T - 1) / BWD_T;      // 5 slots per thread and image
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                              // input-gradient weights [9][32][32] bf16, swizzled rows (k_conv32_mfma)
    unsigned char* sXl = smem + 9 * 2048;                  // x halo tile, 64 B per pixel
    unsigned char* sDp = sXl + NPIX * 64;                  // dy halo tile, IPS = 80 B per pixel
    unsigned char* sS = sDp + NPIX * IPS;                  // epilogue transpose scratch: 4 waves x 1 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    for (int i = tid; i < 9 * 32 * 4; i += BWD_T) {
        int row = i >> 2, c4 = i & 3;
        uint4 v = reinterpret_cast<const uint4*>(wp)[i];
        *reinterpret_cast<uint4*>(sW + row * 64 + ((c4 ^ ((row >> 2) & 3)) << 4)) = v;
    }
    const int c = tid & 3;
    int s_rc[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        int pl = (tid >> 2) + j * (BWD_T / 4);
        bool in = pl < NPIX;
        int lr = in ? pl / LW : 0x3fff, lc = in ? pl - (pl / LW) * LW : 0;
        s_rc[j] = (lr << 16) | lc;
    }
    const uint32_t img_bytes = (uint32_t)H * (uint32_t)W * 64u;
    u32x4 px[SLOTS], pd[SLOTS];
    auto prefetch = [&](int tile) {                        // x and dy have the same geometry: one offset serves both loads
        const int tw = tile % tilesW;
        const int t2 = tile / tilesW;
        const int th = t2 % tilesH;
        const int n = t2 / tilesH;
        const int hb = th * TH - 1, wb = tw * TW - 1;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int hi = hb + (s_rc[j] >> 16), wi = wb + (s_rc[j] & 0xffff);
            const bool ok = (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
            const uint32_t off = ok ? (uint32_t)((hi * W + wi) * 64 + c * 16) : OOB_OFF;
            px[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0);
            pd[j] = __builtin_amdgcn_raw_buffer_load_b128(rd, off, 0, 0);
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j) {
            const int lr = s_rc[j] >> 16, lc = s_rc[j] & 0xffff;
            if (lr != 0x3fff) {
                const int p = lr * LW + lc;
                *reinterpret_cast<u32x4*>(sXl + p * 64 + c * 16) = px[j];
                *reinterpret_cast<u32x4*>(sDp + p * IPS + c * 16) = pd[j];
            }
        }
    };
    // ---- role state.  Input gradient (waves 0-3): fragment bases as in k_conv32_mfma (SQ tiles: M-tile = one tile row of 32 pixels)
    const int wsw = (r >> 2) & 3;
    const unsigned char* wA0 = sW + r * 64 + ((hh ^ wsw) << 4);
    const unsigned char* wA1 = sW + r * 64 + (((2 + hh) ^ wsw) << 4);
    const int dwave = wave & 3;
    const unsigned char* xB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) xB[t] = sDp + ((dwave * 4 + t) * LW + r) * IPS + hh * 16;
    // Weight gradient (waves 4-7): tap group tg (taps 5 tg .. 5 tg + 4), chunk parity wi
    const int tg = (wave >> 1) & 1, wi = wave & 1;
    const int tap0 = tg * 5;
    int poff[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int tap = tap0 + t, ky = tap / 3, kx = tap - ky * 3;
        poff[t] = tap < 9 ? (ky * LW + kx) * 64 : 0;
    }
    // ONE set of five accumulators per wave: the weight-gradient waves keep their taps in it for the whole kernel, the input-gradient
    // waves re-use the first four as the per-tile output accumulators (a wave never changes role)
    f32x16 accw[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) accw[t][k] = 0.f;
    float bsum = 0.f;
    const int li = lane & 15, lq = li >> 2, lpp = li & 3, lg = lane >> 4;
    const unsigned char* lbX = sXl + (8 * (lg >> 1) + lq) * 64 + (16 * (lg & 1) + 4 * lpp) * 2;
    const unsigned char* lbD = sDp + (8 * (lg >> 1) + lq) * IPS + (16 * (lg & 1) + 4 * lpp) * 2;
    auto tr8 = [&](const unsigned char* p, int stride) {   // 16 consecutive pixels from p: 8 pixel values of channel lane & 31
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * stride));
        s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    };

    int tile = blockIdx.x;
    if (tile < ntiles) {
        prefetch(tile);
        __syncthreads();
        stage();
        if (tile + (int)gridDim.x < ntiles) prefetch(tile + gridDim.x);
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int tw = tile % tilesW;
        const int t2 = tile / tilesW;
        const int th = t2 % tilesH;
        const int n = t2 / tilesH;
        const int h0 = th * TH, w0 = tw * TW;
        f32x16* const acc = accw;
        if (wave < 4) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
            struct Frag { bf16x8 a, b[4]; };
            auto load_stage = [&](Frag& f, int ky, int kx, int half) {
                f.a = *reinterpret_cast<const bf16x8*>((half ? wA1 : wA0) + (ky * 3 + kx) * 2048);
                const int po = (ky * LW + kx) * IPS + half * 32;
#pragma unroll
                for (int t = 0; t < 4; ++t) f.b[t] = *reinterpret_cast<const bf16x8*>(xB[t] + po);
            };
            Frag f[2];
            load_stage(f[0], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                if (k + 1 < 18) load_stage(f[(k + 1) & 1], ((k + 1) >> 1) / 3, ((k + 1) >> 1) % 3, (k + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[k & 1].a, f[k & 1].b[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // 32 chunks of 16 pixels (tile row a = ch / 2, half s = ch % 2); this wave takes the 16 chunks of its parity.  The fragments
            // of the next chunk are read while the MFMAs of the current one run (two named buffers; sched_barrier pins the order):
            // unpipelined, every chunk exposed one LDS latency in front of its five MFMAs.
            struct WF { bf16x8 a, b[5]; };
            auto load_chunk = [&](WF& f, int ch) {
                const int a = ch >> 1, s16 = (ch & 1) * 16;
                f.a = tr8(lbD + ((a + 1) * LW + s16 + 1) * IPS, IPS);
                const unsigned char* pxb = lbX + (a * LW + s16) * 64;
#pragma unroll
                for (int t = 0; t < 5; ++t) f.b[t] = tr8(pxb + poff[t], 64);
            };
            auto mma_chunk = [&](const WF& f) {
                if (tg == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) bsum = dot2_ones(f.a, j, bsum);      // 4 x v_dot2c_f32_bf16 instead of 8 x (unpack + add): the bias sum was 12 % of the compute side
                }
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    if (tap0 + t < 9) accw[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a, f.b[t], accw[t], 0, 0, 0);
            };
            WF f0, f1;
            load_chunk(f0, wi);
            for (int ch = wi; ch < 32; ch += 4) {
                load_chunk(f1, ch + 2);
                __builtin_amdgcn_sched_barrier(0);
                mma_chunk(f0);
                __builtin_amdgcn_sched_barrier(0);
                if (ch + 4 < 32) load_chunk(f0, ch + 4);
                __builtin_amdgcn_sched_barrier(0);
                mma_chunk(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                    // both roles are done with the staged tile
        if (tile + (int)gridDim.x < ntiles) stage();
        if (tile + 2 * (int)gridDim.x < ntiles) prefetch(tile + 2 * gridDim.x);
        if (wave < 4) {                                     // input-gradient epilogue (k_conv32_mfma's, + the second consumer's gradient)
            const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc((void*)(dx + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc((void*)((dskip ? dskip : dx) + (int64_t)n * H * W * 32), 0, img_bytes, 0x00020000);
            unsigned char* sc = sS + dwave * 1024;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int a = dwave * 4 + t;
                uint2 o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    o[q].x = pack_bf16x2(acc[t][4 * q], acc[t][4 * q + 1]);
                    o[q].y = pack_bf16x2(acc[t][4 * q + 2], acc[t][4 * q + 3]);
                }
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    if ((r >> 4) == h2) {
                        const int rr = r & 15, fsw = (rr >> 1) & 3;
#pragma unroll
                        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(sc + rr * 64 + ((q ^ fsw) << 4) + hh * 8) = o[q];
                    }
                    wave_lds_fence();
                    const int p16 = lane >> 2, cch = lane & 3;
                    u32x4 ov = *reinterpret_cast<const u32x4*>(sc + p16 * 64 + ((cch ^ ((p16 >> 1) & 3)) << 4));
                    const int ho = h0 + a, wo = w0 + 16 * h2 + p16;
                    const uint32_t off = (ho < H && wo < W) ? (uint32_t)((ho * W + wo) * 64 + cch * 16) : OOB_OFF;
                    if (dskip) {                            // block-uniform
                        const u32x4 sv = __builtin_amdgcn_raw_buffer_load_b128(rsk, off, 0, 0);
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            ov[k] = pack_bf16x2(__uint_as_float(ov[k] << 16) + __uint_as_float(sv[k] << 16),
                                                __uint_as_float(ov[k] & 0xffff0000u) + __uint_as_float(sv[k] & 0xffff0000u));
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(ov, ws, off, 0, 0);
                    wave_lds_fence();
                }
            }
        }
        __syncthreads();                                    // the next tile's images are complete
    }
    // ---- weight-gradient reduction: the two waves of a tap group take turns in LDS, then OIHW-linear atomics by the whole block
    __syncthreads();
    float* red = reinterpret_cast<float*>(sXl);             // [9][32][32] floats = 36 KB over the (dead) x image
    for (int turn = 0; turn < 2; ++turn) {
        if (wave >= 4 && wi == turn) {
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const int tap = tap0 + t;
                if (tap < 9) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int co = (k & 3) + 8 * (k >> 2) + 4 * hh;
                        float* dst = &red[tap * 1024 + co * 32 + r];
                        *dst = turn == 0 ? accw[t][k] : *dst + accw[t][k];
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < 9 * 1024; i += BWD_T) {
        const int tap = i % 9, cc = i / 9;                  // cc = co*32 + ci: consecutive lanes -> consecutive addresses
        atomicAdd(&dw[(int64_t)cc * 9 + tap], red[tap * 1024 + cc]);
    }
    if (dbias && wave >= 4 && tg == 0) {
        bsum += __shfl_xor(bsum, 32, 64);
        if (lane < 32) atomicAdd(&dbias[r], bsum);
    }
}

/* Fused backward of a dense 3x3 32->32 'same' convolution (bf16 NHWC [N,H,W,32]): dx = conv(dy, flipped weights) (+ dskip, nullable:
 * the gradient that reaches x through its other consumers), dw OIHW fp32 [32,32,3,3] += dy (x) x, dbias [32] += sum dy (nullable).
 * wp_t: the input-gradient pack of the weights (second half of tcct_conv32_pack_weights_both, or pack_weights(transposed=1)).
 * dw / dbias are cleared first unless tcct_set_outputs_prezeroed(1). */
extern "C" int tcct_conv32_bwd3x3(const void* x, const void* dy, const void* wp_t, const void* dskip, void* dx, float* dw, float* dbias,
                                  int N, int H, int W, tcct_stream_t stream) {
    TCCT_CHECK((int64_t)H * W * 64 < (1LL << 31), "conv32_bwd3x3: image exceeds the 2 GiB buffer-descriptor range");
    TCCT_CHECK(dskip != dx, "conv32_bwd3x3: dskip must be a separate tensor");
    hipStream_t st = (hipStream_t)stream;
    if (!tcct_skip_zero_fill() && hipMemsetAsync(dw, 0, sizeof(float) * 9 * 1024, st) != hipSuccess) { tcct_set_error("conv32_bwd3x3: memset failed"); return -2; }
    if (dbias && !tcct_skip_zero_fill() && hipMemsetAsync(dbias, 0, sizeof(float) * 32, st) != hipSuccess) { tcct_set_error("conv32_bwd3x3: memset failed"); return -2; }
    const int tilesH = (H + 15) / 16, tilesW = (W + 31) / 32;
    const int64_t nt = (int64_t)N * tilesH * tilesW;
    TCCT_CHECK(nt > 0 && nt < (1LL << 31), "conv32_bwd3x3: bad tile count");
    const size_t lds = 9 * 2048 + (size_t)18 * 34 * 64 + (size_t)18 * 34 * IPS + 4096;
    const int grid = (int)(nt < 256 ? nt : 256);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_conv32_bwd33<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL((k_conv32_bwd33<0>), dim3(grid), dim3(BWD_T), lds, st, (const bf16*)x, (const bf16*)dy, (const bf16*)wp_t, (const bf16*)dskip,
                       (bf16*)dx, dw, dbias, N, H, W, tilesH, tilesW, (int)nt);
    TCCT_LAUNCH_OK();
}
