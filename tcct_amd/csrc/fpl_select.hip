// Feature-polarization loss, bin assignment WITHOUT a sort (round 3; replaces the rocPRIM radix_sort_pairs call of fpl_optim.hip).
//
// Reference (nets/fcs.py:25-50, nets/reg.py:86-105): per class c, the n_c pixels labelled c are sorted by their detached softmax probability,
// descending; the first 32 * N_c of them (N_c = n_c // 32, the tail is dropped) are cut into 32 consecutive bins of N_c pixels and every bin is
// averaged over its 32-channel feature rows.  Only the BIN of a pixel is needed, i.e. its rank relative to the 32 boundaries r_b = (b + 1) N_c,
// not its exact rank: a radix MULTI-SELECT finds the 32 boundary elements per class with histogram passes, most significant byte first.
//
//   key(p)  = (~bits(prob[p]) << IB) | p        unique 64-bit key; ascending key order == descending probability, ties by ascending pixel index
//             (exactly the order the former stable sort of (label, ~prob) with the pixel index as value produced)
//   level l : histogram of byte l of the key for every ACTIVE (class, prefix) slot -- level 0: one slot per class; later: the distinct prefixes
//             that still contain an unresolved boundary (<= 32 per class) -- in LDS-private counters, merged with global atomics;
//             a one-block kernel then walks every boundary one byte down (prefix scan of its slot's 256 counters) and resolves it as soon as
//             its bucket holds ONE element (always at the last byte: keys are unique), and builds the next level's slot list
//   assign  : bin(p) = #{b : key(p) >= key of boundary b} by binary search over the class's 32 (prefix, shift) pairs, one thread per pixel;
//   bin sums: the feature rows are read in PIXEL order (one coalesced pass; the sorted order forced a scattered 64-byte gather per pixel)
//             with run-length accumulation in registers and LDS-private [class][bin][32] sums.
// 4 probability bytes + ceil(log2 M / 8) index bytes = 7 levels at the bench shape; every launch has constant arguments (no host sync).
// Round 5: ONE launch per level (was two histogram passes + a one-block resolve kernel: 13 + 7 launches): the block-private counters are 16 bits wide (a block
// sees < 65536 pixels), so all 160 slots of five classes fit one pass, and the block that takes the LAST ticket of a level (a relaxed atomic counter behind
// `s_waitcnt vmcnt(0)` + the block barrier; no __threadfence on the release side) runs the resolve step itself behind ONE agent-scope acquire fence, then
// reads the merged histogram with plain loads.
#include "common.h"

#define FS_MAXC 16
#define FS_BINS 32
#define FS_SLOTS 160            // slots histogrammed per pass of a level: 160 x 256 16-bit counters (two per LDS word) = 80 KB; 5 classes x 32 bins in ONE pass
#define FS_BLOCK_PIX 60000      // pixels per block at most: a 16-bit LDS counter cannot overflow
#define FS_TB 1024
#define FS_RUN 32              // consecutive pixels per 8-lane group of the bin-sum pass

struct FplState {
    uint32_t nslots, pad0, pad1, pad2;
    uint32_t slot_base[FS_MAXC + 1];                 // class c owns slots [slot_base[c], slot_base[c+1])
    uint32_t counts[FS_MAXC], nb[FS_MAXC];           // n_c, N_c
    uint64_t slot_prefix[FS_MAXC * FS_BINS];         // ascending inside a class
    uint64_t bnd_prefix[FS_MAXC * FS_BINS];          // key >> bnd_shift >= bnd_prefix  <=>  key >= key of the boundary element
    uint32_t bnd_resid[FS_MAXC * FS_BINS];           // rank of the boundary inside its current slot
    int32_t bnd_slot[FS_MAXC * FS_BINS];
    uint32_t bnd_shift[FS_MAXC * FS_BINS];
    uint32_t bnd_state[FS_MAXC * FS_BINS];           // 0 unresolved, 1 resolved, 2 beyond the last element (+inf), 3 class has no full bin
};

__device__ __forceinline__ unsigned long long fs_key(float p, uint32_t i, int ib) {
    return ((unsigned long long)(~__float_as_uint(p)) << ib) | (unsigned long long)i;        // prob >= 0: uint order == float order
}

// one pass of level `level` over the slots [slot_lo, slot_lo + FS_SLOTS)
__device__ void fs_resolve_body(int C, int level, int nlevels, int tb, FplState* __restrict__ st, uint32_t* __restrict__ hist);

__global__ void __launch_bounds__(FS_TB) k_fs_hist(const uint8_t* __restrict__ lab, const float* __restrict__ prob, int64_t M, int C, int level,
                                                   int ib, int tb, int passes, int nlevels, FplState* __restrict__ st, uint32_t* __restrict__ hist,
                                                   uint32_t* __restrict__ tickets) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* lh = reinterpret_cast<uint32_t*>(smem);                                   // [FS_SLOTS][128]: two 16-bit counters per word
    unsigned long long* sp = reinterpret_cast<unsigned long long*>(lh + FS_SLOTS * 128);   // [nslots] prefixes
    uint32_t* sb = reinterpret_cast<uint32_t*>(sp + FS_MAXC * FS_BINS);                 // [C + 1]
    __shared__ uint32_t s_last;
    const int nslots = level == 0 ? C : (int)st->nslots;
    const int pass = blockIdx.y, slot_lo = pass * FS_SLOTS;
    const bool counting = slot_lo < nslots;            // block-uniform: a pass with nothing left to count only takes its ticket
    if (counting) {
    for (int i = threadIdx.x; i < FS_SLOTS * 128; i += FS_TB) lh[i] = 0;
    if (level > 0) {
        for (int i = threadIdx.x; i < nslots; i += FS_TB) sp[i] = st->slot_prefix[i];
        if (threadIdx.x <= C) sb[threadIdx.x] = st->slot_base[threadIdx.x];
    }
    __syncthreads();
    const int sh_pre = tb - 8 * level, sh_dig = tb - 8 * (level + 1);
    // four pixels per thread and round, their eight loads issued together: one dependent load pair per round (27 rounds per block at the bench shape) made
    // the pass latency-bound (~40 us per level for 35 MB)
    const int64_t stride = (int64_t)gridDim.x * FS_TB;
    for (int64_t base = (int64_t)blockIdx.x * FS_TB; base < M; base += 4 * stride) {
        int cq[4]; float pq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t iu = base + u * stride + threadIdx.x;
            cq[u] = iu < M ? (int)lab[iu] : 255;
            pq[u] = iu < M ? prob[iu] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
        const int64_t i = base + u * stride + threadIdx.x;
        const int c = cq[u];
        if (c >= C) continue;
        const unsigned long long key = fs_key(pq[u], (uint32_t)i, ib);
        int s;
        if (level == 0) s = c;
        else {
            const unsigned long long pf = key >> sh_pre;
            int lo = (int)sb[c], hi = (int)sb[c + 1];        // first slot with prefix >= pf
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (sp[mid] < pf) lo = mid + 1; else hi = mid; }
            s = (lo < (int)sb[c + 1] && sp[lo] == pf) ? lo : -1;
        }
        // Counter index of this lane, -1: none.  Probabilities concentrate (a freshly initialised network gives ~1/C everywhere, a trained one 1.0):
        // whole waves then hit ONE counter and same-address LDS atomics serialise lane by lane.  Up to four rounds of "the first live lane's
        // counter takes all lanes that share it" turn those 64 atomics into one; what is left after four rounds is spread out and goes lane by lane.
        int ci = (s >= slot_lo && s < slot_lo + FS_SLOTS) ? (s - slot_lo) * 256 + (int)((key >> sh_dig) & 255ull) : -1;
#pragma unroll 1
        for (int round = 0; round < 4; ++round) {
            const unsigned long long live = __ballot(ci >= 0);
            if (live == 0ull) break;
            const int lead = __ffsll((long long)live) - 1;
            const int cl = __shfl(ci, lead, 64);
            const unsigned long long same = __ballot(ci == cl);
            if ((int)(threadIdx.x & 63) == lead) atomicAdd(&lh[cl >> 1], (uint32_t)__popcll(same) << (16 * (cl & 1)));
            if (ci == cl) ci = -1;
        }
        if (ci >= 0) atomicAdd(&lh[ci >> 1], 1u << (16 * (ci & 1)));
        }
    }
    __syncthreads();
    const int nloc = min(FS_SLOTS, nslots - slot_lo) * 128;
    // every block merges the same counters and all blocks get here at about the same time: each starts somewhere else (same-address global atomics serialise:
    // level 2 of the bench distribution -- ~40 K busy counters, 256 adders each -- took 140 us of a 55 us average)
    const int rot = nloc > 0 ? (int)(((unsigned)blockIdx.x * 2654435761u) % (unsigned)nloc) & ~63 : 0;
    for (int k = threadIdx.x; k < nloc; k += FS_TB) {
        int i = k + rot;
        if (i >= nloc) i -= nloc;
        const uint32_t v = lh[i];
        if (v & 0xffffu) atomicAdd(&hist[(size_t)slot_lo * 256 + 2 * i], v & 0xffffu);
        if (v >> 16) atomicAdd(&hist[(size_t)slot_lo * 256 + 2 * i + 1], v >> 16);
    }
    }
    // The block that takes the last ticket of this level has every other block's counts behind it: it walks the boundaries one digit down.  Ordering: the
    // histogram atomics execute at the memory side; a wave's `s_waitcnt vmcnt(0)` returns when they have been acknowledged, the barrier collects the waves,
    // then ONE lane takes the ticket.  (__threadfence() here -- an L2 write-back + invalidate per wave, 4096 of them per launch -- made the whole select
    // 1.08 ms instead of 0.69.)  Acquire side: the last block alone, once per launch, invalidates what its caches may hold of the histogram and the state
    // (an agent-scope acquire fence) before it reads them with plain loads -- one fence per launch instead of 4096.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&tickets[level], 1u) == (uint32_t)(gridDim.x * gridDim.y) - 1u ? 1u : 0u;
    __syncthreads();
    if (s_last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        fs_resolve_body(C, level, nlevels, tb, st, hist);
    }
}

// one block (the last one of a level's histogram launch): walk every unresolved boundary one byte down, build the slot list of the next level, clear the histogram for it
__device__ void fs_resolve_body(int C, int level, int nlevels, int tb, FplState* __restrict__ st, uint32_t* __restrict__ hist) {
    __shared__ uint32_t tot[FS_MAXC * FS_BINS];
    __shared__ uint32_t ncls[FS_MAXC + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nslots = level == 0 ? C : (int)st->nslots;
    if (level > 0 && nslots == 0) return;                          // everything was resolved earlier
    // exclusive prefix sums of every slot's 256 counters, in place; tot[s] = the slot's element count (one wave per slot, 4 digits per lane)
    for (int s = wave; s < nslots; s += FS_TB / 64) {
        uint32_t* h = hist + (size_t)s * 256;
        const uint4 v = reinterpret_cast<const uint4*>(h)[lane];
        const uint32_t mine = v.x + v.y + v.z + v.w;
        uint32_t inc = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        const uint32_t ex = inc - mine;
        reinterpret_cast<uint4*>(h)[lane] = make_uint4(ex, ex + v.x, ex + v.x + v.y, ex + v.x + v.y + v.z);
        if (lane == 63) tot[s] = inc;
    }
    __syncthreads();
    if (tid < C * FS_BINS) {
        const int c = tid / FS_BINS, b = tid % FS_BINS, id = c * FS_BINS + b;
        uint32_t state, resid;
        int slot;
        unsigned long long pf;
        if (level == 0) {
            const uint32_t n = tot[c], nbin = n / FS_BINS;
            if (b == 0) { st->counts[c] = n; st->nb[c] = nbin; }
            const unsigned long long t = (unsigned long long)(b + 1) * nbin;
            state = nbin == 0 ? 3u : (t >= n ? 2u : 0u);
            resid = (uint32_t)t; slot = c; pf = 0;
        } else { state = st->bnd_state[id]; resid = st->bnd_resid[id]; slot = st->bnd_slot[id]; pf = st->bnd_prefix[id]; }
        if (state == 0) {
            const uint32_t* h = hist + (size_t)slot * 256;
            int lo = 0, hi = 255;                                  // largest digit d with excl[d] <= resid
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (h[mid] <= resid) lo = mid; else hi = mid - 1; }
            const uint32_t cnt = (lo < 255 ? h[lo + 1] : tot[slot]) - h[lo];
            pf = (pf << 8) | (unsigned long long)lo;
            resid -= h[lo];
            if (cnt == 1u || level + 1 == nlevels) state = 1u;
            st->bnd_prefix[id] = pf; st->bnd_resid[id] = resid; st->bnd_shift[id] = (uint32_t)(tb - 8 * (level + 1));
        }
        st->bnd_state[id] = state;
    }
    __syncthreads();
    // next level's slots: the distinct prefixes of the boundaries that are still open (boundaries of a class are ordered by rank, so equal
    // prefixes are adjacent)
    if (tid < C) {
        uint32_t k = 0;
        unsigned long long last = 0;
        for (int b = 0; b < FS_BINS; ++b)
            if (st->bnd_state[tid * FS_BINS + b] == 0) { const unsigned long long p = st->bnd_prefix[tid * FS_BINS + b]; if (k == 0 || p != last) { ++k; last = p; } }
        ncls[tid] = k;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t a = 0;
        for (int c = 0; c < C; ++c) { st->slot_base[c] = a; a += ncls[c]; }
        st->slot_base[C] = a;
        st->nslots = a;
        ncls[FS_MAXC] = a;
    }
    __syncthreads();
    if (tid < C) {
        uint32_t k = st->slot_base[tid];
        unsigned long long last = 0;
        bool any = false;
        for (int b = 0; b < FS_BINS; ++b) {
            const int id = tid * FS_BINS + b;
            if (st->bnd_state[id] == 0) {
                const unsigned long long p = st->bnd_prefix[id];
                if (!any || p != last) { st->slot_prefix[k] = p; ++k; last = p; any = true; }
                st->bnd_slot[id] = (int)k - 1;
            }
        }
    }
    const uint32_t nz = ncls[FS_MAXC] * 256u;
    for (uint32_t i = tid; i < nz; i += FS_TB) hist[i] = 0;            // counters of the next level (every read of this level is behind a barrier)
}

// bin of every pixel: one thread per pixel, binary search over the class's 32 (prefix, shift) boundary records
__global__ void __launch_bounds__(FS_TB) k_fs_assign(const uint8_t* __restrict__ lab, const float* __restrict__ prob, int64_t M, int C, int ib,
                                                     const FplState* __restrict__ st, uint8_t* __restrict__ binmap) {
    __shared__ unsigned long long tp[FS_MAXC * FS_BINS];       // boundary prefixes
    __shared__ uint32_t ts[FS_MAXC * FS_BINS];                 // shift (0xffffffff: never reached, 0xfffffffe: class without a full bin)
    for (int i = threadIdx.x; i < C * FS_BINS; i += FS_TB) {
        const uint32_t s = st->bnd_state[i];
        tp[i] = st->bnd_prefix[i];
        ts[i] = s == 1u ? st->bnd_shift[i] : (s == 3u ? 0xfffffffeu : 0xffffffffu);
    }
    __syncthreads();
    for (int64_t p = (int64_t)blockIdx.x * FS_TB + threadIdx.x; p < M; p += (int64_t)gridDim.x * FS_TB) {
        const int c = lab[p];
        int bin = FS_BINS;
        if (c < C && ts[c * FS_BINS] != 0xfffffffeu) {
            const unsigned long long key = fs_key(prob[p], (uint32_t)p, ib);
            int lo = 0, hi = FS_BINS;               // number of boundaries b with key >= boundary_b (monotone: boundaries ascend with b)
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                if (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    const uint32_t sh = ts[c * FS_BINS + mid];
                    const bool ge = sh != 0xffffffffu && (key >> sh) >= tp[c * FS_BINS + mid];
                    if (ge) lo = mid + 1; else hi = mid;
                }
            }
            bin = lo;
        }
        binmap[p] = bin < FS_BINS ? (uint8_t)bin : (uint8_t)255;
    }
}

// bin sums in PIXEL order: an 8-lane group (4 channels per lane) walks FS_RUN consecutive pixels and keeps the running sum in registers while
// (class, bin) stays the same -- bins of tied probabilities are contiguous pixel ranges (ties break by index) and labels are layered, so
// neighbouring pixels usually share their bin; eight groups of a wave adding to the SAME 32 LDS words would serialise (the first version, one
// pixel per group and an LDS atomic per value, took 1.34 ms at the bench shape against 0.61 ms for the gather it replaced).  The group's 32 labels
// and bins are fetched with two 16-byte loads each, the feature rows four pixels ahead.
template <typename T>
__global__ void __launch_bounds__(FS_TB) k_fs_binsum(const T* __restrict__ feat, const uint8_t* __restrict__ lab, const uint8_t* __restrict__ binmap,
                                                     int64_t M, int C, float* __restrict__ pro_sum) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* acc = reinterpret_cast<float*>(smem);                                                    // [C * 32][32]
    for (int i = threadIdx.x; i < C * FS_BINS * 32; i += FS_TB) acc[i] = 0.f;
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const int64_t ngroups = M / FS_RUN;                 // whole runs (M % FS_RUN pixels are handled below, one by one)
    for (int64_t grp = ((int64_t)blockIdx.x * FS_TB + threadIdx.x) >> 3; grp < ngroups; grp += ((int64_t)gridDim.x * FS_TB) >> 3) {
        const int64_t p0 = grp * FS_RUN;
        const uint4 l0 = *reinterpret_cast<const uint4*>(lab + p0), l1 = *reinterpret_cast<const uint4*>(lab + p0 + 16);
        const uint4 b0 = *reinterpret_cast<const uint4*>(binmap + p0), b1 = *reinterpret_cast<const uint4*>(binmap + p0 + 16);
        const uint32_t lw[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w}, bw[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        f4 run = f4zero();
        int cur = -1;
#pragma unroll
        for (int q4 = 0; q4 < FS_RUN / 4; ++q4) {
            f4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ld4(feat + (p0 + 4 * q4 + j) * 32 + sub * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = (int)((lw[q4] >> (8 * j)) & 255u), bin = (int)((bw[q4] >> (8 * j)) & 255u);
                const int id = (bin < FS_BINS && c < C) ? c * FS_BINS + bin : -1;
                if (id != cur) {
                    if (cur >= 0) {
                        float* a = acc + cur * 32 + sub * 4;
#pragma unroll
                        for (int k = 0; k < 4; ++k) atomicAdd(a + k, run.v[k]);
                    }
                    run = f4zero();
                    cur = id;
                }
                if (id >= 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) run.v[k] += v[j].v[k];
                }
            }
        }
        if (cur >= 0) {
            float* a = acc + cur * 32 + sub * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) atomicAdd(a + k, run.v[k]);
        }
    }
    if (blockIdx.x == 0) {                              // the ragged tail (< FS_RUN pixels)
        for (int64_t p = ngroups * FS_RUN + (threadIdx.x >> 3); p < M; p += FS_TB >> 3) {
            const int c = lab[p], bin = binmap[p];
            if (bin < FS_BINS && c < C) {
                const f4 v = ld4(feat + p * 32 + sub * 4);
                float* a = acc + (c * FS_BINS + bin) * 32 + sub * 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) atomicAdd(a + k, v.v[k]);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * FS_BINS * 32; i += FS_TB)
        if (acc[i] != 0.f) atomicAdd(&pro_sum[i], acc[i]);
}

// The bin sums as a GEMM on the matrix pipes (bf16 features): pro_sum[class][bin][ch] = sum_p onehot[p][class, bin] feat[p][ch], i.e.
// D[bin][ch] += A[bin][pixel] B[pixel][ch] per class with A the 0/1 indicator "pixel p belongs to (class, bin)" built in registers from the pixel's
// id (8 compares per lane and 16-pixel step) and B the feature rows, transposed on the LDS read (ds_read_b64_tr_b16, as in the weight-gradient
// kernels).  Products with 1.0 are exact and the accumulation is fp32, so this is the same sum as the scalar path -- without 226 M LDS float
// atomics (ds_add_f32 of eight 8-lane groups landing on the same eight banks: 1.13 ms at the bench shape; this kernel: ~2.2 M MFMAs).
// One M-tile per class (rows = its 32 bins), classes dealt to the four waves round-robin (<= 4 accumulators per wave for 16 classes).
typedef __attribute__((ext_vector_type(8))) __bf16 fs_bf16x8;
typedef __attribute__((ext_vector_type(16))) float fs_f32x16;
typedef __attribute__((ext_vector_type(4))) short fs_s16x4;
typedef __attribute__((ext_vector_type(8))) short fs_s16x8;
__global__ void __launch_bounds__(256) k_fs_binsum_mfma(const bf16* __restrict__ feat, const uint8_t* __restrict__ lab, const uint8_t* __restrict__ binmap,
                                                        int64_t M, int C, float* __restrict__ pro_sum) {
    __shared__ __attribute__((aligned(16))) unsigned char sX[128 * 64];      // 128 pixels x 32 channels bf16, linear rows
    __shared__ __attribute__((aligned(16))) unsigned short sI[128];          // id = class * 32 + bin of the pixel, 0xffff: none
    __shared__ uint32_t sMask[2];                                             // classes present in the tile (double-buffered by tile parity)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    fs_f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    const int li = lane & 15, lq = li >> 2, lpp = li & 3, lg = lane >> 4;
    const unsigned char* lbX = sX + (8 * (lg >> 1) + lq) * 64 + (16 * (lg & 1) + 4 * lpp) * 2;
    const int64_t tiles = (M + 127) / 128;
    int par = 0;
    if (tid < 2) sMask[tid] = 0u;
    // the NEXT tile's 8 KB of feature rows and its (label, bin) bytes are requested while this tile's MFMAs run (round 5: the kernel waited a full memory
    // latency per tile between two block barriers: 0.249 ms for 0.47 GB)
    uint4 nv[2];
    int nc = 255, nb = 255;
    auto fetch = [&](int64_t tile) {
        const int64_t q0 = tile * 128;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int q = tid + j * 256;                    // 16-byte chunk q of the tile's contiguous 8 KB: pixel q / 4, chunk q % 4
            nv[j] = make_uint4(0u, 0u, 0u, 0u);
            if (tile < tiles && q0 + (q >> 2) < M) nv[j] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(feat) + (q0 * 64 + (int64_t)q * 16));
        }
        nc = nb = 255;
        if (tid < 128 && tile < tiles && q0 + tid < M) { nc = lab[q0 + tid]; nb = binmap[q0 + tid]; }
    };
    fetch(blockIdx.x);
    for (int64_t tile = blockIdx.x; tile < tiles; tile += gridDim.x, par ^= 1) {
        __syncthreads();
        if (tid == 0) sMask[par ^ 1] = 0u;                  // the other buffer: read by nobody any more, written from the next tile on
#pragma unroll
        for (int j = 0; j < 2; ++j) *reinterpret_cast<uint4*>(sX + (tid + j * 256) * 16) = nv[j];
        const int cc_ = nc, cb_ = nb;
        fetch(tile + gridDim.x);
        if (tid < 128) {
            unsigned short id = 0xffffu;
            if (cc_ < C && cb_ < FS_BINS) id = (unsigned short)(cc_ * FS_BINS + cb_);
            sI[tid] = id;
            // 128 consecutive pixels are a piece of ONE image row and the layers are horizontal bands: a tile usually holds one or two classes, and a
            // wave skips the others (8 compares + an MFMA per class and 16-pixel step otherwise: the kernel was VALU-bound on building indicators)
            const unsigned long long any = __ballot(id != 0xffffu);
            if (any) {
                uint32_t m = id != 0xffffu ? 1u << (id >> 5) : 0u;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) m |= __shfl_xor(m, o, 64);
                if ((tid & 63) == 0) atomicOr(&sMask[par], m);
            }
        }
        __syncthreads();
        const uint32_t present = sMask[par];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const unsigned char* pb = lbX + ch * 16 * 64;
            const fs_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fs_s16x4*)pb);
            const fs_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fs_s16x4*)(pb + 4 * 64));
            const fs_bf16x8 fb = __builtin_bit_cast(fs_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            const uint4 iv = *reinterpret_cast<const uint4*>(sI + ch * 16 + 8 * hh);       // the ids of this lane's 8 pixels
            const uint32_t iw[4] = {iv.x, iv.y, iv.z, iv.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int cls = wave + 4 * t;
                if (cls < C && ((present >> cls) & 1u)) {       // wave-uniform
                    const uint32_t want = (uint32_t)(cls * FS_BINS + r);
                    fs_s16x8 a;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t id = (j & 1) ? (iw[j >> 1] >> 16) : (iw[j >> 1] & 0xffffu);
                        a[j] = id == want ? (short)0x3f80 : (short)0;           // bf16 1.0 / 0.0
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(fs_bf16x8, a), fb, acc[t], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int cls = wave + 4 * t;
        if (cls < C) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int bin = (k & 3) + 8 * (k >> 2) + 4 * hh;
                if (acc[t][k] != 0.f) atomicAdd(&pro_sum[((int64_t)cls * FS_BINS + bin) * 32 + r], acc[t][k]);
            }
        }
    }
}

extern "C" int64_t tcct_fpl_select_workspace_bytes() {
    return (int64_t)sizeof(FplState) + (int64_t)FS_MAXC * FS_BINS * 256 * sizeof(uint32_t) + 8 * sizeof(uint32_t);
}

/* labels uint8 [M], prob fp32 [M] (softmax probability of the labelled class, detached), feat [M,32]: binmap [M] (bin 0..31 of the pixel inside
 * its class, 255 = dropped tail / class without a full bin), counts [16] uint32 (pixels per class), pro_sum [C][32][32] fp32 (feature sums per
 * class and bin; cleared here).  workspace: tcct_fpl_select_workspace_bytes() bytes.  Replaces tcct_fpl_sort + the gather in tcct_fpl_forward. */
extern "C" int tcct_fpl_select(const void* feat, const uint8_t* labels, const float* prob, int64_t M, int C, void* workspace, uint32_t* counts,
                               uint8_t* binmap, float* pro_sum, int dtype, tcct_stream_t stream) {
    TCCT_CHECK(C >= 1 && C <= FS_MAXC, "fpl_select: C=%d unsupported (1..%d)", C, FS_MAXC);
    TCCT_CHECK(M > 0 && M < (1LL << 32), "fpl_select: M out of range");
    hipStream_t st = (hipStream_t)stream;
    FplState* state = (FplState*)workspace;
    uint32_t* hist = (uint32_t*)((unsigned char*)workspace + sizeof(FplState));
    int ib = 1;
    while ((1LL << ib) < M) ++ib;
    ib = (ib + 7) / 8 * 8;                                  // index bits, whole bytes
    const int tb = 32 + ib, nlevels = tb / 8;
    uint32_t* tickets = hist + (size_t)FS_MAXC * FS_BINS * 256;          // [8] level tickets behind the histogram
    if (hipMemsetAsync(hist, 0, sizeof(uint32_t) * FS_MAXC * 256, st) != hipSuccess) { tcct_set_error("fpl_select: memset failed"); return -2; }
    if (hipMemsetAsync(tickets, 0, sizeof(uint32_t) * 8, st) != hipSuccess) { tcct_set_error("fpl_select: memset failed"); return -2; }
    if (hipMemsetAsync(pro_sum, 0, sizeof(float) * C * FS_BINS * 32, st) != hipSuccess) { tcct_set_error("fpl_select: memset failed"); return -2; }
    const size_t lds_h = (size_t)FS_SLOTS * 128 * 4 + (size_t)FS_MAXC * FS_BINS * 8 + (FS_MAXC + 1) * 4;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)k_fs_hist, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); attr = true; }      // + ~2 KB static (the resolve step)
    int grid = tcct_grid(M, FS_TB, 256);            // one 1024-thread block per CU (128 blocks: 0.46 ms for the seven levels against 0.33)
    if ((M + grid - 1) / grid > FS_BLOCK_PIX) grid = (int)((M + FS_BLOCK_PIX - 1) / FS_BLOCK_PIX);       // 16-bit block-private counters
    TCCT_CHECK(nlevels <= 8, "fpl_select: %d radix levels", nlevels);
    const int passes = (C * FS_BINS + FS_SLOTS - 1) / FS_SLOTS;
    for (int level = 0; level < nlevels; ++level)           // histogram of the level + (in its last block) the resolve step: one launch
        hipLaunchKernelGGL(k_fs_hist, dim3(grid, level == 0 ? 1 : passes), dim3(FS_TB), lds_h, st, labels, prob, M, C, level, ib, tb, passes, nlevels, state, hist, tickets);
    hipLaunchKernelGGL(k_fs_assign, dim3(tcct_grid(M, FS_TB, 1024)), dim3(FS_TB), 0, st, labels, prob, M, C, ib, (const FplState*)state, binmap);
    const size_t lds_a = (size_t)C * FS_BINS * 32 * 4;
    const int gb = tcct_grid((M / FS_RUN + 1) * 8, FS_TB, 512);
    if (dtype == TCCT_F32) {
        static bool a32 = false;
        if (!a32) { (void)hipFuncSetAttribute((const void*)k_fs_binsum<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a32 = true; }
        hipLaunchKernelGGL(k_fs_binsum<float>, dim3(gb), dim3(FS_TB), lds_a, st, (const float*)feat, labels, (const uint8_t*)binmap, M, C, pro_sum);
    } else if (dtype == TCCT_BF16) {
        static bool a16 = false;
        if (!a16) { (void)hipFuncSetAttribute((const void*)k_fs_binsum<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a16 = true; }
        (void)a16;
        const int64_t tiles = (M + 127) / 128;
        hipLaunchKernelGGL(k_fs_binsum_mfma, dim3((unsigned)(tiles < 1536 ? tiles : 1536)), dim3(256), 0, st, (const bf16*)feat, labels, (const uint8_t*)binmap, M, C, pro_sum);
    } else { tcct_set_error("fpl_select: bad dtype %d", dtype); return -1; }
    if (hipMemcpyAsync(counts, state->counts, sizeof(uint32_t) * FS_MAXC, hipMemcpyDeviceToDevice, st) != hipSuccess) { tcct_set_error("fpl_select: copy failed"); return -2; }
    TCCT_LAUNCH_OK();
}
