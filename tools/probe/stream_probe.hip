// Which shape of a streaming kernel reaches the copy rate on the MI355X?  y[m][c] = a[c] * x[m][c] + b[c] on bf16 [M][32] (the BatchNorm apply).
//   hipcc --offload-arch=gfx950 -O3 -o stream_probe stream_probe.hip && ./stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned short u16;
__device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pk(float a, float b) {
    uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    x += 0x7fffu + ((x >> 16) & 1u); y += 0x7fffu + ((y >> 16) & 1u);
    return (x >> 16) | (y & 0xffff0000u);
}
// A: persistent, thread = fixed 4-channel vector, 2 rows in flight, 8-byte accesses (k_bn_apply today)
__global__ void kA(const u16* x, u16* y, int64_t M, const float* ab) {
    const int C = 32, CV = 8, R = 256 / CV, t = threadIdx.x, cv = t % CV, r = t / CV;
    float a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = ab[cv * 4 + k]; b[k] = ab[C + cv * 4 + k]; }
    const int64_t step = (int64_t)gridDim.x * R;
    for (int64_t m = (int64_t)blockIdx.x * R + r; m < M; m += 2 * step) {
        const int64_t m2 = m + step; const bool two = m2 < M;
        uint2 v1 = *(const uint2*)(x + m * C + cv * 4), v2 = make_uint2(0, 0);
        if (two) v2 = *(const uint2*)(x + m2 * C + cv * 4);
        uint2 o1, o2;
        o1.x = pk(a[0] * lo(v1.x) + b[0], a[1] * hi(v1.x) + b[1]); o1.y = pk(a[2] * lo(v1.y) + b[2], a[3] * hi(v1.y) + b[3]);
        o2.x = pk(a[0] * lo(v2.x) + b[0], a[1] * hi(v2.x) + b[1]); o2.y = pk(a[2] * lo(v2.y) + b[2], a[3] * hi(v2.y) + b[3]);
        *(uint2*)(y + m * C + cv * 4) = o1;
        if (two) *(uint2*)(y + m2 * C + cv * 4) = o2;
    }
}
// B: U x 16-byte accesses per thread, block-contiguous chunks, NP: one chunk per block (huge grid) / persistent grid-stride
template <int U, bool PERSIST>
__global__ void kB(const u16* x, u16* y, int64_t n16, const float* ab) {      // n16 = number of 16-byte groups (8 channels each; 4 groups per pixel)
    const int t = threadIdx.x, cg = t & 3;        // 256 % 4 == 0 and chunk bases are multiples of 4: the channel group of a thread is fixed
    float a[8], b[8];
    for (int k = 0; k < 8; ++k) { a[k] = ab[cg * 8 + k]; b[k] = ab[32 + cg * 8 + k]; }
    const int64_t chunk = 256 * U;
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < n16; base += PERSIST ? (int64_t)gridDim.x * chunk : n16) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = base + u * 256 + t; v[u] = i < n16 ? ((const uint4*)x)[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256 + t;
            uint4 o;
            o.x = pk(a[0] * lo(v[u].x) + b[0], a[1] * hi(v[u].x) + b[1]); o.y = pk(a[2] * lo(v[u].y) + b[2], a[3] * hi(v[u].y) + b[3]);
            o.z = pk(a[4] * lo(v[u].z) + b[4], a[5] * hi(v[u].z) + b[5]); o.w = pk(a[6] * lo(v[u].w) + b[6], a[7] * hi(v[u].w) + b[7]);
            if (i < n16) ((uint4*)y)[i] = o;
        }
    }
}
// C: like B with 8-byte accesses
template <int U, bool PERSIST>
__global__ void kC(const u16* x, u16* y, int64_t n8, const float* ab) {
    const int t = threadIdx.x, cg = t & 7;
    float a[4], b[4];
    for (int k = 0; k < 4; ++k) { a[k] = ab[cg * 4 + k]; b[k] = ab[32 + cg * 4 + k]; }
    const int64_t chunk = 256 * U;
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < n8; base += PERSIST ? (int64_t)gridDim.x * chunk : n8) {
        uint2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int64_t i = base + u * 256 + t; v[u] = i < n8 ? ((const uint2*)x)[i] : make_uint2(0, 0); }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = base + u * 256 + t;
            uint2 o;
            o.x = pk(a[0] * lo(v[u].x) + b[0], a[1] * hi(v[u].x) + b[1]); o.y = pk(a[2] * lo(v[u].y) + b[2], a[3] * hi(v[u].y) + b[3]);
            if (i < n8) ((uint2*)y)[i] = o;
        }
    }
}
template <typename F> float timeit(F f, int iters = 20) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0); for (int i = 0; i < iters; ++i) f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / iters;
}
int main() {
    const int64_t M = 8ll * 800 * 1104, C = 32, n = M * C;
    u16 *x, *y; float* ab;
    hipMalloc(&x, n * 2); hipMalloc(&y, n * 2); hipMalloc(&ab, 64 * 4);
    hipMemset(x, 0x3f, n * 2); std::vector<float> h(64, 1.0f); hipMemcpy(ab, h.data(), 256, hipMemcpyHostToDevice);
    const double gb = 2.0 * n * 2 / 1e9;
    auto rep = [&](const char* name, float ms) { printf("%-44s %.3f ms  %.0f GB/s\n", name, ms, gb / ms * 1e3); };
    rep("hipMemcpyDtoD", timeit([&] { hipMemcpyAsync(y, x, n * 2, hipMemcpyDeviceToDevice, 0); }));
    for (int g : {2048, 4096, 8192}) { char s[64]; snprintf(s, 64, "A persistent 8B x2 rows, grid %d", g); rep(s, timeit([&] { hipLaunchKernelGGL(kA, dim3(g), dim3(256), 0, 0, x, y, M, ab); })); }
    const int64_t n16 = n / 8, n8 = n / 4;
#define NP16(U) rep("B 16B x" #U " one chunk per block", timeit([&] { hipLaunchKernelGGL((kB<U, false>), dim3((unsigned)((n16 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, x, y, n16, ab); }))
    NP16(1); NP16(2); NP16(4); NP16(8);
#define P16(U, G) rep("B 16B x" #U " persistent grid " #G, timeit([&] { hipLaunchKernelGGL((kB<U, true>), dim3(G), dim3(256), 0, 0, x, y, n16, ab); }))
    P16(2, 2048); P16(4, 2048); P16(4, 4096); P16(8, 2048); P16(2, 8192); P16(4, 1024);
#define NP8(U) rep("C 8B x" #U " one chunk per block", timeit([&] { hipLaunchKernelGGL((kC<U, false>), dim3((unsigned)((n8 + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, x, y, n8, ab); }))
    NP8(1); NP8(2); NP8(4); NP8(8);
#define P8(U, G) rep("C 8B x" #U " persistent grid " #G, timeit([&] { hipLaunchKernelGGL((kC<U, true>), dim3(G), dim3(256), 0, 0, x, y, n8, ab); }))
    P8(4, 2048); P8(8, 2048); P8(4, 4096);
    return 0;
}
