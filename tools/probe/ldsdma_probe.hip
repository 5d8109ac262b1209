// LDS-DMA probe (gfx950): buffer_load_dwordx4 ... lds -- destination layout and the out-of-range behaviour
//   hipcc --offload-arch=gfx950 -O3 -o tools/probe/ldsdma_probe.bin tools/probe/ldsdma_probe.hip && ./tools/probe/ldsdma_probe.bin   (measured: OK)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4;
__global__ void k(const uint32_t* src, uint32_t* out, uint32_t nbytes) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    // lane l fetches the 16-byte chunk (63 - l): per-lane SOURCE address; lanes >= 48 use an offset beyond the descriptor range
    const uint32_t off = threadIdx.x < 48 ? (63u - threadIdx.x) * 16u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, off, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 256), 16, threadIdx.x * 16u, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<uint32_t> h(256); for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    uint32_t *d, *o; hipMalloc(&d, 1024); hipMalloc(&o, 2048);
    hipMemcpy(d, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1024u);
    std::vector<uint32_t> r(512); hipMemcpy(r.data(), o, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        uint32_t want = l < 48 ? 1000 + (63 - l) * 4 + j : 0u;
        if (r[l * 4 + j] != want) { if (bad < 8) printf("first: lane %d dword %d got %u (0x%x) want %u\n", l, j, r[l * 4 + j], r[l * 4 + j], want); ++bad; }
        if (r[256 + l * 4 + j] != 1000u + l * 4 + j) { if (bad < 8) printf("second: lane %d dword %d got %u\n", l, j, r[256 + l * 4 + j]); ++bad; }
    }
    printf("ldsdma probe: %s (%d mismatches); out-of-range lanes wrote 0x%x\n", bad ? "MISMATCH" : "OK: LDS dst = base + 16*lane, per-lane source, out-of-range lanes write zeros", bad, r[48 * 4]);
    return bad != 0;
}
