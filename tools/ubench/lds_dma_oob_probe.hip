// What an LDS-DMA piece (buffer_load_dwordx4 ... lds) leaves in LDS for lanes whose address fails the buffer range check:
// zeros, or the old LDS bytes?  (The K-tail handling of the fast GEMM kernels relies on the answer.)  Also: does the SGPR offset
// take part in the range check?   hipcc --offload-arch=gfx950 -O2 lds_dma_oob_probe.hip -o /tmp/oob && /tmp/oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(const uint8_t* src, uint32_t* out, int num_records, int soffset) {
    __shared__ __attribute__((aligned(1024))) uint32_t lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xABABABABu;
    __syncthreads();
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(src), 0, num_records, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, threadIdx.x * 16, soffset, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    uint8_t h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 1 + (i / 16) % 200;
    uint8_t* d; uint32_t* o; uint32_t r[256];
    hipMalloc(&d, 4096); hipMalloc(&o, 1024);
    hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
    const int cases[3][2] = {{512, 0}, {1024, 512}, {2048, 512}};
    for (auto& c : cases) {
        probe<<<1, 64>>>(d, o, c[0], c[1]);
        hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
        printf("num_records=%d soffset=%d: first dword of the 16-byte chunk written by lane 0,8,..,56:", c[0], c[1]);
        for (int l = 0; l < 64; l += 8) printf(" %08x", r[l * 4]);
        printf("\n");
    }
    return 0;
}
