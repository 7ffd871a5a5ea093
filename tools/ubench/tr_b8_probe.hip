// Probe of ds_read_b64_tr_b8 (gfx950 LDS transpose read, 8-bit elements): which LDS bytes does lane l receive, given the
// per-lane addresses?  LDS byte at offset a holds (a & 0xff) in pass 0 and (a >> 8) in pass 1, so the two passes together
// give the 16-bit source offset of every returned byte.  Build: hipcc --offload-arch=gfx950 -O2 tr_b8_probe.hip -o tr_b8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const int* addr, uint32_t* out, int pass) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64)
        lds[i] = pass == 0 ? (i & 0xff) : (i >> 8);
    __syncthreads();
    // the array's address must escape into the asm, or hipcc drops the fill as dead stores
    const int a = addr[threadIdx.x] + static_cast<int>(reinterpret_cast<uintptr_t>(lds));
    uint32_t lo, hi;
    typedef uint32_t v2u __attribute__((ext_vector_type(2)));
    v2u r;
    asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a), "v"(lds) : "memory");
    lo = r[0]; hi = r[1];
    out[threadIdx.x * 2] = lo;
    out[threadIdx.x * 2 + 1] = hi;
}

static void run(const char* name, const std::vector<int>& addr) {
    int* d_addr; uint32_t* d_out;
    hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_out, 128 * 4);
    hipMemcpy(d_addr, addr.data(), 64 * 4, hipMemcpyHostToDevice);
    uint32_t res[2][128];
    for (int pass = 0; pass < 2; ++pass) {
        hipMemset(d_out, 0xee, 128 * 4);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_addr, d_out, pass);
        hipError_t e1 = hipGetLastError(), e2 = hipDeviceSynchronize();
        if (e1 != hipSuccess || e2 != hipSuccess) printf("launch %s / sync %s\n", hipGetErrorString(e1), hipGetErrorString(e2));
        hipMemcpy(res[pass], d_out, 128 * 4, hipMemcpyDeviceToHost);
    }
    printf("== %s\n", name);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d addr %4d:", l, addr[l]);
        for (int j = 0; j < 8; ++j) {
            const int lo = (res[0][l * 2 + j / 4] >> (8 * (j % 4))) & 0xff, hi = (res[1][l * 2 + j / 4] >> (8 * (j % 4))) & 0xff;
            printf(" %4d", hi * 256 + lo);
        }
        printf("\n");
    }
    hipFree(d_addr); hipFree(d_out);
}

int main() {
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) a[l] = l * 8;                       // consecutive 8-byte blocks
    run("addr = lane * 8", a);
    for (int l = 0; l < 64; ++l) a[l] = (l >> 4) * 2048 + ((l & 15) >> 1) * 256 + (l & 1) * 8;   // 8 rows of pitch 256 per 16-lane group
    run("addr = group * 2048 + ((lane & 15) >> 1) * 256 + (lane & 1) * 8", a);
    for (int l = 0; l < 64; ++l) a[l] = (l & 15) * 128 + (l >> 4) * 8;   // one row per lane
    run("addr = (lane & 15) * 128 + (lane >> 4) * 8", a);
    return 0;
}
