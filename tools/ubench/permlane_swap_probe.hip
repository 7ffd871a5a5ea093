// What v_permlane32_swap / v_permlane16_swap do to two registers (gfx950): prints, per lane, the source (register, lane) of both
// results.  hipcc --offload-arch=gfx950 -O2 permlane_swap_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
__global__ void probe(unsigned* out) {
    const unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    const u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    const u2 s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x] = r[0];
    out[64 + threadIdx.x] = r[1];
    out[128 + threadIdx.x] = s[0];
    out[192 + threadIdx.x] = s[1];
}
int main() {
    unsigned* d;
    unsigned h[256];
    hipMalloc(&d, sizeof(h));
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[4] = {"permlane32_swap vdst", "permlane32_swap src ", "permlane16_swap vdst", "permlane16_swap src "};
    for (int q = 0; q < 4; ++q) {
        printf("%s:", names[q]);
        for (int l = 0; l < 64; l += 8)
            printf(" [%d]=%u", l, h[q * 64 + l]);
        printf("\n");
    }
    return 0;
}
