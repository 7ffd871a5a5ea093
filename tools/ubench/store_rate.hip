// Round 4 probe: what bounds the output tail of a one-tile-per-CU GEMM (256 workgroups each writing a 256 x 256 BF16 tile = 128 KiB,
// 33.5 MB chip-wide)?  HISTORY.md measured 10.2 k cycles for it (~12.8 B/clk/CU, 5.6 TB/s chip-wide) with plain stores.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/store_rate tools/ubench/store_rate.hip && tools/ubench/store_rate
// Every workgroup (8 waves) writes its tile of a 4096 x 4096 BF16 matrix the way store_rows_full_line does: one wave instruction =
// 8 rows x 128 contiguous bytes (16 bytes per lane).  Modes = the cache-policy bits of the store (none / nt / sc0 / sc1 / sc0 sc1 /
// sc0 sc1 nt), a linear 128 KiB per workgroup (no row pitch) and a launch with only 32 workgroups (per-CU limit or chip limit?).
// Output: us per launch (events over 50 launches), TB/s, and the median per-workgroup s_memtime span.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int N = 4096, PITCH = N * 2;

template <int POLICY>
__device__ __forceinline__ void store16(uint8_t* p, v4i v) {
    if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
}

// LINEAR: workgroup b writes bytes [b * 128 KiB, (b + 1) * 128 KiB) instead of a 256 x 256 tile
template <int POLICY, bool LINEAR>
__global__ __launch_bounds__(512) void store_kernel(uint8_t* d, long long* cycles, int spin) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4i v = {lane, wave, (int)blockIdx.x, spin};
    // a little work first so that all workgroups are resident and start storing together
    float x = lane;
    for (int i = 0; i < spin; ++i)
        x = __builtin_fmaf(x, 1.0001f, 0.5f);
    v[3] = __float_as_int(x);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if constexpr (LINEAR) {
        uint8_t* base = d + (size_t)blockIdx.x * 131072 + wave * 16384;
        #pragma unroll
        for (int i = 0; i < 16; ++i)
            store16<POLICY>(base + i * 1024 + lane * 16, v);
    } else {
        const int tm = blockIdx.x >> 4, tn = blockIdx.x & 15;
        // wave (wm = wave >> 2, wn = wave & 3): 128 rows x 64 columns (128 bytes); instruction i covers rows i * 8 .. i * 8 + 7
        uint8_t* base = d + (size_t)(tm * 256 + (wave >> 2) * 128) * PITCH + tn * 512 + (wave & 3) * 128;
        #pragma unroll
        for (int i = 0; i < 16; ++i)
            store16<POLICY>(base + (size_t)(i * 8 + (lane >> 3)) * PITCH + (lane & 7) * 16, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
}

typedef void (*kernel_t)(uint8_t*, long long*, int);

int main() {
    uint8_t* d;
    long long* cyc;
    CHECK(hipMalloc(&d, (size_t)N * PITCH));
    CHECK(hipMalloc(&cyc, 256 * 8));
    struct { const char* name; kernel_t k; int grid; } modes[] = {
        {"plain", store_kernel<0, false>, 256}, {"nt", store_kernel<1, false>, 256}, {"sc0", store_kernel<2, false>, 256},
        {"sc1", store_kernel<3, false>, 256}, {"sc0 sc1", store_kernel<4, false>, 256}, {"sc0 sc1 nt", store_kernel<5, false>, 256},
        {"plain linear", store_kernel<0, true>, 256}, {"nt linear", store_kernel<1, true>, 256},
        {"plain 32 wg", store_kernel<0, false>, 32}, {"plain 64 wg", store_kernel<0, false>, 64}, {"plain 128 wg", store_kernel<0, false>, 128},
    };
    printf("%-14s %10s %10s %14s %14s   (128 KiB per workgroup; s_memtime cycles)\n", "mode", "us/launch", "TB/s", "wg cyc med", "wg cyc max");
    for (int rep = 0; rep < 2; ++rep)
        for (auto& m : modes) {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            for (int i = 0; i < 5; ++i)
                hipLaunchKernelGGL(m.k, dim3(m.grid), dim3(512), 0, 0, d, cyc, 300);
            CHECK(hipEventRecord(a));
            for (int i = 0; i < 50; ++i)
                hipLaunchKernelGGL(m.k, dim3(m.grid), dim3(512), 0, 0, d, cyc, 300);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            // the same launch without stores costs: measured by the spin-only variant below (grid 256, spin 2000, no stores) -- subtract by eye
            std::vector<long long> c(256);
            CHECK(hipMemcpy(c.data(), cyc, m.grid * 8, hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.begin() + m.grid);
            if (rep == 1)
                printf("%-14s %10.2f %10.2f %14lld %14lld\n", m.name, ms * 1e3 / 50, m.grid * 131072.0 / (ms * 1e-3 / 50) / 1e12, c[m.grid / 2], c[m.grid - 1]);
        }
    return 0;
}
