// Round 5 probe for VERDICT item 1(c): can ONE wave per SIMD issue the FP32-scale step of a 4-wave 256 x 256 kernel inside its 32-cycle
// MFMA gap?  (Ubench before kernel; kill criterion: the 6-VALU arm above 36 cycles per step.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/issue_rate tools/ubench/issue_rate.hip && tools/ubench/issue_rate
// One workgroup per CU, 4 waves (one per SIMD).  A step = one v_mfma_f32_16x16x128_f8f6f4 with a ZERO C operand writing a 4-register
// result into VGPRs (the FP32-scale recipe promotes every K block: final += scale * partial) + NV VALU operations on data the MFMA of two
// steps ago produced (4 = the promotion's v_fmac_f32; 5, 6, 7 = with the v_accvgpr_read / write traffic of accumulators parked in AGPRs:
// 256 final accumulators + partials + scale products do not fit 256 arch VGPRs) + per step 0.5 ds_read_b128 and 0.25 LDS-DMA piece (what a
// 128 x 128 wave tile needs per MFMA: 32 reads and 16 pieces per 64 MFMAs).  Operands of the MFMA in VGPRs ("v") or in AGPRs ("a": the
// fragments would have to live there, ds_read_b128 straight into AGPRs).  MEM = 0: matrix + VALU only.
// Output: cycles per step (s_memtime, median over workgroups); the matrix floor is 32.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int PITCH = 7168, ROWS = 256;

template <int NV, bool AGPR_SRC, bool MEM>
__global__ __launch_bounds__(256) void step_kernel(const uint8_t* base, int steps, long long* cycles, float* sink) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[64 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, ROWS * PITCH + 8192, 0x00020000);
    const int piece_voff = (lane >> 3) * PITCH + (((lane & 7) ^ (lane >> 3)) << 4) + wave * 8 * PITCH;
    v8i afrag = {lane, 1, 2, 3, 4, 5, 6, 7}, bfrag = {7, 6, 5, 4, 3, 2, 1, lane};
    v4f fin[24];                    // 96 "final accumulator" registers (enough independent chains; the real kernel has 256)
    #pragma unroll
    for (int i = 0; i < 24; ++i) fin[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f part[4];                    // partial results in flight
    #pragma unroll
    for (int i = 0; i < 4; ++i) part[i] = v4f{1.f, 2.f, 3.f, 4.f};
    float scale = 1.0f + lane * 1e-3f, spare0 = 1.f, spare1 = 2.f, spare2 = 3.f;
    v4i ring[4] = {};
    const int rd_addr = static_cast<int>(reinterpret_cast<uintptr_t>(lds)) + lane * 16;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s += 48) {
        #pragma unroll
        for (int u = 0; u < 48; ++u) {
            v4f& dst = part[u & 3];
            if constexpr (AGPR_SRC)
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0" : "=v"(dst) : "a"(afrag), "a"(bfrag));
            else
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0" : "=v"(dst) : "v"(afrag), "v"(bfrag));
            // the promotion of the partial produced two steps ago (its MFMA has long retired: no dependency stall)
            const v4f& src = part[(u + 2) & 3];
            v4f& f = fin[u % 24];
            asm volatile("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %6\n\tv_fmac_f32 %2, %4, %7\n\tv_fmac_f32 %3, %4, %8"
                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(scale), "v"(src[0]), "v"(src[1]), "v"(src[2]), "v"(src[3]));
            if constexpr (NV >= 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(spare0) : "v"(scale));
            if constexpr (NV >= 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(spare1) : "v"(scale));
            if constexpr (NV >= 7) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(spare2) : "v"(scale));
            if constexpr (MEM) {
                if (u & 1)          // (asm: no compiler-placed wait behind it -- the real kernel consumes a fragment 16+ steps after its read)
                    asm volatile("ds_read_b128 %0, %1" : "+v"(ring[(u >> 1) & 3]) : "v"(rd_addr + (((u >> 1) & 31) * 1024)) : "memory");
                if ((u & 3) == 1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + 32768 + (((u >> 2) * 4 + wave) & 31) * 1024), 16,
                                                             piece_voff + ((u >> 2) & 7) * 32 * PITCH, ((s / 48) % 56) * 128, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MEM) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    float t = spare0 + spare1 + spare2;
    #pragma unroll
    for (int i = 0; i < 24; ++i) t += fin[i][0] + fin[i][3];
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (sink != nullptr && t == 12345.f) sink[threadIdx.x] = t + ring[0][0] + ring[1][1] + ring[2][2] + ring[3][3];
}

typedef void (*kernel_t)(const uint8_t*, int, long long*, float*);

int main() {
    uint8_t* buf; long long* cyc; float* sink;
    CHECK(hipMalloc(&buf, ROWS * PITCH + 8192));
    std::vector<uint8_t> host(ROWS * PITCH);
    srand(1);
    for (auto& b : host) b = static_cast<uint8_t>(rand() & 0x7f) % 0x7e;
    CHECK(hipMemcpy(buf, host.data(), host.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&cyc, 256 * 8));
    CHECK(hipMalloc(&sink, 4096));
    struct { const char* name; kernel_t k; } modes[] = {
        {"4 VALU, VGPR operands, no memory", step_kernel<4, false, false>},
        {"4 VALU, AGPR operands, no memory", step_kernel<4, true, false>},
        {"4 VALU, VGPR operands, + reads + pieces", step_kernel<4, false, true>},
        {"5 VALU, VGPR operands, + reads + pieces", step_kernel<5, false, true>},
        {"6 VALU, VGPR operands, + reads + pieces", step_kernel<6, false, true>},
        {"7 VALU, VGPR operands, + reads + pieces", step_kernel<7, false, true>},
        {"4 VALU, AGPR operands, + reads + pieces", step_kernel<4, true, true>},
        {"6 VALU, AGPR operands, + reads + pieces", step_kernel<6, true, true>},
    };
    const int steps = 48 * 400;
    printf("%-44s %12s %12s %10s   (matrix floor: 32 cycles per step)\n", "mode", "cyc/step med", "cyc/step max", "us total");
    for (int rep = 0; rep < 2; ++rep)
        for (auto& m : modes) {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(m.k, dim3(256), dim3(256), 0, 0, buf, steps, cyc, sink);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            std::vector<long long> c(256);
            CHECK(hipMemcpy(c.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            if (rep == 1)
                printf("%-44s %12.1f %12.1f %10.1f\n", m.name, double(c[128]) / steps, double(c[255]) / steps, ms * 1e3);
        }
    return 0;
}
