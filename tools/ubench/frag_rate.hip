// Round 4 probe for VERDICT item 1(b): "keep one operand out of the LDS-DMA path" -- is it worth building?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/frag_rate tools/ubench/frag_rate.hip && tools/ubench/frag_rate
// One workgroup per CU, 4 waves (one per SIMD), every wave runs the per-quarter-K-block mix of a 256 x 256 x 128 tile with
// 128 x 128 (2 x 2 waves) or 64 x 256 (4 x 1 waves) wave tiles: 16 scaled-shape MFMAs (v_mfma_f32_16x16x128_f8f6f4, register
// operands) + 8 ds_read_b128 of 1 KiB + the step's share of the operand fill, in one of these forms:
//   all-dma   : 4 LDS-DMA pieces (buffer_load_dwordx4 ... lds, 1 KiB each)          -- what e8_quad / hipBLASLt do (16 per K block)
//   a-direct  : 2 LDS-DMA pieces + ONE fragment-shaped global load pair (16 rows x 128 B, row pitch 7168: lane (r, g) pulls bytes
//               16 g .. and 64 + 16 g .. of row r) that feeds the step's MFMAs directly        -- the 4 x 1 wave layout of item 1(b)
//   a-direct2 : as a-direct with TWO fragment pairs per step (the 2 x 2 wave layout: every A fragment is pulled by two waves)
//   no-fill   : no fill at all (matrix stream + fragment reads: the floor)
//   dma-m0    : all-dma with the four pieces sharing ONE M0 value (instruction offsets 0 / 1024 / 2048 / 3072)
// Source: 256 rows x 7168 bytes per workgroup-independent matrix (1.75 MiB: L2-resident), walked K block by K block.
// Output: cycles per step (s_memtime, wave 0 of workgroup 0 and the median over workgroups) against the 512-cycle matrix floor.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int PITCH = 7168, ROWS = 256, KBS = PITCH / 128;

// MODE 0 all-dma, 1 a-direct, 2 a-direct2, 3 no-fill, 4 dma-m0
template <int MODE>
__global__ __launch_bounds__(256) void step_kernel(const uint8_t* base, int steps, long long* cycles, int* sink) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, ROWS * PITCH + 8192, 0x00020000);
    // LDS-DMA piece: 8 rows x 128 bytes (lane: row l >> 3, chunk l & 7, swizzled like the kernels' pieces)
    const int piece_voff = (lane >> 3) * PITCH + (((lane & 7) ^ (lane >> 3)) << 4);
    // fragment-shaped load: row l & 15, bytes 16 (l >> 4) and 64 + 16 (l >> 4)
    const int frag_voff = (lane & 15) * PITCH + ((lane >> 4) << 4);
    v8i bfrag = {lane, 1, 2, 3, 4, 5, 6, 7};
    v4i rlo[4], rhi[4], xlo[4], xhi[4];
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        rlo[i] = v4i{i, lane, 2, 3}; rhi[i] = v4i{4, 5, 6, 7}; xlo[i] = rlo[i]; xhi[i] = rhi[i];
    }
    v4f acc[16];
    #pragma unroll
    for (int i = 0; i < 16; ++i)
        acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v4i rd = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; s += 4) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int step = s + u;
            const int kb = (step >> 2) % KBS;                   // four steps = one K block of the wave's work
            const int lds_slot = ((step & 15) * 4 + wave) * 4096;            // 16 x 4 x 4 KiB = the whole 128 KiB... (wraps)
            // ---- fill ----
            if constexpr (MODE == 0 || MODE == 4) {
                auto piece = [&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const int rows = ((u * 4 + q) * 4 + wave) * 8 % ROWS;
                    if constexpr (MODE == 4)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (lds_slot & 0x1ffff)), 16,
                                                                 piece_voff + rows * PITCH - q * 1024 + 4096, kb * 128, q * 1024, 0);
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + ((lds_slot + q * 1024) & 0x1ffff)), 16,
                                                                 piece_voff + rows * PITCH, kb * 128, 0, 0);
                };
                piece(std::integral_constant<int, 0>{}); piece(std::integral_constant<int, 1>{});
                piece(std::integral_constant<int, 2>{}); piece(std::integral_constant<int, 3>{});
            } else if constexpr (MODE == 1 || MODE == 2) {
                #pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int rows = ((u * 2 + q) * 4 + wave) * 8 % ROWS;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + ((lds_slot + q * 1024) & 0x1ffff)), 16,
                                                             piece_voff + rows * PITCH, kb * 128, 0, 0);
                }
                // the fragment that the MFMAs of step + 3 consume: ring of four, three steps (~1.5 k cycles) of lead
                {
                    const int rows = ((wave * 4 + u) * 16) % ROWS;
                    asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen\n\t"
                                 "buffer_load_dwordx4 %1, %2, %3, %4 offen offset:64"
                                 : "=&v"(rlo[(u + 3) & 3]), "=&v"(rhi[(u + 3) & 3]) : "v"(frag_voff + rows * PITCH), "s"(rsrc), "s"(kb * 128) : "memory");
                }
                if constexpr (MODE == 2) {      // (the 2 x 2 layout: a second wave pulls the same rows -- here: another fragment, pulled and dropped)
                    const int rows = ((wave * 4 + u) * 16 + 128) % ROWS;
                    asm volatile("buffer_load_dwordx4 %0, %2, %3, %4 offen\n\t"
                                 "buffer_load_dwordx4 %1, %2, %3, %4 offen offset:64"
                                 : "=&v"(xlo[(u + 3) & 3]), "=&v"(xhi[(u + 3) & 3]) : "v"(frag_voff + rows * PITCH), "s"(rsrc), "s"(kb * 128) : "memory");
                }
                // the loads of step - 3 have landed: at most three steps' worth of younger operations outstanding
                asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(3 * (2 + (MODE == 2 ? 4 : 2))) : "memory");
                asm volatile("" : "+v"(rlo[u & 3]), "+v"(rhi[u & 3]));
                if constexpr (MODE == 2) {
                    asm volatile("" : "+v"(xlo[u & 3]), "+v"(xhi[u & 3]));
                    rd ^= xlo[u & 3] ^ xhi[u & 3];
                }
            }
            // ---- matrix stream + fragment reads ----
            #pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(__builtin_shufflevector(rlo[u & 3], rhi[u & 3], 0, 1, 2, 3, 4, 5, 6, 7)), "v"(bfrag));
                if (i & 1)
                    rd ^= *reinterpret_cast<const v4i*>(lds + (((step + i) & 127) * 1024) + lane * 16);
            }
        }
        if constexpr (MODE == 0 || MODE == 4)
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    float t = 0.f;
    #pragma unroll
    for (int i = 0; i < 16; ++i)
        t += acc[i][0];
    if (threadIdx.x == 0)
        cycles[blockIdx.x] = t1 - t0;
    if (sink != nullptr && t == 12345.f)
        sink[threadIdx.x] = rd[0] + rd[1] + rd[2] + rd[3] + reinterpret_cast<int*>(lds)[lane];
}

typedef void (*kernel_t)(const uint8_t*, int, long long*, int*);

int main() {
    uint8_t* buf;
    long long* cyc;
    int* sink;
    CHECK(hipMalloc(&buf, ROWS * PITCH + 8192));
    std::vector<uint8_t> host(ROWS * PITCH);
    srand(1);
    for (auto& b : host) b = static_cast<uint8_t>(rand() & 0x7f) % 0x7e;       // e4m3 bytes, no NaN
    CHECK(hipMemcpy(buf, host.data(), host.size(), hipMemcpyHostToDevice));
    CHECK(hipMalloc(&cyc, 256 * 8));
    CHECK(hipMalloc(&sink, 4096));
    struct { const char* name; kernel_t k; } modes[] = {
        {"no-fill", step_kernel<3>}, {"all-dma", step_kernel<0>}, {"dma-m0", step_kernel<4>}, {"a-direct", step_kernel<1>}, {"a-direct2", step_kernel<2>},
    };
    const int steps = 4096;
    printf("%-10s %12s %12s %10s   (matrix floor: 512 cycles per step; 4 steps = one K block of a 256 x 256 tile)\n", "mode", "cyc/step med", "cyc/step max", "us total");
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
                printf("%-10s %12.1f %12.1f %10.1f\n", m.name, double(c[128]) / steps, double(c[255]) / steps, ms * 1e3);
        }
    return 0;
}
