// How fast can ONE CU pull bytes -- from its XCD's L2 and from HBM -- and through which path?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fill_rate tools/ubench/fill_rate.hip && tools/ubench/fill_rate
// Every GEMM kernel of this library is, per CU, a fill stream next to a matrix stream: a 256 x 256 x 128 K block is 64 KiB of
// operands against 2048 matrix-pipe cycles, a decode-size weight stream is nothing but fill.  This probe measures the fill alone:
// `waves` waves per workgroup, one workgroup per CU, every wave keeps `depth` 1 KiB requests (64 lanes x 16 B) in flight, either as
// LDS-DMA (buffer_load_dwordx4 ... lds, what the GEMM kernels use) or as plain global loads into VGPRs (what the skinny kernel uses).
// Source: "l2" = every workgroup walks the same 1 MiB over and over (resident in each XCD's L2, far larger than the 32 KiB L1);
// "hbm" = every workgroup walks its own 8 MiB once (2 GiB in total: nothing is re-used).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(1024) void fill_lds_kernel(const uint8_t* base, size_t wg_stride, unsigned region_mask, int steps, int* sink) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint8_t* mine = base + static_cast<size_t>(blockIdx.x) * wg_stride;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(mine), 0, 0x7fffffff, 0x00020000);
    const unsigned slots = 128u / waves;                         // 1 KiB LDS slots of this wave
    for (int s = 0; s < steps; ++s) {
        const unsigned off = (static_cast<unsigned>(s * waves + wave) * 1024u) & region_mask;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (wave * slots + (s % slots)) * 1024), 16,
                                                 lane * 16, off, 0, 0);
        asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(DEPTH - 1) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink != nullptr && threadIdx.x == 0)
        sink[blockIdx.x] = reinterpret_cast<int*>(lds)[blockIdx.x & 1023];
}

template <int DEPTH>
__global__ __launch_bounds__(1024) void fill_vgpr_kernel(const uint8_t* base, size_t wg_stride, unsigned region_mask, int steps, int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint8_t* mine = base + static_cast<size_t>(blockIdx.x) * wg_stride + lane * 16;
    v4i r[DEPTH];
    v4i acc = {0, 0, 0, 0};
    #pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        r[i] = *reinterpret_cast<const v4i*>(mine + ((static_cast<unsigned>(i * waves + wave) * 1024u) & region_mask));
    for (int s = DEPTH; s < steps; s += DEPTH) {
        #pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            acc ^= r[i];
            r[i] = *reinterpret_cast<const v4i*>(mine + ((static_cast<unsigned>((s + i) * waves + wave) * 1024u) & region_mask));
        }
    }
    #pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        acc ^= r[i];
    if (sink != nullptr && (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678)
        sink[blockIdx.x] = 1;
}

// global -> VGPR -> ds_write_b128: the classic staging path (what a kernel without LDS-DMA does)
template <int DEPTH>
__global__ __launch_bounds__(1024) void fill_vgpr_lds_kernel(const uint8_t* base, size_t wg_stride, unsigned region_mask, int steps, int* sink) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint8_t* mine = base + static_cast<size_t>(blockIdx.x) * wg_stride + lane * 16;
    const unsigned slots = 128u / waves;
    v4i r[DEPTH];
    #pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        r[i] = *reinterpret_cast<const v4i*>(mine + ((static_cast<unsigned>(i * waves + wave) * 1024u) & region_mask));
    for (int s = DEPTH; s < steps; s += DEPTH) {
        #pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            *reinterpret_cast<v4i*>(lds + (wave * slots + ((s + i) % slots)) * 1024 + lane * 16) = r[i];
            r[i] = *reinterpret_cast<const v4i*>(mine + ((static_cast<unsigned>((s + i) * waves + wave) * 1024u) & region_mask));
        }
    }
    #pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        *reinterpret_cast<v4i*>(lds + (wave * slots + i) * 1024 + lane * 16) = r[i];
    __syncthreads();
    if (sink != nullptr && threadIdx.x == 0)
        sink[blockIdx.x] = reinterpret_cast<int*>(lds)[blockIdx.x & 1023];
}

// LDS-DMA one dword per lane (256 B per instruction): the pre-gfx950 width
template <int DEPTH>
__global__ __launch_bounds__(1024) void fill_lds_b32_kernel(const uint8_t* base, size_t wg_stride, unsigned region_mask, int steps, int* sink) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint8_t* mine = base + static_cast<size_t>(blockIdx.x) * wg_stride;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(mine), 0, 0x7fffffff, 0x00020000);
    const unsigned slots = 128u / waves;
    for (int s = 0; s < steps; ++s) {
        const unsigned off = (static_cast<unsigned>(s * waves + wave) * 1024u) & region_mask;
        #pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (wave * slots + (s % slots)) * 1024 + q * 256), 4,
                                                     lane * 4, off + q * 256, 0, 0);
        asm volatile("s_waitcnt vmcnt(%c0)" :: "i"(4 * DEPTH - 4) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink != nullptr && threadIdx.x == 0)
        sink[blockIdx.x] = reinterpret_cast<int*>(lds)[blockIdx.x & 1023];
}

// Does an LDS-DMA instruction hold up the wave that issued it?  One x4 LDS-DMA + NFMA independent v_fma per step.
template <int NFMA>
__global__ __launch_bounds__(1024) void fill_lds_valu_kernel(const uint8_t* base, size_t wg_stride, unsigned region_mask, int steps, int* sink) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint8_t* mine = base + static_cast<size_t>(blockIdx.x) * wg_stride;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(mine), 0, 0x7fffffff, 0x00020000);
    const unsigned slots = 128u / waves;
    float f[8];
    #pragma unroll
    for (int i = 0; i < 8; ++i)
        f[i] = static_cast<float>(lane + i);
    for (int s = 0; s < steps; ++s) {
        const unsigned off = (static_cast<unsigned>(s * waves + wave) * 1024u) & region_mask;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (wave * slots + (s % slots)) * 1024), 16,
                                                 lane * 16, off, 0, 0);
        #pragma unroll
        for (int i = 0; i < NFMA; ++i)
            asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i & 7]));
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float t = 0.f;
    #pragma unroll
    for (int i = 0; i < 8; ++i)
        t += f[i];
    if (sink != nullptr && threadIdx.x == 0)
        sink[blockIdx.x] = reinterpret_cast<int*>(lds)[blockIdx.x & 1023] + static_cast<int>(t);
}

// LDS-DMA next to a matrix stream: one x4 LDS-DMA + NMFMA register-resident v_mfma_f32_16x16x128_f8f6f4 per step (and, READS > 0, that
// many ds_read_b128 of 1 KiB each): what is left of the LDS-DMA rate when the matrix pipe and the LDS read port are busy?
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v4f_t __attribute__((ext_vector_type(4)));
template <int NMFMA, int READS>
__global__ __launch_bounds__(1024) void fill_lds_mfma_kernel(const uint8_t* base, size_t wg_stride, unsigned region_mask, int steps, int* sink) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint8_t* mine = base + static_cast<size_t>(blockIdx.x) * wg_stride;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(mine), 0, 0x7fffffff, 0x00020000);
    const unsigned slots = 128u / waves;
    v8i_t a = {lane, 1, 2, 3, 4, 5, 6, 7}, b = {7, 6, 5, 4, 3, 2, 1, lane};
    v4f_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    v4i rd = {0, 0, 0, 0};
    for (int s = 0; s < steps; ++s) {
        const unsigned off = (static_cast<unsigned>(s * waves + wave) * 1024u) & region_mask;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (wave * slots + (s % slots)) * 1024), 16,
                                                 lane * 16, off, 0, 0);
        #pragma unroll
        for (int i = 0; i < NMFMA; ++i)
            asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
        #pragma unroll
        for (int i = 0; i < READS; ++i)
            rd ^= *reinterpret_cast<const v4i*>(lds + ((wave * slots + ((s + 3 + i) % slots)) * 1024) + lane * 16);
        asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (sink != nullptr && threadIdx.x == 0)
        sink[blockIdx.x] = reinterpret_cast<int*>(lds)[blockIdx.x & 1023] + static_cast<int>(acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0]) + rd[0];
}

// Half the bytes as LDS-DMA, half as global -> VGPR -> ds_write_b128: do the two paths add up?
template <int DEPTH>
__global__ __launch_bounds__(1024) void fill_mixed_kernel(const uint8_t* base, size_t wg_stride, unsigned region_mask, int steps, int* sink) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[128 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const uint8_t* mine = base + static_cast<size_t>(blockIdx.x) * wg_stride;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(mine), 0, 0x7fffffff, 0x00020000);
    const unsigned slots = 128u / waves;
    v4i r[DEPTH];
    #pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        r[i] = *reinterpret_cast<const v4i*>(mine + lane * 16 + ((static_cast<unsigned>((2 * i) * waves + wave) * 1024u) & region_mask));
    for (int s = 2 * DEPTH; s < steps; s += 2 * DEPTH) {
        #pragma unroll
        for (int i = 0; i < DEPTH; ++i) {
            *reinterpret_cast<v4i*>(lds + (wave * slots + ((s + 2 * i) % slots)) * 1024 + lane * 16) = r[i];
            r[i] = *reinterpret_cast<const v4i*>(mine + lane * 16 + ((static_cast<unsigned>((s + 2 * i) * waves + wave) * 1024u) & region_mask));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds + (wave * slots + ((s + 2 * i + 1) % slots)) * 1024), 16,
                                                     lane * 16, (static_cast<unsigned>((s + 2 * i + 1) * waves + wave) * 1024u) & region_mask, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    #pragma unroll
    for (int i = 0; i < DEPTH; ++i)
        *reinterpret_cast<v4i*>(lds + (wave * slots + i) * 1024 + lane * 16) = r[i];
    __syncthreads();
    if (sink != nullptr && threadIdx.x == 0)
        sink[blockIdx.x] = reinterpret_cast<int*>(lds)[blockIdx.x & 1023];
}

typedef void (*kernel_t)(const uint8_t*, size_t, unsigned, int, int*);

static double run(kernel_t k, int wgs, int waves, const uint8_t* buf, size_t wg_stride, unsigned mask, int steps, int* sink) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k, dim3(wgs), dim3(waves * 64), 0, 0, buf, wg_stride, mask, steps, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k, dim3(wgs), dim3(waves * 64), 0, 0, buf, wg_stride, mask, steps, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    return static_cast<double>(wgs) * waves * steps * 1024.0 / (best * 1e-3) / 1e9;         // GB/s
}

int main() {
    const size_t total = 2ull << 30;
    uint8_t* buf;
    int* sink;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMemset(buf, 1, total));
    CHECK(hipMalloc(&sink, 4096 * 4));
    struct { const char* name; kernel_t k; int depth; } kernels[] = {
        {"lds-dma", fill_lds_kernel<4>, 4}, {"lds-dma", fill_lds_kernel<8>, 8}, {"lds-dma", fill_lds_kernel<16>, 16}, {"lds-dma", fill_lds_kernel<32>, 32},
        {"vgpr", fill_vgpr_kernel<4>, 4}, {"vgpr", fill_vgpr_kernel<8>, 8}, {"vgpr", fill_vgpr_kernel<16>, 16},
        {"vgpr+dsw", fill_vgpr_lds_kernel<4>, 4}, {"vgpr+dsw", fill_vgpr_lds_kernel<8>, 8}, {"vgpr+dsw", fill_vgpr_lds_kernel<16>, 16},
        {"dma-b32", fill_lds_b32_kernel<4>, 4}, {"dma-b32", fill_lds_b32_kernel<8>, 8},
        {"dma+0fma", fill_lds_valu_kernel<0>, 8}, {"dma+16fma", fill_lds_valu_kernel<16>, 8}, {"dma+32fma", fill_lds_valu_kernel<32>, 8},
        {"dma+64fma", fill_lds_valu_kernel<64>, 8},
        {"mixed", fill_mixed_kernel<4>, 8}, {"mixed", fill_mixed_kernel<8>, 16},
        {"dma+2mfma", fill_lds_mfma_kernel<2, 0>, 8}, {"dma+4mfma", fill_lds_mfma_kernel<4, 0>, 8}, {"dma+8mfma", fill_lds_mfma_kernel<8, 0>, 8},
        {"dma+4mfma+2rd", fill_lds_mfma_kernel<4, 2>, 8}, {"dma+4mfma+4rd", fill_lds_mfma_kernel<4, 4>, 8}, {"dma+0mfma+2rd", fill_lds_mfma_kernel<0, 2>, 8},
    };
    printf("%-14s %-4s %5s %5s %5s %10s %10s %12s\n", "path", "src", "wgs", "waves", "depth", "GB/s", "GB/s/CU", "KiB in flight/CU");
    for (const char* src : {"l2", "hbm"})
        for (int wgs : {256})
            for (int waves : {4, 8, 16})
                for (auto& kr : kernels) {
                    if (waves * kr.depth > 128)
                        continue;
                    const bool l2 = strcmp(src, "l2") == 0;
                    const size_t stride = l2 ? 0 : total / 256;
                    const unsigned mask = l2 ? (1u << 20) - 1 : static_cast<unsigned>(total / 256 - 1);
                    const int steps = l2 ? 65536 / waves : static_cast<int>(total / 256 / 1024 / waves);     // l2: 64 MiB per workgroup; hbm: its 8 MiB once
                    const double gbs = run(kr.k, wgs, waves, buf, stride, mask, steps, sink);
                    printf("%-14s %-4s %5d %5d %5d %10.0f %10.1f %12d\n", kr.name, src, wgs, waves, kr.depth, gbs, gbs / wgs, waves * kr.depth);
                }
    return 0;
}
