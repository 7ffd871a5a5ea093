// Matrix-pipe micro-benchmark for v_mfma_f32_16x16x128_f8f6f4 (FP8 x FP8) on gfx950: cycles per MFMA per SIMD and the
// shader clock under load, for 1 or 2 waves per SIMD, zero / random operand bits, with / without the promotion FMAs.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <string>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

// MODE 0: zero-C MFMA into a 4-deep ring, one token VALU per step (the abl4 stream)
// MODE 1: zero-C MFMA + 4 promotion FMAs per step (the abl5 stream)
// MODE 2: classic accumulate-in-place MFMA (C = D), 8 independent accumulators, no VALU
// MODE 3: zero-C MFMA into a ring, no VALU at all
template <int MODE, int PAD = 0, bool BAR = false>
__global__ __launch_bounds__(512) void mfma_rate_kernel(const int* __restrict__ src, float* __restrict__ out,
                                                        long long* __restrict__ cycles, int iters) {
    const int tid = threadIdx.x + blockIdx.x * blockDim.x;
    v8i a[4], b[2];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) a[i][j] = src[(tid * 67 + i * 8 + j) & 0xffff];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 8; ++j) b[i][j] = src[(tid * 131 + 4096 + i * 8 + j) & 0xffff];
    v4f part[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float acc[32];
    for (int i = 0; i < 32; ++i) acc[i] = 0.f;
    v4f accv[8];
    for (int i = 0; i < 8; ++i) accv[i] = v4f{0, 0, 0, 0};
    float scale = 1.0f + tid * 1e-9f;
    v16f big0, big1;
    for (int i = 0; i < 16; ++i) { big0[i] = 0.f; big1[i] = 0.f; }
    float dummy = 0.f;
    int sc_a = 127 + (tid & 1), sc_b = 126 + (tid & 3);
    float fill[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    typedef float v2f_ __attribute__((ext_vector_type(2)));
    v2f_ acc2[16];
    for (int i = 0; i < 16; ++i) acc2[i] = v2f_{0.f, 0.f};
    __syncthreads();
    const long long r0 = __builtin_amdgcn_s_memrealtime();      // constant 100 MHz counter: shader clock = memtime ticks / real time
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        #pragma unroll
        for (int i = 0; i < 32; ++i) {
            if constexpr (MODE == 0) {
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %2, %3, 0\n\tv_add_f32 %1, %1, %4"
                             : "=&v"(part[i & 3]), "+v"(acc[i & 7])
                             : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]), "v"(part[(i + 1) & 3][0]));
            } else if constexpr (MODE == 1) {
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %5, %6, 0\n\t"
                             "v_fmac_f32 %1, %7, %8\n\tv_fmac_f32 %2, %7, %9\n\tv_fmac_f32 %3, %7, %10\n\tv_fmac_f32 %4, %7, %11"
                             : "=&v"(part[i & 3]), "+v"(acc[(i * 4) & 31]), "+v"(acc[(i * 4 + 1) & 31]), "+v"(acc[(i * 4 + 2) & 31]),
                               "+v"(acc[(i * 4 + 3) & 31])
                             : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]), "v"(scale), "v"(part[(i + 1) & 3][0]),
                               "v"(part[(i + 1) & 3][1]), "v"(part[(i + 1) & 3][2]), "v"(part[(i + 1) & 3][3]));
            } else if constexpr (MODE == 4) {
                // the promote step, padded with PAD extra idle issue cycles, and a workgroup barrier every 16 steps
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %5, %6, 0\n\t"
                             "v_fmac_f32 %1, %7, %8\n\tv_fmac_f32 %2, %7, %9\n\tv_fmac_f32 %3, %7, %10\n\tv_fmac_f32 %4, %7, %11"
                             : "=&v"(part[i & 3]), "+v"(acc[(i * 4) & 31]), "+v"(acc[(i * 4 + 1) & 31]), "+v"(acc[(i * 4 + 2) & 31]),
                               "+v"(acc[(i * 4 + 3) & 31])
                             : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]), "v"(scale), "v"(part[(i + 1) & 3][0]),
                               "v"(part[(i + 1) & 3][1]), "v"(part[(i + 1) & 3][2]), "v"(part[(i + 1) & 3][3]));
                if constexpr (PAD > 0 && PAD <= 8) asm volatile("s_nop %c0" :: "i"(PAD - 1));
                if constexpr (PAD > 8) { asm volatile("s_nop 7"); asm volatile("s_nop %c0" :: "i"(PAD - 9)); }
                if (i % 16 == 15) __builtin_amdgcn_s_barrier();
            } else if constexpr (MODE == 5) {
                // role-split: 16-step MFMA+promote segments alternate with "load" segments (PAD x 32 idle cycles stand in
                // for the LDS / DMA work), a workgroup barrier between segments, waves 4-7 one segment behind waves 0-3
                if (i == 0 && it == 0 && threadIdx.x >= 256) __builtin_amdgcn_s_barrier();
                if (i % 16 == 0) {
                    for (int q = 0; q < PAD; ++q) asm volatile("s_nop 7");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_s_setprio(1);
                }
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %5, %6, 0\n\t"
                             "v_fmac_f32 %1, %7, %8\n\tv_fmac_f32 %2, %7, %9\n\tv_fmac_f32 %3, %7, %10\n\tv_fmac_f32 %4, %7, %11"
                             : "=&v"(part[i & 3]), "+v"(acc[(i * 4) & 31]), "+v"(acc[(i * 4 + 1) & 31]), "+v"(acc[(i * 4 + 2) & 31]),
                               "+v"(acc[(i * 4 + 3) & 31])
                             : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]), "v"(scale), "v"(part[(i + 1) & 3][0]),
                               "v"(part[(i + 1) & 3][1]), "v"(part[(i + 1) & 3][2]), "v"(part[(i + 1) & 3][3]));
                if (i % 16 == 15) {
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_s_barrier();
                }
            } else if constexpr (MODE == 6) {
                // one 32x32x128 tile step: MFMA (C = 0), MFMA (C = D), 16 promotion FMAs of the previous tile; PAD extra
                // single-slot instructions (v_mov) per tile step stand in for LDS / DMA / scalar work
                if (i % 4 == 0) {
                    v16f& pn = (i & 4) ? big1 : big0;
                    v16f& po = (i & 4) ? big0 : big1;
                    asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, 0" : "=&v"(pn) : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]));
                    #pragma unroll
                    for (int r = 0; r < 8; ++r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[r]) : "v"(scale), "v"(po[r]));
                    #pragma unroll
                    for (int r = 0; r < PAD / 2; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(dummy) : "v"(scale));
                    asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0" : "+v"(pn) : "v"(a[(i + 1) & 3]), "v"(b[(i >> 2) & 1]));
                    #pragma unroll
                    for (int r = 8; r < 16; ++r) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[r]) : "v"(scale), "v"(po[r]));
                    #pragma unroll
                    for (int r = 0; r < PAD - PAD / 2; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(dummy) : "v"(scale));
                    if (BAR && i % 16 == 12) __builtin_amdgcn_s_barrier();
                }
            } else if constexpr (MODE == 7) {
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %5, %6, 0\n\t"
                             "v_fmac_f32 %1, %7, %8\n\tv_fmac_f32 %2, %7, %9\n\tv_fmac_f32 %3, %7, %10\n\tv_fmac_f32 %4, %7, %11"
                             : "=&v"(part[i & 3]), "+v"(acc[(i * 4) & 31]), "+v"(acc[(i * 4 + 1) & 31]), "+v"(acc[(i * 4 + 2) & 31]),
                               "+v"(acc[(i * 4 + 3) & 31])
                             : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]), "v"(scale), "v"(part[(i + 1) & 3][0]),
                               "v"(part[(i + 1) & 3][1]), "v"(part[(i + 1) & 3][2]), "v"(part[(i + 1) & 3][3]));
                #pragma unroll
                for (int r = 0; r < (PAD + (i & 3)) / 4; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(dummy) : "v"(scale));
                if (BAR && i % 16 == 15) __builtin_amdgcn_s_barrier();
            } else if constexpr (MODE == 8) {
                // promotion with packed FMAs: 2 v_pk_fma_f32 per step instead of 4 v_fmac; PAD/4 independent filler VALU
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f sc2 = {scale, scale};
                const v4f po = part[(i + 1) & 3];
                const v2f p01 = {po[0], po[1]}, p23 = {po[2], po[3]};
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %3, %4, 0\n\t"
                             "v_pk_fma_f32 %1, %5, %6, %1\n\tv_pk_fma_f32 %2, %5, %7, %2"
                             : "=&v"(part[i & 3]), "+v"(acc2[(i * 2) & 15]), "+v"(acc2[(i * 2 + 1) & 15])
                             : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]), "v"(sc2), "v"(p01), "v"(p23));
                #pragma unroll
                for (int r = 0; r < (PAD + (i & 3)) / 4; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(fill[(i + r) & 7]) : "v"(scale));
                if (BAR && i % 16 == 15) __builtin_amdgcn_s_barrier();
            } else if constexpr (MODE == 9) {
                // as MODE 7 but the fillers write 8 different registers (no write-after-write chain)
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %5, %6, 0\n\t"
                             "v_fmac_f32 %1, %7, %8\n\tv_fmac_f32 %2, %7, %9\n\tv_fmac_f32 %3, %7, %10\n\tv_fmac_f32 %4, %7, %11"
                             : "=&v"(part[i & 3]), "+v"(acc[(i * 4) & 31]), "+v"(acc[(i * 4 + 1) & 31]), "+v"(acc[(i * 4 + 2) & 31]),
                               "+v"(acc[(i * 4 + 3) & 31])
                             : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]), "v"(scale), "v"(part[(i + 1) & 3][0]),
                               "v"(part[(i + 1) & 3][1]), "v"(part[(i + 1) & 3][2]), "v"(part[(i + 1) & 3][3]));
                #pragma unroll
                for (int r = 0; r < (PAD + (i & 3)) / 4; ++r) asm volatile("v_mov_b32 %0, %1" : "=v"(fill[(i + r) & 7]) : "v"(scale));
                if (BAR && i % 16 == 15) __builtin_amdgcn_s_barrier();
            } else if constexpr (MODE == 10) {
                // hardware-scaled form, scales from VGPRs, accumulate in place (8 independent accumulators)
                accv[i & 7] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[i & 3], b[(i >> 2) & 1], accv[i & 7], 0, 0, 0,
                                                                             sc_a, 0, sc_b);
            } else if constexpr (MODE == 2) {
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0"
                             : "+v"(accv[i & 7]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]));
            } else {
                asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0"
                             : "=&v"(part[i & 3]) : "v"(a[i & 3]), "v"(b[(i >> 2) & 1]));
            }
        }
    }
    if constexpr (MODE == 5) { if (threadIdx.x < 256) __builtin_amdgcn_s_barrier(); }   // balance the stagger barrier
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long r1 = __builtin_amdgcn_s_memrealtime();
    float r = 0.f;
    for (int i = 0; i < 32; ++i) r += acc[i];
    for (int i = 0; i < 4; ++i) r += part[i][0] + part[i][1] + part[i][2] + part[i][3];
    for (int i = 0; i < 8; ++i) r += accv[i][0] + accv[i][3];
    for (int i = 0; i < 16; ++i) r += big0[i] + big1[i];
    r += dummy;
    for (int i = 0; i < 8; ++i) r += fill[i];
    for (int i = 0; i < 16; ++i) r += acc2[i][0] + acc2[i][1];
    out[tid] = r;
    if ((threadIdx.x & 63) == 0) {
        cycles[tid >> 6] = t1 - t0;
        cycles[4096 + (tid >> 6)] = r1 - r0;
    }
}

template <int MODE, int PAD = 0, bool BAR = false>
void run(const char* name, int threads, const int* src, float* out, long long* cyc, int iters) {
    const int blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_rate_kernel<MODE, PAD, BAR>), dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const int waves = blocks * threads / 64;
    std::vector<long long> h(waves), hr(waves);
    hipMemcpy(h.data(), cyc, waves * sizeof(long long), hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), cyc + 4096, waves * sizeof(long long), hipMemcpyDeviceToHost);
    double mean = 0, mx = 0, mn = 1e30, clk = 0;
    for (auto c : h) { mean += c; mx = c > mx ? c : mx; mn = c < mn ? c : mn; }
    for (int w = 0; w < waves; ++w) clk += static_cast<double>(h[w]) / static_cast<double>(hr[w]) * 100.0;     // MHz per wave
    clk /= waves;
    mean /= waves;
    const double mfma_per_simd = 32.0 * iters * (threads / 256);
    const double flops = 2.0 * 16 * 16 * 128 * 32.0 * iters * waves;
    printf("%-46s waves/SIMD=%d  wall=%8.1f us  ticks/wave min %9.0f mean %9.0f max %9.0f  max-ticks per MFMA per SIMD=%6.2f  clock~%6.1f MHz (ticks/wall) %6.1f MHz (s_memtime/s_memrealtime)  %7.1f TFLOPS\n",
           name, threads / 256, ms * 1e3, mn, mean, mx, mx / mfma_per_simd, mx / (ms * 1e3), clk, flops / (ms * 1e-3) / 1e12);
    fflush(stdout);
}

// `mfma_rate ceiling [file]`: the table behind the "what can this part sustain" question (profiles/r03_ceiling): register-resident
// MFMA streams with zero / uniform-random / reference-quantised operand bytes (file = 256 KiB of per_token_cast_to_fp8 output,
// tools/ceiling.py writes it), 1 and 2 waves per SIMD, long enough (iters) that launch overhead does not matter.
int ceiling_table(const char* data_file) {
    const int n = 1 << 16;
    std::vector<int> rnd(n), zer(n, 0), quant;
    srand(1);
    for (auto& x : rnd) {
        unsigned v = 0;
        for (int b = 0; b < 4; ++b) { unsigned byte = rand() & 0xff; if ((byte & 0x7f) == 0x7f) byte ^= 1; v |= byte << (8 * b); }
        x = (int)v;
    }
    if (data_file != nullptr) {
        FILE* f = fopen(data_file, "rb");
        if (f != nullptr) {
            quant.resize(n);
            if (fread(quant.data(), 4, n, f) != (size_t)n) quant.clear();
            fclose(f);
        }
        if (quant.empty()) printf("(could not read %d bytes from %s: the reference-quantised rows are skipped)\n", n * 4, data_file);
    }
    int *d_rnd, *d_zer, *d_q = nullptr; float* out; long long* cyc;
    hipMalloc(&d_rnd, n * 4); hipMalloc(&d_zer, n * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2 * 4096 * 8);
    hipMemcpy(d_rnd, rnd.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_zer, zer.data(), n * 4, hipMemcpyHostToDevice);
    if (!quant.empty()) { hipMalloc(&d_q, n * 4); hipMemcpy(d_q, quant.data(), n * 4, hipMemcpyHostToDevice); }
    const int iters = 8000;          // 256 k MFMAs per wave: 3.5 - 5 ms per launch
    run<3>("warm-up", 512, d_rnd, out, cyc, iters);
    run<3>("warm-up", 512, d_rnd, out, cyc, iters);
    struct Fill { const char* name; const int* ptr; } fills[3] = {{"zeros", d_zer}, {"uniform random bytes", d_rnd}, {"reference-quantised bytes", d_q}};
    for (const Fill& f : fills) {
        if (f.ptr == nullptr) continue;
        printf("--- operand fill: %s\n", f.name);
        for (int threads : {256, 512}) {
            run<2>("unscaled MFMA, accumulate in place", threads, f.ptr, out, cyc, iters);
            run<10>("scaled MFMA (UE8M0), accumulate in place", threads, f.ptr, out, cyc, iters);
            run<3>("zero-C MFMA ring, no VALU", threads, f.ptr, out, cyc, iters);
            run<1>("zero-C MFMA + 4 promotion FMAs per MFMA", threads, f.ptr, out, cyc, iters);
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "ceiling")
        return ceiling_table(argc > 2 ? argv[2] : nullptr);
    const int n = 1 << 16;
    std::vector<int> rnd(n), zer(n, 0);
    srand(1);
    for (auto& x : rnd) {
        // random FP8 e4m3 bytes without NaN encodings (0x7f / 0xff)
        unsigned v = 0;
        for (int b = 0; b < 4; ++b) { unsigned byte = rand() & 0xff; if ((byte & 0x7f) == 0x7f) byte ^= 1; v |= byte << (8 * b); }
        x = (int)v;
    }
    int *d_rnd, *d_zer; float* out; long long* cyc;
    hipMalloc(&d_rnd, n * 4); hipMalloc(&d_zer, n * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 2 * 4096 * 8);
    hipMemcpy(d_rnd, rnd.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_zer, zer.data(), n * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    run<3>("warm-up", 512, d_rnd, out, cyc, iters);
    run<10>("scaled MFMA, accumulate in place, 1 wave", 256, d_rnd, out, cyc, iters);
    run<10>("scaled MFMA, accumulate in place, 2 waves", 512, d_rnd, out, cyc, iters);
    run<2>("unscaled MFMA, accumulate in place, 2 waves", 512, d_rnd, out, cyc, iters);
    return 0;
    run<9, 0, true>("fmac x4, fill 0, barrier/16", 512, d_rnd, out, cyc, iters);
    run<9, 8, true>("fmac x4, fill 2/step indep, barrier/16", 512, d_rnd, out, cyc, iters);
    run<9, 16, true>("fmac x4, fill 4/step indep, barrier/16", 512, d_rnd, out, cyc, iters);
    run<9, 24, true>("fmac x4, fill 6/step indep, barrier/16", 512, d_rnd, out, cyc, iters);
    run<8, 0, true>("pk_fma x2, fill 0, barrier/16", 512, d_rnd, out, cyc, iters);
    run<8, 8, true>("pk_fma x2, fill 2/step indep, barrier/16", 512, d_rnd, out, cyc, iters);
    run<8, 16, true>("pk_fma x2, fill 4/step indep, barrier/16", 512, d_rnd, out, cyc, iters);
    run<8, 24, true>("pk_fma x2, fill 6/step indep, barrier/16", 512, d_rnd, out, cyc, iters);
    run<9, 16>("fmac x4, fill 4/step indep, 1 wave", 256, d_rnd, out, cyc, iters);
    run<8, 16>("pk_fma x2, fill 4/step indep, 1 wave", 256, d_rnd, out, cyc, iters);
    return 0;
    run<6, 0>("32x32x64 pair+16fmac, fill 0", 256, d_rnd, out, cyc, iters);
    run<6, 8>("32x32x64 pair+16fmac, fill 8", 256, d_rnd, out, cyc, iters);
    run<6, 12>("32x32x64 pair+16fmac, fill 12", 256, d_rnd, out, cyc, iters);
    run<6, 16>("32x32x64 pair+16fmac, fill 16", 256, d_rnd, out, cyc, iters);
    run<7, 0>("16x16x128 +4fmac, fill 0/4", 256, d_rnd, out, cyc, iters);
    run<7, 8>("16x16x128 +4fmac, fill 8/4", 256, d_rnd, out, cyc, iters);
    run<7, 12>("16x16x128 +4fmac, fill 12/4", 256, d_rnd, out, cyc, iters);
    run<7, 16>("16x16x128 +4fmac, fill 16/4", 256, d_rnd, out, cyc, iters);
    run<6, 0>("32x32x64 pair+16fmac, fill 0", 512, d_rnd, out, cyc, iters);
    run<6, 8>("32x32x64 pair+16fmac, fill 8", 512, d_rnd, out, cyc, iters);
    run<6, 16>("32x32x64 pair+16fmac, fill 16", 512, d_rnd, out, cyc, iters);
    run<6, 24>("32x32x64 pair+16fmac, fill 24", 512, d_rnd, out, cyc, iters);
    run<6, 8, true>("32x32x64 fill 8 + barrier/4 tiles", 512, d_rnd, out, cyc, iters);
    run<6, 16, true>("32x32x64 fill 16 + barrier/4 tiles", 512, d_rnd, out, cyc, iters);
    run<7, 8>("16x16x128 +4fmac, fill 8/4", 512, d_rnd, out, cyc, iters);
    run<7, 16>("16x16x128 +4fmac, fill 16/4", 512, d_rnd, out, cyc, iters);
    run<7, 8, true>("16x16x128 fill 8/4 + barrier/16", 512, d_rnd, out, cyc, iters);
    run<7, 16, true>("16x16x128 fill 16/4 + barrier/16", 512, d_rnd, out, cyc, iters);
    return 0;
    run<5, 0>("role-split 16/seg, load 0", 512, d_rnd, out, cyc, iters);
    run<5, 4>("role-split 16/seg, load 128cyc", 512, d_rnd, out, cyc, iters);
    run<5, 8>("role-split 16/seg, load 256cyc", 512, d_rnd, out, cyc, iters);
    run<5, 12>("role-split 16/seg, load 384cyc", 512, d_rnd, out, cyc, iters);
    run<5, 16>("role-split 16/seg, load 512cyc", 512, d_rnd, out, cyc, iters);
    run<4, 0>("promote+barrier/16, pad 0", 512, d_rnd, out, cyc, iters);
    run<4, 2>("promote+barrier/16, pad 2", 512, d_rnd, out, cyc, iters);
    run<4, 4>("promote+barrier/16, pad 4", 512, d_rnd, out, cyc, iters);
    run<4, 8>("promote+barrier/16, pad 8", 512, d_rnd, out, cyc, iters);
    run<4, 12>("promote+barrier/16, pad 12", 512, d_rnd, out, cyc, iters);
    run<4, 16>("promote+barrier/16, pad 16", 512, d_rnd, out, cyc, iters);
    run<4, 20>("promote+barrier/16, pad 20", 512, d_rnd, out, cyc, iters);
    run<4, 24>("promote+barrier/16, pad 24", 512, d_rnd, out, cyc, iters);
    run<4, 0>("promote+barrier/16 1w, pad 0", 256, d_rnd, out, cyc, iters);
    run<4, 8>("promote+barrier/16 1w, pad 8", 256, d_rnd, out, cyc, iters);
    run<4, 16>("promote+barrier/16 1w, pad 16", 256, d_rnd, out, cyc, iters);
    for (int threads : {256, 512}) {
        run<3>("bare zero-C ring, zeros", threads, d_zer, out, cyc, iters);
        run<3>("bare zero-C ring, random", threads, d_rnd, out, cyc, iters);
        run<2>("accumulate in place, random", threads, d_rnd, out, cyc, iters);
        run<0>("zero-C + 1 VALU, random", threads, d_rnd, out, cyc, iters);
        run<1>("zero-C + 4 FMA promote, random", threads, d_rnd, out, cyc, iters);
        run<1>("zero-C + 4 FMA promote, zeros", threads, d_zer, out, cyc, iters);
    }
    return 0;
}
