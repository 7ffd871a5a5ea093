// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 scale-operand semantics on gfx950 (which lane's scale byte applies to
// which row / 32-K group, byte selection by opsel, E8M0 encoding).  A = B = all 1.0 (e4m3 0x38): unscaled D = 128.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
    v4f c = {0.f, 0.f, 0.f, 0.f};
    int sa = 127, sb = 127;
    if (MODE == 1) sa = 127 + (lane & 15);                 // slot-A scale varies with the lane's row index
    if (MODE == 2) sa = 127 + (lane >> 4);                 // ... with the lane's K group
    if (MODE == 3) sb = 127 + (lane & 15);                 // slot-B scale varies with the lane's row (= D column) index
    if (MODE == 4) sb = 127 + (lane >> 4);
    if (MODE == 5) sa = (127) | (128 << 8) | (129 << 16) | (130 << 24);   // byte select test, opsel_a = 2 below
    if (MODE == 6) sa = 126;                                               // 0.5
    v4f d;
    if (MODE == 5) d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, sa, 0, sb);
    else d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    // D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
    for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = d[r];
}

template <int MODE> void run(const char* what) {
    float* d_out; hipMalloc(&d_out, 256 * 4);
    hipLaunchKernelGGL(probe<MODE>, dim3(1), dim3(64), 0, 0, d_out);
    std::vector<float> h(256); hipMemcpy(h.data(), d_out, 256 * 4, hipMemcpyDeviceToHost);
    printf("%s\n  row 0..15 of column 0: ", what);
    for (int i = 0; i < 16; ++i) printf("%g ", h[i * 16]);
    printf("\n  column 0..15 of row 0: ");
    for (int j = 0; j < 16; ++j) printf("%g ", h[j]);
    printf("\n");
    hipFree(d_out);
}
int main() {
    run<0>("unit scales (expect 128 everywhere)");
    run<1>("slot-A scale = 2^(lane&15)");
    run<2>("slot-A scale = 2^(lane>>4)");
    run<3>("slot-B scale = 2^(lane&15)");
    run<4>("slot-B scale = 2^(lane>>4)");
    run<5>("slot-A scale VGPR bytes {1,2,4,8}, opsel_a = 2 (expect x4 if byte 2)");
    run<6>("slot-A scale 126 (expect 64)");
    return 0;
}
