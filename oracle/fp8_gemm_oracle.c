/*
 * CPU oracle for the FP8 blockwise-scaled GEMM hot path  --  TEST INFRASTRUCTURE ONLY.
 *
 * This file restates, in plain C, the arithmetic of the reference's FP8 GEMM kernels so that
 * the HIP kernels can be checked against it.  Nothing in the product path (deepgemm_amd/) may
 * call, link or import it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Reference behaviour restated here (paths relative to /root/reference):
 *   - per-element arithmetic: D[m,n] = cast( sum_kb (sfa[m,kb] * sfb[n/gran_n,kb]) * (sum_{k in kb} Aq[m,k]*Bq[n,k]) )
 *     inner sum = tensor-core FP32 accumulate over one 128-K block, scale product formed first in FP32,
 *     outer sum = FP32 running sum in k-block order, ONE round-to-nearest-even cast at the end
 *       deep_gemm/include/deep_gemm/impls/sm90_fp8_gemm_1d2d.cuh:283-347 (promotion), :416-418 (bf16 cast)
 *       deep_gemm/include/deep_gemm/impls/sm90_fp8_gemm_1d1d.cuh:303-311 (per-column SFB form)
 *   - accumulation (c given): D_mem = D_mem + cast(result), done in D's dtype memory (TMA reduce-add)
 *       csrc/apis/gemm.hpp:35-45, deep_gemm/include/deep_gemm/epilogue/sm100_store_cd.cuh:121-129,
 *       deep_gemm/include/deep_gemm/impls/sm90_fp8_gemm_1d1d.cuh:333-335
 *   - M-grouped contiguous: row block -> group = grouped_layout[first row of block], negative => zeros
 *       deep_gemm/include/deep_gemm/scheduler/gemm.cuh:160-162, :311-319; deep_gemm/legacy/m_grouped_gemm.py:35-40
 *   - M-grouped masked: only rows < masked_m[g] are defined
 *       deep_gemm/include/deep_gemm/scheduler/gemm.cuh:200-216, tests/test_fp8_fp4.py:166-174
 *   - SF transpose into the MN-major, 16-byte-aligned layout
 *       csrc/jit_kernels/impls/smxx_layout.hpp:120-153, deep_gemm/include/deep_gemm/impls/smxx_layout.cuh:12-50
 *
 * The inner 128-K dot product is evaluated in double and rounded once to float: every e4m3*e4m3
 * product is exact in float and the tensor core's internal summation order is unspecified, so the
 * correctly rounded block sum is the natural fixed point for a tolerance-based comparison.
 *
 * Parity pinning: see oracle/README.md (the reference's CUDA kernel cannot run here; the oracle is
 * pinned against fixtures generated with the reference's own Python quantisers / test expression).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#define DGO_BLOCK_K 128

/* Scale granularity along K: 128 (the SM90 recipes) or 32 (the reference's SM100 MX recipe for FP8 x FP8: csrc/apis/gemm.hpp:311-312,
 * csrc/apis/layout.hpp:48-58, per_token_cast_to_fp8(..., gran_k=32) in deep_gemm/utils/math.py:26-38).  The arithmetic is the same statement
 * with 32-K blocks: D = cast( sum_kb (sfa[m,kb] * sfb[n,kb]) * sum_{k in 32-block kb} Aq Bq ); the scale tensors then hold ceil(K / 32) columns.
 * A process-wide setting of the checker (test infrastructure), set around a call by oracle.py. */
static int g_gran_k = DGO_BLOCK_K;
static int g_grouped_gran_n = 128;        /* rows of B per SFB row in the grouped entries: 128 (block scales) or 1 (per-row scales: the packed-UE8M0 input format) */
int dgo_set_gran_k(int gran_k) {
    if (gran_k != 32 && gran_k != 128)
        return 1;
    g_gran_k = gran_k;
    return 0;
}
int dgo_set_grouped_gran_n(int gran_n) {
    if (gran_n != 1 && gran_n != 32 && gran_n != 128)
        return 1;
    g_grouped_gran_n = gran_n;
    return 0;
}

static float g_e4m3_lut[256];
static int g_lut_ready = 0;

/* OCP e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa; no inf; S.1111.111 = NaN. */
float dgo_e4m3_to_f32(uint8_t v) {
    const int sign = v >> 7, exp = (v >> 3) & 0xF, man = v & 0x7;
    float x;
    if (exp == 0xF && man == 0x7)
        x = NAN;
    else if (exp == 0)
        x = ldexpf((float) man, -9);               /* subnormal: man * 2^-3 * 2^-6 */
    else
        x = ldexpf(1.0f + (float) man / 8.0f, exp - 7);
    return sign ? -x : x;
}

static void dgo_init_lut(void) {
    if (g_lut_ready)
        return;
    for (int i = 0; i < 256; ++i)
        g_e4m3_lut[i] = dgo_e4m3_to_f32((uint8_t) i);
    g_lut_ready = 1;
}

/* float -> bf16 bits, round to nearest even (what __float2bfloat16_rn / torch .to(bfloat16) do). */
uint16_t dgo_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u)            /* NaN stays NaN (quiet) */
        return (uint16_t) ((u >> 16) | 0x0040u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t) (u >> 16);
}

float dgo_bf16_to_f32(uint16_t h) {
    uint32_t u = ((uint32_t) h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* Store one result element with the reference's cast / reduce-add semantics. d_dtype: 0 = bf16, 1 = fp32. */
static void dgo_store(void* d, int64_t idx, int d_dtype, int accumulate, float value) {
    if (d_dtype == 1) {
        float* p = (float*) d + idx;
        *p = accumulate ? (*p + value) : value;
    } else {
        uint16_t* p = (uint16_t*) d + idx;
        const uint16_t r = dgo_f32_to_bf16(value);
        *p = accumulate ? dgo_f32_to_bf16(dgo_bf16_to_f32(*p) + dgo_bf16_to_f32(r)) : r;
    }
}

/*
 * One row-range of D = [C +] A @ B^T with blockwise scales.  All strides in elements.
 *   a   : e4m3 bytes, element (m,k) at a[m*a_sm + k*a_sk]
 *   sfa : FP32, element (m,kb) at sfa[m*sfa_sm + kb*sfa_sk]           (1 x 128 granularity)
 *   b   : e4m3 bytes, element (n,k) at b[n*b_sn + k*b_sk]
 *   sfb : FP32, element (n/gran_n,kb) at sfb[(n/gran_n)*sfb_sn + kb*sfb_sk], gran_n in {1,128}
 *   d   : row-major, row stride d_sm, accumulate => reduce-add onto the existing contents
 * Rows [m_begin, m_end) of A/D are computed; `zero_rows` forces zeros instead (contiguous padding).
 */
static void dgo_rows(const uint8_t* a, int64_t a_sm, int64_t a_sk,
                     const float* sfa, int64_t sfa_sm, int64_t sfa_sk,
                     const uint8_t* b, int64_t b_sn, int64_t b_sk,
                     const float* sfb, int64_t sfb_sn, int64_t sfb_sk, int gran_n,
                     void* d, int64_t d_sm, int d_dtype, int accumulate,
                     int m_begin, int m_end, int n, int k, int zero_rows) {
    dgo_init_lut();
    const int block_k = g_gran_k;
    const int num_kb = (k + block_k - 1) / block_k;
    if (zero_rows) {
        for (int m = m_begin; m < m_end; ++m)
            for (int j = 0; j < n; ++j)
                dgo_store(d, (int64_t) m * d_sm + j, d_dtype, 0, 0.0f);
        return;
    }

    /* Decode B once (n x k floats). */
    float* bf = (float*) malloc(sizeof(float) * (size_t) n * (size_t) k);
    for (int j = 0; j < n; ++j)
        for (int kk = 0; kk < k; ++kk)
            bf[(size_t) j * k + kk] = g_e4m3_lut[b[j * b_sn + kk * b_sk]];

    #pragma omp parallel
    {
        float* af = (float*) malloc(sizeof(float) * (size_t) k);
        #pragma omp for schedule(dynamic, 4)
        for (int m = m_begin; m < m_end; ++m) {
            for (int kk = 0; kk < k; ++kk)
                af[kk] = g_e4m3_lut[a[m * a_sm + kk * a_sk]];
            for (int j = 0; j < n; ++j) {
                const float* brow = bf + (size_t) j * k;
                float total = 0.0f;
                for (int kb = 0; kb < num_kb; ++kb) {
                    const int k0 = kb * block_k;
                    const int k1 = (k0 + block_k < k) ? k0 + block_k : k;
                    double block = 0.0;
                    for (int kk = k0; kk < k1; ++kk)
                        block += (double) af[kk] * (double) brow[kk];
                    const float scale = sfa[m * sfa_sm + kb * sfa_sk] * sfb[(j / gran_n) * sfb_sn + kb * sfb_sk];
                    total += scale * (float) block;
                }
                dgo_store(d, (int64_t) m * d_sm + j, d_dtype, accumulate, total);
            }
        }
        free(af);
    }
    free(bf);
}

/* Dense D = [D +] A @ B^T.  Mirrors fp8_gemm_nt (csrc/apis/gemm.hpp:73-124); nn/tn/tt are stride changes. */
int dgo_fp8_gemm(const uint8_t* a, int64_t a_sm, int64_t a_sk, const float* sfa, int64_t sfa_sm, int64_t sfa_sk,
                 const uint8_t* b, int64_t b_sn, int64_t b_sk, const float* sfb, int64_t sfb_sn, int64_t sfb_sk,
                 int gran_n, void* d, int64_t d_sm, int d_dtype, int accumulate, int m, int n, int k) {
    if (gran_n != 1 && gran_n != 32 && gran_n != 128)
        return 1;
    if (m == 0 || n == 0)
        return 0;
    dgo_rows(a, a_sm, a_sk, sfa, sfa_sm, sfa_sk, b, b_sn, b_sk, sfb, sfb_sn, sfb_sk, gran_n,
             d, d_sm, d_dtype, accumulate, 0, m, n, k, 0);
    return 0;
}

/*
 * M-grouped contiguous (csrc/apis/gemm.hpp:166-232).  A [M,K] K-major, B [G,N,K] (strides b_sg/b_sn/b_sk),
 * SFB [G, N/128, K/128] (sfb_sg/...).  `layout` is either per-row group ids (use_psum = 0; the group of an
 * m_alignment-row block is read from its FIRST row, negative => the block is written as zeros) or
 * cumulative group ends (use_psum = 1; group g covers rows [align(end[g-1], m_alignment), end[g]), rows in the
 * alignment gaps are left untouched).
 */
int dgo_fp8_gemm_m_grouped_contiguous(const uint8_t* a, int64_t a_sm, int64_t a_sk,
                                      const float* sfa, int64_t sfa_sm, int64_t sfa_sk,
                                      const uint8_t* b, int64_t b_sg, int64_t b_sn, int64_t b_sk,
                                      const float* sfb, int64_t sfb_sg, int64_t sfb_sn, int64_t sfb_sk,
                                      uint16_t* d, int64_t d_sm, const int32_t* layout, int num_groups,
                                      int use_psum, int m_alignment, int m, int n, int k) {
    if (m_alignment <= 0)
        return 1;
    if (use_psum) {
        int start = 0;
        for (int g = 0; g < num_groups; ++g) {
            const int end = layout[g];
            if (end > m || end < start)
                return 2;
            if (end > start)
                dgo_rows(a, a_sm, a_sk, sfa, sfa_sm, sfa_sk, b + g * b_sg, b_sn, b_sk,
                         sfb + g * sfb_sg, sfb_sn, sfb_sk, g_grouped_gran_n, d, d_sm, 0, 0, start, end, n, k, 0);
            start = (end + m_alignment - 1) / m_alignment * m_alignment;
        }
        return 0;
    }
    for (int m0 = 0; m0 < m; m0 += m_alignment) {
        const int m1 = (m0 + m_alignment < m) ? m0 + m_alignment : m;
        const int g = layout[m0];
        if (g >= num_groups)
            return 2;
        const int gg = g < 0 ? 0 : g;
        dgo_rows(a, a_sm, a_sk, sfa, sfa_sm, sfa_sk, b + gg * b_sg, b_sn, b_sk,
                 sfb + gg * sfb_sg, sfb_sn, sfb_sk, g_grouped_gran_n, d, d_sm, 0, 0, m0, m1, n, k, g < 0);
    }
    return 0;
}

/*
 * M-grouped masked (csrc/apis/gemm.hpp:250-297).  A [G,Mmax,K], B [G,N,K], D [G,Mmax,N] bf16; only rows
 * < masked_m[g] of each group are written (the reference leaves the rest undefined; the oracle leaves them untouched).
 */
int dgo_fp8_gemm_m_grouped_masked(const uint8_t* a, int64_t a_sg, int64_t a_sm, int64_t a_sk,
                                  const float* sfa, int64_t sfa_sg, int64_t sfa_sm, int64_t sfa_sk,
                                  const uint8_t* b, int64_t b_sg, int64_t b_sn, int64_t b_sk,
                                  const float* sfb, int64_t sfb_sg, int64_t sfb_sn, int64_t sfb_sk,
                                  uint16_t* d, int64_t d_sg, int64_t d_sm, const int32_t* masked_m,
                                  int num_groups, int m_max, int n, int k) {
    for (int g = 0; g < num_groups; ++g) {
        const int rows = masked_m[g];
        if (rows < 0 || rows > m_max)
            return 2;
        if (rows == 0)
            continue;
        dgo_rows(a + g * a_sg, a_sm, a_sk, sfa + g * sfa_sg, sfa_sm, sfa_sk, b + g * b_sg, b_sn, b_sk,
                 sfb + g * sfb_sg, sfb_sn, sfb_sk, g_grouped_gran_n, d + g * d_sg, d_sm, 0, 0, 0, rows, n, k, 0);
    }
    return 0;
}

/*
 * SF layout transform (csrc/jit_kernels/impls/smxx_layout.hpp:120-153): [batches, mn, sf_k] row-major FP32
 * -> element (b, i, j) at out[b * aligned_mn * sf_k + j * aligned_mn + i], aligned_mn = ceil(mn / 4) * 4.
 * Padding slots (i >= mn) are not written, as in the reference's kernel.
 */
int dgo_transpose_sf(const float* sf, float* out, int batches, int mn, int sf_k) {
    const int64_t aligned_mn = ((int64_t) mn + 3) / 4 * 4;
    for (int bi = 0; bi < batches; ++bi)
        for (int i = 0; i < mn; ++i)
            for (int j = 0; j < sf_k; ++j)
                out[(int64_t) bi * aligned_mn * sf_k + (int64_t) j * aligned_mn + i] =
                    sf[((int64_t) bi * mn + i) * sf_k + j];
    return 0;
}

/* get_mn_major_tma_aligned_packed_ue8m0_tensor, following the reference's torch statement of it
 * (csrc/jit_kernels/impls/smxx_layout.hpp:156-179; the CUDA kernels deep_gemm/include/deep_gemm/impls/smxx_layout.cuh:56,148
 * produce the same words): sf [batches, mn, sf_k] FP32 with element strides (sb, sm, sk) -> int32 words; byte j of word
 * (b, i, kq) = bits(sf[b][i][4 kq + j]) >> 23 truncated to 8 bits, zero for 4 kq + j >= sf_k; word stored at
 * out[b * packed_k * aligned_mn + kq * aligned_mn + i].  Padding rows (i >= mn) are not written.
 */
int dgo_pack_sf_ue8m0(const float* sf, int32_t* out, int batches, int mn, int sf_k, int64_t sb, int64_t sm, int64_t sk) {
    const int64_t aligned_mn = ((int64_t) mn + 3) / 4 * 4;
    const int packed_k = (sf_k + 3) / 4;
    for (int bi = 0; bi < batches; ++bi)
        for (int i = 0; i < mn; ++i)
            for (int kq = 0; kq < packed_k; ++kq) {
                uint32_t word = 0;
                for (int j = 0; j < 4; ++j) {
                    const int kb = 4 * kq + j;
                    if (kb < sf_k) {
                        uint32_t bits;
                        memcpy(&bits, &sf[bi * sb + i * sm + kb * sk], 4);
                        word |= ((bits >> 23) & 0xffu) << (8 * j);
                    }
                }
                out[(int64_t) bi * packed_k * aligned_mn + (int64_t) kq * aligned_mn + i] = (int32_t) word;
            }
    return 0;
}
