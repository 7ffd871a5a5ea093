/*
 * deepgemm_amd C ABI -- the drop-in boundary of the MI355X FP8 blockwise-scaled GEMM path.
 *
 * Every entry point replaces one host driver of the reference (deepseek-ai/DeepGEMM v2.6.1, paths relative to
 * /root/reference).  A caller (the Python host layer in deepgemm_amd/, or any FFI) passes raw DEVICE pointers,
 * element strides and a hipStream_t; nothing here allocates, synchronises or touches torch.
 *
 * Conventions
 *   - a / b        : FP8 e4m3fn (OCP) bytes.  Element (row, k) lives at ptr[row * stride_mn + k * stride_k]; exactly one of the
 *                    two strides is 1 (K-major or MN-major operand, reference csrc/utils/layout.hpp:13-24).
 *   - sfa          : FP32 per-(row, 128-K-block) scales, element (m, kb) at sfa[m * sfa_stride_m + kb * sfa_stride_k]
 *                    (the reference hands the kernel an MN-major, 16-byte aligned tensor: stride_m = 1,
 *                    stride_k = align(M, 4); any strides are accepted here).
 *   - sfb          : FP32 scales, element (n / sfb_gran_n, kb) at sfb[(n / sfb_gran_n) * sfb_stride_n + kb * sfb_stride_k];
 *                    sfb_gran_n is 128 (recipe (1,128,128)) or 1 (recipe (1,1,128)).
 *   - d            : row-major output, d_dtype DG_BF16 or DG_FP32, row stride d_stride_m (may exceed n).
 *   - accumulate   : 0 => D = A B^T;  1 => D_mem = D_mem + cast(A B^T) in D's dtype (the reference's reduce-add epilogue).
 *   - stream       : hipStream_t (0 = default stream).  Calls are asynchronous and never synchronise.
 *   - return value : 0 on success; non-zero => dg_last_error() describes the violated condition (thread-local text).
 *
 * Arithmetic (reference deep_gemm/include/deep_gemm/impls/sm90_fp8_gemm_1d2d.cuh:283-347, :416-418):
 *   D[m,n] = cast( sum_kb (sfa[m,kb] * sfb[n/gran,kb]) * ( sum_{k in block kb} A[m,k] * B[n,k] ) ), FP32 throughout,
 *   one round-to-nearest-even cast at the end.
 */
#ifndef DEEPGEMM_AMD_H
#define DEEPGEMM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_BF16 0
#define DG_FP32 1

/* Dense GEMM, NT form.  Replaces sm90_fp8_gemm_1d2d (csrc/jit_kernels/impls/sm90_fp8_gemm_1d2d.hpp:81),
 * sm90_fp8_gemm_1d1d (impls/sm90_fp8_gemm_1d1d.hpp:78) and sm100_fp8_fp4_gemm_1d1d (impls/sm100_fp8_fp4_gemm_1d1d.hpp:93)
 * as called from fp8_fp4_gemm_nt (csrc/apis/gemm.hpp:73-124); nn/tn/tt are stride changes on the same entry point. */
int dg_fp8_gemm_nt(const void* a, const float* sfa, const void* b, const float* sfb, void* d,
                   int m, int n, int k,
                   int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_n, int64_t b_stride_k,
                   int64_t sfa_stride_m, int64_t sfa_stride_k, int64_t sfb_stride_n, int64_t sfb_stride_k,
                   int sfb_gran_n, int64_t d_stride_m, int d_dtype, int accumulate, void* stream);

/* Dense GEMM, NT form, with the head-split output map of fp8_gemm_nt_skip_head_mid (csrc/apis/attention.hpp:19-73; column
 * map epilogue/transform.cuh:15-22): the N columns are heads of (head_left + head_right) columns; in D every head occupies
 * head_left + head_mid + head_right columns and the middle head_mid columns are left untouched, i.e. GEMM column n lands in
 * D column n + (n + head_right) / (head_left + head_right) * head_mid.  d is [m, n + n / (left + right) * mid]; no
 * accumulation.  Other arguments as dg_fp8_gemm_nt. */
int dg_fp8_gemm_nt_skip_head_mid(const void* a, const float* sfa, const void* b, const float* sfb, void* d,
                                 int m, int n, int k,
                                 int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_n, int64_t b_stride_k,
                                 int64_t sfa_stride_m, int64_t sfa_stride_k, int64_t sfb_stride_n, int64_t sfb_stride_k,
                                 int sfb_gran_n, int64_t d_stride_m, int d_dtype,
                                 int head_left, int head_mid, int head_right, void* stream);

/* Dense GEMM, NT form, power-of-two scales in the reference's packed UE8M0 format (SM100 input format, recipe (1, 1, 128):
 * sm100_fp8_fp4_gemm_1d1d, impls/sm100_fp8_fp4_gemm_1d1d.hpp:93; packing: deep_gemm/utils/math.py:19-23,
 * csrc/apis/layout.hpp:48-58).  sfa_packed / sfb_packed: int32, byte j of element (row, kq) = biased exponent of the
 * scale of K block 4 kq + j of that row of A / B (127 = 1.0); element (row, kq) at ptr[row * stride_mn + kq * stride_kq],
 * stride_mn must be 1 (MN-major).  The scaled MFMA applies the scales in hardware and accumulates in place over K: no FP32
 * promotion pass (4 waves per 256 x 256 tile, 128 x 128 wave tiles in AGPRs; 128 x 256 tiles for small problems and K tails).
 * Operands: K-major with 16-byte aligned rows (k % 128 == 0, or a dense K tail: k % 16 == 0 and k > 128) or -- round 4 -- MN-major
 * ([K][M] / [K][N], unit stride along m / n, 16-byte aligned k-rows, k % 128 == 0, m resp. n % 16 == 0): the nn / tn / tt layouts read in
 * place by the 8-wave hardware-scaled kernels (the reference's SM100 kernels take either majorness through UMMA descriptors:
 * csrc/apis/gemm.hpp:126-164). */
int dg_fp8_gemm_nt_ue8m0(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed, void* d,
                         int m, int n, int k,
                         int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_n, int64_t b_stride_k,
                         int64_t sfa_stride_m, int64_t sfa_stride_kq, int64_t sfb_stride_n, int64_t sfb_stride_kq,
                         int64_t d_stride_m, int d_dtype, int accumulate, void* stream);

/* Which MN-major operands of a dg_fp8_gemm_nt_ue8m0 call the caller should re-major into K-major scratch first (bit 0: A, bit 1: B); 0 = hand
 * them over as they are (the library has a kernel for this majorness combination and reading in place beats the extra pass at this size).
 * Pointers are only tested for alignment. */
int dg_ue8m0_dense_operand_plan(const void* a, const void* b, int m, int n, int k, int64_t a_stride_m, int64_t a_stride_k,
                                int64_t b_stride_n, int64_t b_stride_k);

/* M-grouped GEMMs with packed UE8M0 scales (the reference's SM100 drivers sm100_m_grouped_fp8_fp4_gemm_contiguous_1d1d /
 * _masked_1d1d, impls/sm100_fp8_fp4_gemm_1d1d.hpp:161,244, reached from csrc/apis/gemm.hpp:217-231,280-296 with int scale
 * tensors).  Tensors and grouped_layout / masked_m as in the FP32-scale entry points below; sfa_packed: one word per row of A
 * per four K blocks, element (row, kq) at ptr[row * stride_m + kq * stride_kq] (masked: + group * stride_g), stride_m must
 * be 1; sfb_packed: one word per ROW of B (recipe (1, 1, 128)), element (g, n, kq) at ptr[g * stride_g + n * stride_n +
 * kq * stride_kq], stride_n must be 1.  A and B K-major with 16-byte aligned rows, k % 128 == 0, m_alignment % 128 == 0.  The
 * contiguous entry also takes MN-major weights ([G][K][N]: b_stride_n == 1, the reference's m_grouped_fp8_gemm_nn_contiguous,
 * csrc/apis/gemm.hpp:234-248) where dg_ue8m0_grouped_operand_plan answers 0 -- read in place, no re-majoring pass. */
int dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed,
                                              void* d, const int32_t* grouped_layout, int num_groups, int m, int n, int k,
                                              int64_t a_stride_m, int64_t a_stride_k,
                                              int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k,
                                              int64_t sfa_stride_m, int64_t sfa_stride_kq,
                                              int64_t sfb_stride_g, int64_t sfb_stride_n, int64_t sfb_stride_kq,
                                              int64_t d_stride_m, int use_psum, int m_alignment, void* stream);
/* The same with a caller-owned scratch buffer (as dg_m_grouped_fp8_gemm_nt_contiguous_ws: 16-byte aligned, >= dg_split_k_workspace_bytes(), its
 * first 4096 bytes zero, one buffer per stream): the group-relative tiling of the contiguous layout (K >= 4096, M alignment 128, <= 64 row
 * blocks) then cuts its 128-row remainder tiles along K over the idle CUs.  workspace == NULL: whole remainder tiles. */
int dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0_ws(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed,
                                                 void* d, const int32_t* grouped_layout, int num_groups, int m, int n, int k,
                                                 int64_t a_stride_m, int64_t a_stride_k,
                                                 int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k,
                                                 int64_t sfa_stride_m, int64_t sfa_stride_kq,
                                                 int64_t sfb_stride_g, int64_t sfb_stride_n, int64_t sfb_stride_kq,
                                                 int64_t d_stride_m, int use_psum, int m_alignment,
                                                 void* workspace, int64_t workspace_bytes, void* stream);
/* dg_fp8_gemm_nt_ue8m0 / _g32 with a caller-owned scratch buffer (dg_split_k_workspace_bytes() bytes, 16-byte aligned; NULL = none): under-filled
 * launches with a long K loop (packed-scale dgrad shapes such as 4096 x 512 x 32768, narrow-layer weight gradients: m > 128, k % 512 == 0, at most
 * half the CUs' worth of 256 x 256 tiles) are cut along K into pieces that run as the groups of one launch of the K-grouped hardware-scaled
 * kernel; a second kernel on the same stream adds the FP32 partials in piece order and performs the output step.  gran_k 128 or 32.  The C ABI
 * never allocates: without a workspace the call runs as one ordinary launch.  dg_ue8m0_dense_wants_workspace: would a K-major, aligned problem
 * of this shape be cut (the host layer asks before it creates a buffer). */
int dg_fp8_gemm_nt_ue8m0_ws(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed, void* d,
                            int m, int n, int k,
                            int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_n, int64_t b_stride_k,
                            int64_t sfa_stride_m, int64_t sfa_stride_kq, int64_t sfb_stride_n, int64_t sfb_stride_kq,
                            int64_t d_stride_m, int d_dtype, int accumulate, int gran_k, void* workspace, int64_t workspace_bytes, void* stream);
int dg_ue8m0_dense_wants_workspace(int m, int n, int k);
/* Scale granularity 32 along K (round 6) -- the reference's SM100 MX recipe for FP8 x FP8 operands: recipe (1, 1, 32) / recipe_a = (1, 32)
 * (csrc/apis/gemm.hpp:311-312 gran_k == 32 or gran_k == 128; csrc/apis/layout.hpp:48-58 the (INT, 1, gran_k) and the FP32 cast branches;
 * per_token_cast_to_fp8(..., gran_k = 32, use_packed_ue8m0 = True), deep_gemm/utils/math.py:26-38; sweep tests/generators.py:192-194,230).
 * The packed words keep their meaning -- four consecutive exponents along K per int32 -- so a word now covers ONE 128-K block: byte j of
 * element (row, kb) = biased exponent of the scale of K bytes [128 kb + 32 j, 128 kb + 32 j + 32) of that row; element (row, kb) at
 * ptr[row * stride_mn + kb * stride_kq] (the *_stride_kq arguments: words per step of 128 along K), stride_mn must be 1.  It is the native block
 * size of v_mfma_scale_f32_16x16x128_f8f6f4 (one scale byte per MX block of 32 K-bytes, supplied by lane group j for block j).  Arguments otherwise as the gran-128 entries of the
 * same name; operands K-major with 16-byte aligned rows and k % 128 == 0 (MN-major operands: re-majored by the caller, dg_transpose_fp8);
 * the contiguous entry takes the workspace arguments of its _ws twin (unused by these kernels: NULL / 0 is fine). */
int dg_fp8_gemm_nt_ue8m0_g32(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed, void* d,
                             int m, int n, int k,
                             int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_n, int64_t b_stride_k,
                             int64_t sfa_stride_m, int64_t sfa_stride_kq, int64_t sfb_stride_n, int64_t sfb_stride_kq,
                             int64_t d_stride_m, int d_dtype, int accumulate, void* stream);
int dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0_g32(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed,
                                                  void* d, const int32_t* grouped_layout, int num_groups, int m, int n, int k,
                                                  int64_t a_stride_m, int64_t a_stride_k,
                                                  int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k,
                                                  int64_t sfa_stride_m, int64_t sfa_stride_kq,
                                                  int64_t sfb_stride_g, int64_t sfb_stride_n, int64_t sfb_stride_kq,
                                                  int64_t d_stride_m, int use_psum, int m_alignment,
                                                  void* workspace, int64_t workspace_bytes, void* stream);
int dg_m_grouped_fp8_gemm_nt_masked_ue8m0_g32(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed,
                                              void* d, const int32_t* masked_m, int num_groups, int m_max, int n, int k,
                                              int expected_m,
                                              int64_t a_stride_g, int64_t a_stride_m, int64_t a_stride_k,
                                              int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k,
                                              int64_t sfa_stride_g, int64_t sfa_stride_m, int64_t sfa_stride_kq,
                                              int64_t sfb_stride_g, int64_t sfb_stride_n, int64_t sfb_stride_kq,
                                              int64_t d_stride_g, int64_t d_stride_m, void* stream);

/* 2 = re-major the MN-major B of a dg_m_grouped_fp8_gemm_nt_contiguous_ue8m0 call into K-major scratch first, 0 = hand it over as it is (the
 * grouped twin of dg_ue8m0_dense_operand_plan: eligibility of the in-place kernel + the model of when the pass over all groups' weights costs
 * more than the slower K loop).  Pointers are only tested for alignment. */
int dg_ue8m0_grouped_operand_plan(const void* a, const void* b, int num_groups, int m, int n, int k, int64_t a_stride_m,
                                  int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k, int use_psum, int m_alignment);
int dg_m_grouped_fp8_gemm_nt_masked_ue8m0(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed,
                                          void* d, const int32_t* masked_m, int num_groups, int m_max, int n, int k,
                                          int expected_m,
                                          int64_t a_stride_g, int64_t a_stride_m, int64_t a_stride_k,
                                          int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k,
                                          int64_t sfa_stride_g, int64_t sfa_stride_m, int64_t sfa_stride_kq,
                                          int64_t sfb_stride_g, int64_t sfb_stride_n, int64_t sfb_stride_kq,
                                          int64_t d_stride_g, int64_t d_stride_m, void* stream);

/* SF packing: FP32 power-of-two scales [batches, mn, sf_k] (element strides given) -> packed UE8M0 words int32
 * [batches, mn, ceil(sf_k / 4)] in the MN-major layout (strides (ceil(sf_k / 4) * align(mn, 4), 1, align(mn, 4))): byte j
 * of word (row, kq) = (bits(sf[row][4 kq + j]) >> 23) & 0xff, zero for K blocks past sf_k.  Replaces
 * get_mn_major_tma_aligned_packed_ue8m0_tensor (csrc/jit_kernels/impls/smxx_layout.hpp:181-246; kernels
 * deep_gemm/include/deep_gemm/impls/smxx_layout.cuh:56,148; torch twin smxx_layout.hpp:156-179). */
int dg_pack_sf_ue8m0(const float* sf, int32_t* out, int batches, int mn, int sf_k,
                     int64_t sf_stride_b, int64_t sf_stride_mn, int64_t sf_stride_k, void* stream);

/* The same packing with the two steps the reference's FP32 -> UE8M0 cast branch puts around it (transform_sf_into_required_layout,
 * csrc/apis/layout.hpp:48-54, the default of its SM100 path) fused in:
 *   gran_mn > 1: `sf` holds one row of scales per `gran_mn` rows ([batches, ceil(mn / gran_mn), sf_k]); output row r takes source row
 *     r / gran_mn -- the reference materialises this broadcast with index_select first (layout.hpp:52-53), here there is no temporary;
 *   psum_layout != NULL (device pointer to num_psum_groups cumulative row ends of the psum contiguous layout, batches == 1): rows that
 *     lie in no group's range [align(end[g-1], m_alignment), end[g]) get zero words, as transpose_and_pack_fp32_into_ue8m0 does for
 *     the layout's uninitialised gap rows (deep_gemm/include/deep_gemm/impls/smxx_layout.cuh:76-94,139-141).
 * `mn` counts OUTPUT rows. */
int dg_pack_sf_ue8m0_ex(const float* sf, int32_t* out, int batches, int mn, int sf_k,
                        int64_t sf_stride_b, int64_t sf_stride_mn, int64_t sf_stride_k, int gran_mn,
                        const int32_t* psum_layout, int num_psum_groups, int m_alignment, void* stream);

/* Both scale tensors of one GEMM call in ONE launch (one kernel boundary in front of the GEMM instead of two): the SFA half takes the
 * arguments of dg_pack_sf_ue8m0_ex (psum_layout applies to SFA only, csrc/apis/layout.hpp:63-66), the SFB half has no psum layout;
 * both have sf_k = ceil(k / 128) K blocks.  What transform_sf_pair_into_required_layout (csrc/apis/layout.hpp:61-90) does for FP32
 * scales on the reference's SM100 path, as one kernel. */
int dg_pack_sf_pair_ue8m0(const float* sfa, int32_t* out_a, int batches_a, int m, int64_t sfa_stride_b, int64_t sfa_stride_m,
                          int64_t sfa_stride_k, int gran_m, const int32_t* psum_layout, int num_psum_groups, int m_alignment,
                          const float* sfb, int32_t* out_b, int batches_b, int n, int64_t sfb_stride_b, int64_t sfb_stride_n,
                          int64_t sfb_stride_k, int gran_n, int sf_k, void* stream);

/* First GEMM of an expert MLP with the SwiGLU activation and the per-token FP8 re-quantisation for the second GEMM fused into its
 * epilogue: the single-GPU half of the reference's Mega-MoE kernel (the L1 -> L2 hand-off of
 * deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_mega_moe.cuh; host side csrc/apis/mega.hpp:30-159, deep_gemm/mega/__init__.py:155;
 * its NVLink dispatch / combine is not part of this entry -- deepgemm_amd/ep.py exchanges tokens with RCCL around it).
 *   a [G, m_max, k] e4m3 K-major, sfa FP32 MN-major (element (g, m, kb) at sfa[g * stride_g + kb * stride_k + m]), masked_m as in
 *   dg_m_grouped_fp8_gemm_nt_masked.
 *   b_interleaved [G, n, k], n = 2 I: the expert's W1 with its gate rows (first I) and up rows (last I) interleaved in blocks of 64 --
 *   rows [128 j, 128 j + 64) = gate rows [64 j, 64 j + 64), rows [128 j + 64, 128 j + 128) = up rows [64 j, ...) -- and sfb its
 *   128 x 128 FP32 block scales [G, n / 128, k / 128] with the scale rows interleaved [gate 0, up 0, gate 1, up 1, ...]
 *   (deepgemm_amd.transform_weights_for_mega_moe; the reference interleaves at granularity 8 for its own kernel,
 *   deep_gemm/mega/__init__.py:115-122).
 *   out_fp8 [G, m_max, I] e4m3 (row stride out_stride_m) and out_sf FP32 (element (g, m, j) at out_sf[g * stride_g + j * stride_k + m],
 *   j = 128-block of I: the MN-major layout GEMM2 reads zero-copy): rows m < masked_m[g] hold
 *       y = bf16( silu(bf16(h[m, gate])) * bf16(h[m, up]) ),  h = A B^T with the blockwise scales (FP32 accumulation),
 *       optional clamp g <= c, |u| <= c (activation_clamp > 0);  per_token_cast_to_fp8(y) (deep_gemm/utils/math.py:26-38), 1 x 128 blocks,
 *   bit for bit what "dg_m_grouped_fp8_gemm_nt_masked -> BF16 -> SwiGLU -> per_token_cast_to_fp8" produces; other rows are untouched.
 *   workspace: dg_swiglu_workspace_bytes(num_groups, m_max, n) bytes of device memory, ZEROED ONCE by the caller; the kernel leaves it
 *   zeroed (the two workgroups that share a 1 x 128 quantisation block exchange their row amax through it), so one workspace serves every
 *   later launch and hipGraph replay on the same stream.  One workspace per launch that may be in flight at the same time (two
 *   streams, a graph replay racing an eager call): launches that share a workspace concurrently steal each other's slots.
 *   The partner wait is BOUNDED (dg_set_swiglu_exchange_timeout_us, default 10 s; reference: comm/barrier.cuh:12,36-40): a wait that
 *   times out -- a workspace that was not all-zero, a lost partner -- increments the uint32 at workspace[0] and gives the rows involved
 *   NaN scales and bytes and takes its own slot back; the caller re-zeroes the workspace all the same before the next launch on it.
 *   (No entry point injects a fault: the test of this bound sets DG_TEST_SWIGLU_FAULT=1 in its own environment + dg_reload_env.) */
int64_t dg_swiglu_workspace_bytes(int num_groups, int m_max, int n);
void dg_set_swiglu_exchange_timeout_us(int64_t us);
/* The same with the row's routing weight applied to the SwiGLU output BEFORE the re-quantisation, as the reference's fused kernel does
 * (deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_mega_moe.cuh:1001-1020): gate and up rounded to BF16 (and clamped there), then
 * y = silu(g) * u * row_weight[g, m] in FP32 -- no BF16 rounding of the product -- into the row amax and the FP8 cast;
 * row_weight FP32 [G, >= align(m_max, 64)] (element (g, m) at row_weight[g * stride_g + m]; 16-byte aligned, stride a multiple of 4);
 * NULL = the unweighted entry above. */
int dg_m_grouped_fp8_gemm_nt_masked_swiglu_weighted(const void* a, const float* sfa, const void* b_interleaved, const float* sfb, void* out_fp8,
                                                    float* out_sf, const int32_t* masked_m, int num_groups, int m_max, int n, int k, int expected_m,
                                                    int64_t a_stride_g, int64_t a_stride_m, int64_t b_stride_g, int64_t b_stride_n,
                                                    int64_t sfa_stride_g, int64_t sfa_stride_k, int64_t sfb_stride_g, int64_t sfb_stride_n,
                                                    int64_t sfb_stride_k, int64_t out_stride_g, int64_t out_stride_m, int64_t out_sf_stride_g,
                                                    int64_t out_sf_stride_k, float activation_clamp, int use_ue8m0, const float* row_weight,
                                                    int64_t row_weight_stride_g, void* workspace, int64_t workspace_bytes, void* stream);

/* World-size-1 dispatch / combine of the fused MoE operator (reference: the dispatch and combine stages of the Mega-MoE kernel,
 * sm100_fp8_fp4_mega_moe.cuh:357-405 and :523-595, host side csrc/apis/mega.hpp:30-159; with one rank they are a scatter into the masked
 * layout and a gather-sum back).  dg_moe_scatter_to_masked: every (token t, top-k entry j) with 0 <= topk_idx[t, j] < num_experts gets
 * the next free row slot of that expert: a_out[e, slot] = x_fp8[t] (hidden bytes), sfa_out (MN-major: element (e, kb, slot) at
 * sfa_out[e * stride_g + kb * stride_k + slot]) = x_sf[t, kb], row_weight_out[e, slot] = topk_weights[t, j], slot_out[t * topk + j] =
 * e * max_m + slot (-1 for an entry without an expert or beyond max_m rows -- the latter also increments *error_word);
 * masked_m_out[e] = rows of expert e (zeroed by this call, int32: what the masked GEMMs read on the device).
 * dg_moe_combine_from_masked: y[t] = bf16( sum_j float( y2[slot_out[t, j]] ) ) in top-k order, FP32 accumulation. */
int dg_moe_scatter_to_masked(const void* x_fp8, const float* x_sf, const void* topk_idx, int topk_idx_is_int64, const float* topk_weights,
                             int tokens, int hidden, int topk, int num_experts, int max_m, int64_t x_stride_m, int64_t x_sf_stride_m,
                             void* a_out, float* sfa_out, float* row_weight_out, int32_t* slot_out, int32_t* masked_m_out, void* error_word,
                             int64_t a_stride_g, int64_t a_stride_m, int64_t sfa_stride_g, int64_t sfa_stride_k, int64_t row_weight_stride_g,
                             void* stream);
int dg_moe_combine_from_masked(const void* y2_bf16, const int32_t* slot, int tokens, int topk, int hidden, int64_t y2_row_stride, void* y_bf16,
                               int64_t y_stride_m, void* stream);
int dg_m_grouped_fp8_gemm_nt_masked_swiglu(const void* a, const float* sfa, const void* b_interleaved, const float* sfb, void* out_fp8,
                                           float* out_sf, const int32_t* masked_m, int num_groups, int m_max, int n, int k, int expected_m,
                                           int64_t a_stride_g, int64_t a_stride_m, int64_t b_stride_g, int64_t b_stride_n,
                                           int64_t sfa_stride_g, int64_t sfa_stride_k, int64_t sfb_stride_g, int64_t sfb_stride_n,
                                           int64_t sfb_stride_k, int64_t out_stride_g, int64_t out_stride_m, int64_t out_sf_stride_g,
                                           int64_t out_sf_stride_k, float activation_clamp, int use_ue8m0, void* workspace, int64_t workspace_bytes,
                                           void* stream);

/* In-kernel dispatch / combine of the fused MoE operator over peer-mapped memory (round 6) -- the communication half of the reference's
 * Mega-MoE kernel, where tokens are pulled and results pushed between GPUs from inside the kernel with system-scope acquire / release
 * (deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_mega_moe.cuh:357-405 dispatch, :523-595 remote pulls / write-back;
 * deep_gemm/include/deep_gemm/comm/barrier.cuh:47-83 the NVLink barrier; host side csrc/apis/mega.hpp:30-159, buffer
 * deep_gemm/mega/__init__.py:18-58).  Protocol, memory ordering and the bound on every wait: csrc/fp8_gemm_moe.hpp.
 *
 * dg_symm_alloc / dg_symm_free: the rank's symmetric region (zeroed; fine-grained device memory where the runtime grants it,
 *   *out_fine_grained tells which; the environment variable DG_SYMM_COARSE forces ordinary device memory).
 * dg_ipc_get_handle / dg_ipc_open_handle / dg_ipc_close_handle: hipIpcGetMemHandle / hipIpcOpenMemHandle / hipIpcCloseMemHandle with the
 *   handle as 64 opaque bytes -- how the ranks of one node map each other's regions (exchanged by the caller, e.g. all_gather_object).
 * dg_moe_p2p_layout: byte offsets of {counts, arrived, combined, done, l1_acts [E_loc, cap, H] e4m3, l1_sf [E_loc, H / 128, cap] FP32
 *   (MN-major), row_weight [E_loc, cap] FP32, src_info [E_loc, cap] int32, y_rows [T, topk, H] BF16, total bytes} inside a region.
 * All three kernels take the region base of every rank AS MAPPED INTO THE CALLING PROCESS (peer_regions[rank] = the own region), the
 * problem geometry (local_experts = E / world, capacity = rows per local expert, hidden, max_tokens, topk) and the call's epoch (1, 2, ...:
 * the same on every rank, never 0).  errors: device uint32 [4] -- [0] += rows this rank sent that found their expert full (dropped: the pair
 * contributes nothing to y), [1] = word 0 of the fused L1 kernel's workspace, [2] / [3] += dispatch / combine flag waits that timed out
 * (dg_set_moe_p2p_timeout_us, default 10 s; reference: comm/barrier.cuh:12,36-40).
 *   dg_moe_p2p_dispatch: every (token t < tokens, entry j) with 0 <= topk_idx[t, j] < E pushes x_fp8[t], x_sf[t, :], topk_weights[t, j] and its
 *     return address into the next free row of expert e on rank e / local_experts; pair_ok_out[t * topk + j] = 1 if delivered.  Returns after
 *     every rank's rows have arrived here: masked_m_out[e_local] = rows of the local expert (int32, what the masked GEMMs read).
 *   dg_moe_p2p_combine: row slot (e, r < masked_m[e]) of l2_out_bf16 [E_loc, cap, H] goes back to y_rows[t, j] of its source rank.
 *   dg_moe_p2p_reduce: y[t] = bf16( sum_j float(y_rows[t, j]) ) over the delivered pairs in top-k order, after every owner has returned.
 * Stream-ordered, no host synchronisation, no allocation; one call of each per step on every rank of the group, in this order.  The epoch is
 * a launch ARGUMENT: a step captured in a hipGraph would replay its epoch (its waits would see the previous step's flags) -- not capturable. */
void dg_set_moe_p2p_timeout_us(int64_t us);
int dg_symm_alloc(int64_t bytes, void** out_ptr, int* out_fine_grained);
int dg_symm_free(void* ptr);
int dg_ipc_get_handle(void* ptr, void* handle_out_64_bytes);
int dg_ipc_open_handle(const void* handle_64_bytes, void** out_ptr);
int dg_ipc_close_handle(void* ptr);
int dg_moe_p2p_layout(int local_experts, int capacity, int hidden, int max_tokens, int topk, int world, int64_t* offsets_out_10);
int dg_moe_p2p_dispatch(const void* const* peer_regions, int world, int rank, int local_experts, int capacity, int hidden, int max_tokens, int topk,
                        const void* x_fp8, const float* x_sf, const void* topk_idx, int topk_idx_is_int64, const float* topk_weights, int tokens,
                        int64_t x_stride_m, int64_t x_sf_stride_m, uint32_t epoch, int32_t* masked_m_out, void* pair_ok_out, void* errors,
                        void* stream);
int dg_moe_p2p_combine(const void* const* peer_regions, int world, int rank, int local_experts, int capacity, int hidden, int max_tokens, int topk,
                       const void* l2_out_bf16, int64_t l2_stride_g, int64_t l2_stride_m, const int32_t* masked_m, uint32_t epoch, void* errors,
                       void* stream);
int dg_moe_p2p_reduce(const void* const* peer_regions, int world, int rank, int local_experts, int capacity, int hidden, int max_tokens, int topk,
                      const void* pair_ok, int tokens, void* y_bf16, int64_t y_stride_m, const void* swiglu_workspace, uint32_t epoch, void* errors,
                      void* stream);

/* K-grouped contiguous GEMM (MoE weight gradients): D[g] += A_g * B_g^T for every group g, where group g owns the K range
 * [sum(ks[:g]), sum(ks[:g+1])) of both operands.  Replaces sm90_k_grouped_fp8_gemm_1d1d / sm100_k_grouped_fp8_gemm_1d1d as
 * called from k_grouped_fp8_gemm_nt_contiguous / k_grouped_fp8_gemm_tn_contiguous (csrc/apis/gemm.hpp:299-400).
 *   ks_host: HOST array of num_groups K extents (the reference's ks_cpu), each a multiple of 128; 0 = empty group, D[g] kept.
 *   ab_layout DG_KGROUPED_BLOCKS: group g's K-major [m, ks[g]] (resp. [n, ks[g]]) matrix is stored contiguously at element
 *     offset m * sum(ks[:g]) (resp. n * ...): the reference's SM90 NT operand form; a_stride_m / b_stride_n are ignored.
 *   ab_layout DG_KGROUPED_COLUMNS: a is one K-major [m, sum_k] matrix with row stride a_stride_m (b: [n, sum_k], b_stride_n)
 *     and group g is a column range: what an MN-major [sum_k, m] operand (TN form) becomes after dg_transpose_fp8.
 *   ab_layout DG_KGROUPED_ROWS: the MN-major operands themselves, a [sum_k, m] with row pitch a_stride_m (b: [sum_k, n],
 *     b_stride_n), group g = a row range: the TN form without a re-majoring pass (hardware transpose reads out of LDS).
 *     Needs 16-byte aligned rows, MN-major 16-byte aligned scales, m > 64, at most 64 groups; otherwise returns 3 WITHOUT
 *     launching anything and the caller re-majors (dg_transpose_fp8) and uses DG_KGROUPED_COLUMNS -- the host layer tries
 *     this form first and keeps no copy of the conditions.
 *   sfa element (row, kb) at sfa[row * sfa_stride_m + kb * sfa_stride_k], kb counted over the whole K axis; same for sfb
 *   (one scale per row of B per 128-K block: recipe (1, 1, 128)).
 *   d [num_groups, m, n] FP32, dense; the result is accumulated onto it (the caller copies C into D first). */
#define DG_KGROUPED_BLOCKS 0
#define DG_KGROUPED_COLUMNS 1
#define DG_KGROUPED_ROWS 2
int dg_k_grouped_fp8_gemm_nt_contiguous(const void* a, const float* sfa, const void* b, const float* sfb, float* d,
                                        int m, int n, const int32_t* ks_host, int num_groups, int ab_layout,
                                        int64_t a_stride_m, int64_t b_stride_n,
                                        int64_t sfa_stride_m, int64_t sfa_stride_k, int64_t sfb_stride_n, int64_t sfb_stride_k,
                                        void* stream);

/* The TN form with the reference's psum layout, K ranges read ON THE DEVICE (no host copy of the group sizes): replaces
 * sm100_k_grouped_fp8_gemm_1d1d(..., use_psum_layout = true) as called from k_grouped_fp8_gemm_tn_contiguous with ks_cpu missing
 * (csrc/apis/gemm.hpp:48-69,299-346; scheduler: deep_gemm/include/deep_gemm/scheduler/gemm.cuh:74-85,238-261) for FP32 per-channel
 * scales with gran_k = K alignment = 128.
 *   psum_layout (device, int32[num_groups]): group g ends at row psum_layout[g] of the K axis and starts at the previous end rounded up
 *     to 128; rows between an end and the next multiple of 128 hold zeros (the layout's contract: whole 128-row blocks are computed);
 *     an end equal to its start = empty group, D[g] stays as it is.  total_k = rows of a / b, a multiple of 128.
 *   ab_layout DG_KGROUPED_ROWS: a [total_k, m], b [total_k, n] MN-major as they are; DG_KGROUPED_COLUMNS: K-major [m, total_k] /
 *     [n, total_k] (after dg_transpose_fp8).  Needs m > 64, 16-byte aligned rows and MN-major, 16-byte aligned scales: otherwise
 *     returns 3 WITHOUT launching (ROWS callers re-major and retry with COLUMNS; there is no per-group fallback -- the host does not
 *     know the ranges).  Scales and d as in dg_k_grouped_fp8_gemm_nt_contiguous. */
int dg_k_grouped_fp8_gemm_tn_psum(const void* a, const float* sfa, const void* b, const float* sfb, float* d, int m, int n, int total_k,
                                  const int32_t* psum_layout, int num_groups, int ab_layout, int64_t a_stride_m, int64_t b_stride_n,
                                  int64_t sfa_stride_m, int64_t sfa_stride_k, int64_t sfb_stride_n, int64_t sfb_stride_k, void* stream);
/* The same for a K alignment other than 128 (round 6; the reference's SM100 sweep: 32 / 160 / 192 / 224 with gran_k = 128,
 * tests/generators.py:192-194, scheduler/gemm.cuh:74-85, 238-261): group g covers K rows [align(end[g-1], k_alignment), end[g]), its scale rows
 * are COMPACT and count from its own start (ceil(extent / 128) rows per non-empty group, in group order), its last 128-block may be partial (the
 * rows at and beyond end[g] do not contribute, whatever they hold).  k_alignment % 32 == 0, total_k % k_alignment == 0; alignments != 128 take
 * MN-major operands only (DG_KGROUPED_ROWS: the reference's own restriction, tests/generators.py:497). */
int dg_k_grouped_fp8_gemm_tn_psum_aligned(const void* a, const float* sfa, const void* b, const float* sfb, float* d, int m, int n, int total_k,
                                          const int32_t* psum_layout, int num_groups, int ab_layout, int64_t a_stride_m, int64_t b_stride_n,
                                          int64_t sfa_stride_m, int64_t sfa_stride_k, int64_t sfb_stride_n, int64_t sfb_stride_k,
                                          int k_alignment, void* stream);

/* K-grouped GEMM with packed UE8M0 scale words: the reference's SM100 form of k_grouped_fp8_gemm_tn_contiguous.  Replaces
 * sm100_k_grouped_fp8_gemm_1d1d (csrc/jit_kernels/impls/sm100_fp8_fp4_gemm_1d1d.hpp:244-318) as called from csrc/apis/gemm.hpp:299-346
 * with int scale tensors (or FP32 ones after dg_pack_sf_k_grouped_ue8m0).  Recipe (1, 1, gran_k), gran_k = 128 or 32: one exponent per row of a /
 * row of b and gran_k K bytes -- the MX block format of the scaled MFMA; no FP32 promotion, D[g] += A[:, K_g] B[:, K_g]^T accumulates in the
 * matrix core over the group's whole K range.
 *   ab_layout DG_KGROUPED_ROWS: a [total_k, m], b [total_k, n] MN-major as the reference hands them over (k-row pitches a_stride_m / b_stride_n,
 *     16-byte aligned), read in place through transposing fragment reads -- needs m > 128, returns 3 WITHOUT launching otherwise (callers
 *     re-major with dg_transpose_fp8 and retry with DG_KGROUPED_COLUMNS); DG_KGROUPED_COLUMNS: a [m, total_k], b [n, total_k] K-major (row
 *     pitches a_stride_m / b_stride_n, 16-byte aligned).  d [num_groups, m, n] FP32, accumulated in place.
 *   K ranges: psum_layout == NULL: ks_host[g] (host, multiples of 32) one after another; psum_layout != NULL (device int32 [num_groups], ks_host
 *     ignored): group g covers [align(end[g-1], k_alignment), end[g]) and the columns up to align(end[g], k_alignment) hold zeros (the reference's
 *     psum layout, tests/generators.py:480-530, scheduler/gemm.cuh:74-85).  k_alignment % 32 == 0.  A group's last 128-block may be partial.
 *   sfa_packed [packed_sf_k, m], sfb_packed [packed_sf_k, n] int32, unit stride along m / n, row pitches sf*_stride_k (words, multiples of 4,
 *     16-byte aligned base): as the reference packs them (impls/smxx_layout.cuh:148-246) -- group g owns ceil(ceil(k_g / gran_k) / 4) rows counted
 *     from the end of the group before it; byte j of its row r = exponent of its scale block 4 r + j (0 beyond its last block).
 *   At most 64 groups with ks_host, 128 with psum_layout.  Returns 3 without launching when an alignment condition fails. */
int dg_k_grouped_fp8_gemm_ue8m0(const void* a, const int32_t* sfa_packed, const void* b, const int32_t* sfb_packed, float* d,
                                int m, int n, int total_k, const int32_t* ks_host, const int32_t* psum_layout, int num_groups,
                                int k_alignment, int gran_k, int ab_layout, int64_t a_stride_m, int64_t b_stride_n,
                                int64_t sfa_stride_k, int64_t sfb_stride_k, void* stream);
/* FP32 power-of-two scales of a K-grouped operand, [sf_k, mn] row-major (per group ceil(k_g / gran_k) rows, compact, in group order), into the
 * packed words above ([packed_sf_k, mn] int32, row pitch mn).  Replaces pack_fp32_into_ue8m0 with kNumGroups > 1
 * (deep_gemm/include/deep_gemm/impls/smxx_layout.cuh:146-246, host side csrc/jit_kernels/impls/smxx_layout.hpp:255-317).  group_ks: device int32
 * [num_groups] -- K extents, or (use_psum != 0) psum ends with the K alignment k_alignment.  mn % 4 == 0, num_groups <= 128; only the exponent
 * byte of each value is kept (the reference asserts sign and mantissa are zero).  Rows of `out` beyond the groups' last are left untouched. */
int dg_pack_sf_k_grouped_ue8m0(const float* sf, int32_t* out, const int32_t* group_ks, int num_groups, int mn, int sf_k, int packed_sf_k,
                               int gran_k, int k_alignment, int use_psum, void* stream);

/* M-grouped contiguous GEMM.  Replaces sm90_m_grouped_fp8_gemm_contiguous_1d2d (impls/sm90_fp8_gemm_1d2d.hpp:147) /
 * sm100_m_grouped_fp8_fp4_gemm_contiguous_1d1d (impls/sm100_fp8_fp4_gemm_1d1d.hpp:161) as called from
 * m_grouped_fp8_fp4_gemm_nt_contiguous (csrc/apis/gemm.hpp:166-232).
 *   a [m, k] K-major; b [num_groups, n, k] (b_stride_g between groups); sfb [num_groups, ceil(n/128), ceil(k/128)];
 *   d [m, n] BF16.  grouped_layout: int32 [m] group id per row, negative = padding row (use_psum = 0; the group of an
 *   m_alignment-row block is taken from its first row and a negative id makes the block all zeros), or int32 [num_groups]
 *   cumulative row ends (use_psum = 1; group g owns rows [align(end[g-1], m_alignment), end[g]); the alignment-gap rows
 *   of D are written as zeros). */
int dg_m_grouped_fp8_gemm_nt_contiguous(const void* a, const float* sfa, const void* b, const float* sfb, void* d,
                                        const int32_t* grouped_layout, int num_groups, int m, int n, int k,
                                        int64_t a_stride_m, int64_t a_stride_k,
                                        int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k,
                                        int64_t sfa_stride_m, int64_t sfa_stride_k,
                                        int64_t sfb_stride_g, int64_t sfb_stride_n, int64_t sfb_stride_k,
                                        int64_t d_stride_m, int use_psum, int m_alignment, void* stream);

/* The same with a caller-owned scratch buffer for tail balancing: when the tile count is just above a multiple of the CU count
 * (BASELINE config 4: 576 tiles of 128 x 256 on 256 CUs = 2.25 rounds) the partial last round is cut along K over the idle CUs:
 * FP32 partial tiles go through `workspace` and a second kernel, launched behind the first on the same stream, sums them in a fixed
 * order (deterministic) and stores the tiles.  workspace: device memory, 16-byte aligned, at least dg_split_k_workspace_bytes()
 * bytes, contents irrelevant, used by one stream at a time; workspace == NULL gives the entry point above.
 * (The reference's persistent scheduler has no such step: its tiles are not split; the library never allocates, hence the
 * caller-owned buffer -- the host layer keeps one per device and stream.) */
int dg_m_grouped_fp8_gemm_nt_contiguous_ws(const void* a, const float* sfa, const void* b, const float* sfb, void* d,
                                           const int32_t* grouped_layout, int num_groups, int m, int n, int k,
                                           int64_t a_stride_m, int64_t a_stride_k,
                                           int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k,
                                           int64_t sfa_stride_m, int64_t sfa_stride_k,
                                           int64_t sfb_stride_g, int64_t sfb_stride_n, int64_t sfb_stride_k,
                                           int64_t d_stride_m, int use_psum, int m_alignment, void* workspace, int64_t workspace_bytes,
                                           void* stream);
int64_t dg_split_k_workspace_bytes(void);

/* dg_fp8_gemm_nt with the same caller-owned scratch buffer: a dense problem whose tiles do not fill the chip (or leave a partial
 * last round) and whose K loop is long -- e.g. the dgrad shape 4096 x 512 x 32768: 64 tiles of 128 x 256 on 256 CUs -- is cut
 * along K over the idle CUs in the same way.  workspace == NULL gives dg_fp8_gemm_nt. */
int dg_fp8_gemm_nt_ws(const void* a, const float* sfa, const void* b, const float* sfb, void* d,
                      int m, int n, int k,
                      int64_t a_stride_m, int64_t a_stride_k, int64_t b_stride_n, int64_t b_stride_k,
                      int64_t sfa_stride_m, int64_t sfa_stride_k, int64_t sfb_stride_n, int64_t sfb_stride_k,
                      int sfb_gran_n, int64_t d_stride_m, int d_dtype, int accumulate,
                      void* workspace, int64_t workspace_bytes, void* stream);

/* M-grouped masked GEMM.  Replaces sm90_m_grouped_fp8_gemm_masked_1d2d (impls/sm90_fp8_gemm_1d2d.hpp:224) /
 * sm100_m_grouped_fp8_fp4_gemm_masked_1d1d (impls/sm100_fp8_fp4_gemm_1d1d.hpp:244) as called from
 * m_grouped_fp8_fp4_gemm_nt_masked (csrc/apis/gemm.hpp:250-297).
 *   a [num_groups, m_max, k]; b [num_groups, n, k]; d [num_groups, m_max, n] BF16; masked_m int32 [num_groups] ON THE
 *   DEVICE (read inside the kernel); rows >= masked_m[g] are not written.  expected_m is a tuning hint only. */
int dg_m_grouped_fp8_gemm_nt_masked(const void* a, const float* sfa, const void* b, const float* sfb, void* d,
                                    const int32_t* masked_m, int num_groups, int m_max, int n, int k, int expected_m,
                                    int64_t a_stride_g, int64_t a_stride_m, int64_t a_stride_k,
                                    int64_t b_stride_g, int64_t b_stride_n, int64_t b_stride_k,
                                    int64_t sfa_stride_g, int64_t sfa_stride_m, int64_t sfa_stride_k,
                                    int64_t sfb_stride_g, int64_t sfb_stride_n, int64_t sfb_stride_k,
                                    int64_t d_stride_g, int64_t d_stride_m, void* stream);

/* SF layout kernel.  Replaces transpose_fp32 (deep_gemm/include/deep_gemm/impls/smxx_layout.cuh:12-50) as driven by
 * get_mn_major_tma_aligned_tensor (csrc/jit_kernels/impls/smxx_layout.hpp:120-153):
 * sf [batches, mn, sf_k] row-major FP32 -> out element (b, i, j) at out[b * aligned_mn * sf_k + j * aligned_mn + i],
 * aligned_mn = align(mn, 4).  Padding slots are not written. */
int dg_transpose_sf_fp32(const float* sf, float* out, int batches, int mn, int sf_k, void* stream);

/* Fused per-token quantiser, the producer of operand A: BF16 x [m, n] (row stride x_stride_m) -> e4m3fn out [m, n] plus
 * one FP32 scale per 1 x 128 block at sf[row * sf_stride_m + kb * sf_stride_k] (row-major [m, ceil(n/128)] as the reference
 * returns it, or strides (1, align(m, 4)) to land directly in the GEMM's SFA layout).  Arithmetic of
 * per_token_cast_to_fp8 (deep_gemm/utils/math.py:26-38; the reference leaves this cast to the caller, README.md:72):
 * sf = max(amax, 1e-4) / 448, rounded up to a power of two if use_ue8m0 (math.py:13-16), q = e4m3fn(float(x) * (1 / sf)). */
int dg_per_token_cast_to_fp8(const void* x_bf16, void* out_fp8, float* sf, int m, int n,
                             int64_t x_stride_m, int64_t out_stride_m, int64_t sf_stride_m, int64_t sf_stride_k,
                             int use_ue8m0, void* stream);

/* Fused block quantisers, one pass: BF16 x [rows, cols] -> e4m3fn out [rows, cols] plus FP32 scales at
 * sf[rb * sf_stride_r + c * sf_stride_c].  per_channel = 0: one scale per 128 x 128 block, sf [ceil(rows/128), ceil(cols/128)]
 * (per_block_cast_to_fp8, deep_gemm/utils/math.py:51-61; operand B of the GEMMs); per_channel = 1: one scale per column per
 * 128-row block, sf [ceil(rows/128), cols] (per_channel_cast_to_fp8, math.py:41-48; operands of the K-grouped GEMM).
 * Scale arithmetic as dg_per_token_cast_to_fp8. */
int dg_block_cast_to_fp8(const void* x_bf16, void* out_fp8, float* sf, int rows, int cols,
                         int64_t x_stride_r, int64_t out_stride_r, int64_t sf_stride_r, int64_t sf_stride_c,
                         int per_channel, int use_ue8m0, void* stream);

/* Operand re-majoring for the fast path: dst[b][c][r] = src[b][r][c], 1-byte (FP8) elements, `rows` x `cols` per batch,
 * leading dimensions / batch strides in elements.  Turns an MN-major operand (the SM100 form of fp8_gemm_nn/tn/tt,
 * csrc/apis/gemm.hpp:126-164; UMMA descriptors consume it in place there) into the K-major form the LDS-DMA kernels
 * stream; the host layer owns the scratch buffer. */
int dg_transpose_fp8(const void* src, void* dst, int batches, int rows, int cols,
                     int64_t src_ld, int64_t dst_ld, int64_t src_batch_stride, int64_t dst_batch_stride, void* stream);

/* Runtime knobs (reference csrc/apis/runtime.hpp:12-41: set/get_num_sms; the analogue here is the CU budget a
 * persistent launch may occupy, 0 = all CUs of the device). */
int dg_set_num_cus(int num_cus);
int dg_get_num_cus(void);

/* The tuning / diagnostic environment variables (DG_PRINT_CONFIGS, DG_GROUP_M, DG_KS_PIECES, DG_PC_BM, DG_E8_TAB_UNSPLIT, DG_TAB_UNFUSED, DG_TABLE_KERNEL, DG_SK_EXCHANGE, DG_TEST_SWIGLU_FAULT,
 * DG_SFA_ROWMAJOR_IN_PLACE, DG_SWIGLU_ONE_PER_CU) are read once, at the first launch; a process that changes them afterwards calls this to have them read again
 * (tests, tuning scripts); launches in flight on other threads keep the snapshot they started with.  No reference counterpart (its knobs are read per call, csrc/utils/system.hpp get_env). */
void dg_reload_env(void);

/* Tuning / test hook: force a kernel configuration by name for subsequent calls of the process ("auto" restores the
 * heuristic; like the reference's runtime knobs the setting is process-wide and may be changed from any thread).
 * Unknown names are an error.  dg_list_configs() returns a comma-separated list, dg_get_forced_config() the current name. */
int dg_set_forced_config(const char* name);
const char* dg_get_forced_config(void);
const char* dg_list_configs(void);
/* Tuning aid: when non-null, the pipe / ring kernels write 4 int64 s_memtime stamps per wave {kernel entry, K loop
 * begin, K loop end, after the stores} of each block's first tile to device_buffer[(block * waves + wave) * 4 + i].
 * The buffer must hold grid * 8 * 4 int64.  Null (the default) disables it. */
int dg_set_debug_buffer(void* device_buffer);
/* Name of the configuration the last GEMM call on this thread selected (for DG_PRINT_CONFIGS-style logging). */
const char* dg_last_config(void);
/* The kernel configuration the automatic selection picks for a problem of this shape (16-byte aligned, densely packed operands of
 * the given majorness, MN-major SFA; gemm_type: 0 dense, 1 contiguous, 2 contiguous psum, 3 masked; m = rows of the dense / contiguous
 * problem or rows per group of the masked one).  Nothing is launched and no device is needed: the analogue of inspecting
 * get_best_config (csrc/jit_kernels/heuristics/common.hpp:14-52) -- the host tests pin the choices with it. */
const char* dg_select_config(int gemm_type, int m, int n, int k, int num_groups, int expected_m, int a_mn_major, int b_mn_major,
                             int sfb_gran_n, int m_alignment, int has_workspace, int packed_ue8m0);

/* Would a dense dg_fp8_gemm_nt(_ws) call with K-major 16-byte aligned operands, one SFB value per 128 columns and a ROW-major SFA
 * ([m][ceil(k / 128)] floats: sfa_stride_m = ceil(k / 128), sfa_stride_k = 1 -- how the reference's callers hold it before the layout
 * step, tests/test_fp8_fp4.py:45-55) read the SFA in place (config duo_p_rm_256x256)?  1: pass it as it is; 0: transpose first
 * (dg_transpose_sf_fp32; csrc/jit_kernels/impls/smxx_layout.hpp:120-153 is the launch this saves). */
int dg_dense_rowmajor_sfa_native(int m, int n, int k);

/* 1 if the automatic selection would cut this dense problem along K given a workspace (dg_fp8_gemm_nt_ws): under-filled launches
 * with long K loops, partial last rounds of 128 x 256 tiles, under-filled recipe-(1, 1, 128) launches.  The host layer asks before it
 * creates and passes its per-stream buffer (dg_split_k_workspace_bytes()); operands assumed 16-byte aligned and densely packed. */
int dg_dense_wants_workspace(int m, int n, int k, int a_mn_major, int b_mn_major, int sfb_gran_n);

/* Which MN-major FP8 operands the host side should re-major into K-major scratch (dg_transpose_fp8) before calling
 * dg_fp8_gemm_nt_ws / dg_m_grouped_fp8_gemm_nt_contiguous_ws: bit 0 = A, bit 1 = B; 0 = every operand is read where it lies.
 * Decided with the predicates the launch itself applies (pointer and pitch alignment, 32-bit offset range, tile rules), so an
 * operand left in place always finds its native kernel.  The reference reads the majorness off the strides (csrc/apis/gemm.hpp:83-88)
 * and hands it to the TMA descriptors (SM100: csrc/jit_kernels/impls/sm100_fp8_fp4_gemm_1d1d.hpp; SM90 asserts K-major,
 * csrc/jit_kernels/impls/sm90_fp8_gemm_1d2d.hpp:90); here the kernels that read MN-major operands natively have alignment rules.
 * gemm_type as in dg_select_config; b_sg = group pitch of B (0 for dense); sfa_sm = element stride of SFA along m. */
int dg_operand_plan(int gemm_type, const void* a, const void* b, int m, int n, int k, int64_t a_sm, int64_t a_sk, int64_t b_sn,
                    int64_t b_sk, int64_t b_sg, int64_t sfa_sm, int sfb_gran_n, int m_alignment);

const char* dg_last_error(void);
const char* dg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DEEPGEMM_AMD_H */
