// Lab notebook of the gfx950 FP8 GEMM kernels: superseded production forms, timing ablations and rejected variants.
// Compiled only with DG_EXPERIMENTS=1 (python __graft_entry__.py), never selected by the heuristics; HISTORY.md quotes their
// configuration names (naive_256x256, ring_256x256, rabl*, ring_p*, dabl*, duo_pprio, e8_ring, quad_128x256, ...) as evidence.
// The production kernels live in fp8_gemm_kernels.hpp / fp8_gemm_quad.hpp and carry none of these hooks.
//
//   dg_fp8_gemm_fast_kernel      first version of the fast path: hipcc schedules the MFMA + promotion stream (builtins)
//   dg_fp8_gemm_ring_kernel      previous dense production form (3 + 2 slot LDS rings, two barriers per K block, mixed
//                                MFMA / load stream per wave); RABL / PAD ablations
//   dg_fp8_gemm_duo_abl_kernel   the duo kernel with its ~30 DABL timing modes (no stagger, priorities, early barriers,
//                                pieces between MFMAs, per-step traces, L2-resident sources ...)
//   dg_fp8_gemm_e8_kernel        hardware-scaled MFMA in the ring schedule (A/B partner of e8_duo / e8_quad)
//   dg_fp8_gemm_quad_kernel      FP32-scale promotion in the one-wave-per-SIMD schedule of the UE8M0 quad kernel: bit-identical
//                                to duo and 1.4x slower (a lone wave pays ~51 cycles per MFMA + 4 FMA step, ~60 per LDS-DMA piece)
#pragma once
#include "fp8_gemm_kernels.hpp"
#include "fp8_gemm_quad.hpp"

namespace dg {

// Ablation helper: the MFMA of a step with a single token VALU op instead of the four promotion FMAs.
__device__ __forceinline__ void mfma_only_step(v4f& part_new, const v8i& rows_operand, const v8i& cols_operand, float& c,
                                               const v4f& part_old) {
    asm volatile(
        "v_mfma_f32_16x16x128_f8f6f4 %0, %2, %3, 0\n\t"
        "v_add_f32 %1, %1, %4"
        : "=&v"(part_new), "+v"(c)
        : "v"(rows_operand), "v"(cols_operand), "v"(part_old[0])
        : "memory");
}

// One 128-K block of a wave tile, software-pipelined by hand: MFMA i+DEPTH is issued before the FP32 promotion of
// MFMA i, so the matrix pipe never waits for a VALU read of its own result (hipcc serialises the naive form into
// mfma / s_nop 11 / fma).  sched_group_barrier pins the interleave: 1 MFMA, then 4 VALU FMAs (+ the LDS reads of the
// next A fragment at the head of each M-subtile).
template <int MS, int NS, int DEPTH>
__device__ __forceinline__ void compute_block_pipelined(const uint8_t* a_tile, const uint8_t* b_tile, int frag_off,
                                                        const float (&scale)[MS], v4f (&acc)[MS][NS]) {
    constexpr int TOTAL = MS * NS;
    v8i bf[NS];
    #pragma unroll
    for (int ns = 0; ns < NS; ++ns)
        bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
    v8i af[2];
    af[0] = load_fragment(a_tile, frag_off);
    v4f part[DEPTH + 1];
    #pragma unroll
    for (int i = 0; i < TOTAL + DEPTH; ++i) {
        if (i < TOTAL) {
            const int ms = i / NS, ns = i % NS;
            if (ns == 0 && ms + 1 < MS)
                af[(ms + 1) & 1] = load_fragment(a_tile + (ms + 1) * 2048, frag_off);
            part[i % (DEPTH + 1)] = mfma_fp8_k128(bf[ns], af[ms & 1]);
        }
        if (i >= DEPTH) {
            const int j = i - DEPTH, ms = j / NS, ns = j % NS;
            const v4f pr = part[j % (DEPTH + 1)];
            #pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[ms][ns][r] = __builtin_fmaf(scale[ms], pr[r], acc[ms][ns][r]);
        }
        if (i < TOTAL) {
            if (i % NS == 0 && i / NS + 1 < MS)
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // the next A fragment's two ds_read_b128
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
        }
        if (i >= DEPTH)
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);         // its four promotion FMAs
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Fast path: K-major A and B, K % 128 == 0, 16-byte aligned rows.  LDS-DMA double buffer, one barrier per K block.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int PIPE>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_fast_kernel(const GemmParams p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_UNITS = BM / 8, B_UNITS = BN / 8;                  // 1 KiB LDS-DMA pieces (8 rows x 128 B)
    constexpr int A_ITERS = (A_UNITS + NW - 1) / NW, B_ITERS = (B_UNITS + NW - 1) / NW;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && NS % 2 == 0, "wave tile must be a multiple of 16 x 32");
    static_assert(128 % WN == 0 || WN % 128 == 0, "a wave must not straddle an SFB block unevenly");
    static_assert(WN <= 128, "one SFB value per wave");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int num_kb = p.k / 128;

    // Per-lane constants of the LDS-DMA source pattern: lane -> (row lane >> 3 of the piece, stored chunk lane & 7).
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    // Fragment read offsets.
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);

    MaskedWalk walk;
    const int num_launched = gridDim.x;
    for (int tile_id = blockIdx.x;; tile_id += num_launched) {
        const Tile t = get_tile<BM, BN>(p, tile_id, walk);
        if (!t.valid)
            break;

        const int64_t ad_group = (p.gemm_type == kMasked) ? t.group : 0;
        v4f acc[MS][NS];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                acc[ms][ns] = v4f{0.f, 0.f, 0.f, 0.f};

        if (t.m_end > t.m0) {
            const uint8_t* a_base = p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm;
            const uint8_t* b_base = p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn;
            const int m_clamp = t.m_end - 1 - t.m0;               // last loadable local row
            const int n_clamp = p.n - 1 - t.n0;

            int a_off[A_ITERS], b_off[B_ITERS];
            #pragma unroll
            for (int j = 0; j < A_ITERS; ++j) {
                const int row = imin((wave + NW * j) * 8 + piece_row, m_clamp);
                a_off[j] = row * static_cast<int>(p.a_sm) + src_chunk * 16;
            }
            #pragma unroll
            for (int j = 0; j < B_ITERS; ++j) {
                const int row = imin(b_row_perm<WN>((wave + NW * j) * 8 + piece_row), n_clamp);
                b_off[j] = row * static_cast<int>(p.b_sn) + src_chunk * 16;
            }

            auto issue_stage = [&](int stage, int kb) {
                uint8_t* stage_base = lds + stage * STAGE_BYTES;
                const uint8_t* a_k = a_base + kb * 128;
                const uint8_t* b_k = b_base + kb * 128;
                #pragma unroll
                for (int j = 0; j < A_ITERS; ++j) {
                    const int unit = wave + NW * j;
                    if (A_UNITS % NW == 0 || unit < A_UNITS)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(a_k + a_off[j]),
                            (__attribute__((address_space(3))) void*)(stage_base + unit * 1024), 16, 0, 0);
                }
                #pragma unroll
                for (int j = 0; j < B_ITERS; ++j) {
                    const int unit = wave + NW * j;
                    if (B_UNITS % NW == 0 || unit < B_UNITS)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void*)(b_k + b_off[j]),
                            (__attribute__((address_space(3))) void*)(stage_base + A_BYTES + unit * 1024), 16, 0, 0);
                }
            };

            // Scale pointers: one SFA value per lane per M-subtile, one SFB value per wave.
            const float* sfa_lane[MS];
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
                const int row = t.m0 + imin(wm * WM + ms * 16 + (lane & 15), m_clamp);
                sfa_lane[ms] = p.sfa + ad_group * p.sfa_sg + static_cast<int64_t>(row) * p.sfa_sm;
            }
            const float* sfb_wave = p.sfb + static_cast<int64_t>(t.group) * p.sfb_sg +
                                    static_cast<int64_t>((t.n0 + wn * WN) / 128) * p.sfb_sn;

            float sa_cur[MS], sa_nxt[MS];
            float sb_cur, sb_nxt = 0.f;
            issue_stage(0, 0);
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                sa_cur[ms] = sfa_lane[ms][0];
            sb_cur = sfb_wave[0];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            for (int kb = 0; kb < num_kb; ++kb) {
                const int cur = kb & 1;
                if (kb + 1 < num_kb) {
                    issue_stage(cur ^ 1, kb + 1);
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        sa_nxt[ms] = sfa_lane[ms][static_cast<int64_t>(kb + 1) * p.sfa_sk];
                    sb_nxt = sfb_wave[static_cast<int64_t>(kb + 1) * p.sfb_sk];
                }

                const uint8_t* a_tile = lds + cur * STAGE_BYTES + (wm * WM) * 128;
                const uint8_t* b_tile = lds + cur * STAGE_BYTES + A_BYTES + (wn * WN) * 128;
                if constexpr (PIPE > 0) {
                    float scale[MS];
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        scale[ms] = sa_cur[ms] * sb_cur;
                    compute_block_pipelined<MS, NS, PIPE>(a_tile, b_tile, frag_off, scale, acc);
                } else {
                    v8i bf[NS];
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        const v8i af = load_fragment(a_tile + ms * 2048, frag_off);
                        const float scale = sa_cur[ms] * sb_cur;
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns) {
                            const v4f part = mfma_fp8_k128(bf[ns], af);
                            acc[ms][ns] += scale * part;
                        }
                    }
                }

                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    sa_cur[ms] = sa_nxt[ms];
                sb_cur = sb_nxt;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }

        store_tile<MS, NS>(p, t, ad_group * p.d_sg, acc, t.m0 + wm * WM, t.n0 + wn * WN);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Ring kernel: the fast path's production form.  Same tiles, LDS image, LDS-DMA pieces and MFMA+promotion step as
// the pipe kernel above, but the global->LDS stream is never drained inside the K loop:
//   * A lives in a 3-slot ring, B in a 2-slot ring (256x256 tile: 3*32 + 2*32 KiB = all 160 KiB of the CU's LDS);
//   * ONE barrier P per K block, placed before the last M-subtile round (step TOTAL-NS), certifies "block kb+1 has
//     landed" (each wave first waits vmcnt(A_ITERS+B_ITERS): everything but its newest two batches of pieces) and
//     "block kb's A slot is dead"; behind it the wave issues the LDS-DMA of A(kb+3) into that slot and reads the
//     first fragments of block kb+1 (A subtile 0, and each B subtile right after its last MFMA of block kb), so the
//     matrix pipe does not see a restart bubble at the block boundary;
//   * a second, light barrier Q after the first M-subtile round certifies "every wave holds B(kb) in registers",
//     behind it the LDS-DMA of B(kb+2) goes into that slot.
//   A and B are therefore prefetched about two K blocks ahead; the per-row scales ride one block ahead in VGPRs,
//   loaded by inline-asm buffer loads so that hipcc (which would wait vmcnt(0) at their first use and drain the
//   LDS-DMA queue with them) never sees a VGPR-destination load in the loop.
// vmcnt bookkeeping (loads retire in order): issue order is ... SF(kb+1) | A(kb+2) x A_ITERS | B(kb+2) x B_ITERS |
// P_kb: wait vmcnt(A_ITERS+B_ITERS) => SF(kb+1), A(kb+1), B(kb+1) and everything older have landed.
// K tail: pieces and scale loads of blocks >= num_kb are still issued (the counts stay exact) with bit 31 set in
// their voffset, which the buffer descriptor's range check turns into a no-op.
// ---------------------------------------------------------------------------------------------------------------
template <int MS>
struct ScaleLanding { float sa[MS]; float sb; };

// SF loads for one K block: MS row scales (SFA is MN-major here: consecutive M-subtiles are 64 bytes apart) and the
// wave-uniform SFB value, all through buffer descriptors.  The destinations are NOT valid until wait_landing().
template <int MS>
__device__ __forceinline__ void issue_scale_loads(ScaleLanding<MS>& l, const v4i& sfa_rsrc, int sfa_voff,
                                                  const v4i& sfb_rsrc, int sfb_voff) {
    static_assert(MS == 2 || MS == 4 || MS == 8, "unrolled by hand");
    // s_nop 4 opening: the descriptor / soffset SGPRs may have been written by the immediately preceding SALU
    if constexpr (MS == 8) {
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dword %0, %9, %10, 0 offen\n\t"
            "buffer_load_dword %1, %9, %10, 0 offen offset:64\n\t"
            "buffer_load_dword %2, %9, %10, 0 offen offset:128\n\t"
            "buffer_load_dword %3, %9, %10, 0 offen offset:192\n\t"
            "buffer_load_dword %4, %9, %10, 0 offen offset:256\n\t"
            "buffer_load_dword %5, %9, %10, 0 offen offset:320\n\t"
            "buffer_load_dword %6, %9, %10, 0 offen offset:384\n\t"
            "buffer_load_dword %7, %9, %10, 0 offen offset:448\n\t"
            "buffer_load_dword %8, %11, %12, 0 offen"
            : "=&v"(l.sa[0]), "=&v"(l.sa[1]), "=&v"(l.sa[2]), "=&v"(l.sa[3]), "=&v"(l.sa[4]), "=&v"(l.sa[5]),
              "=&v"(l.sa[6]), "=&v"(l.sa[7]), "=&v"(l.sb)
            : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff), "s"(sfb_rsrc)
            : "memory");
    } else if constexpr (MS == 4) {
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dword %0, %5, %6, 0 offen\n\t"
            "buffer_load_dword %1, %5, %6, 0 offen offset:64\n\t"
            "buffer_load_dword %2, %5, %6, 0 offen offset:128\n\t"
            "buffer_load_dword %3, %5, %6, 0 offen offset:192\n\t"
            "buffer_load_dword %4, %7, %8, 0 offen"
            : "=&v"(l.sa[0]), "=&v"(l.sa[1]), "=&v"(l.sa[2]), "=&v"(l.sa[3]), "=&v"(l.sb)
            : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff), "s"(sfb_rsrc)
            : "memory");
    } else {
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dword %0, %3, %4, 0 offen\n\t"
            "buffer_load_dword %1, %3, %4, 0 offen offset:64\n\t"
            "buffer_load_dword %2, %5, %6, 0 offen"
            : "=&v"(l.sa[0]), "=&v"(l.sa[1]), "=&v"(l.sb)
            : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff), "s"(sfb_rsrc)
            : "memory");
    }
}

// Waits until at most ALLOWED vector-memory operations of this wave are outstanding and all its LDS reads have
// returned; names the landing registers so that no consumer of them can be scheduled above the wait.
template <int ALLOWED, int MS>
__device__ __forceinline__ void wait_landing(ScaleLanding<MS>& l) {
    static_assert(ALLOWED >= 0 && ALLOWED < 64, "vmcnt is a 6-bit counter");
    if constexpr (MS == 8)
        asm volatile("s_waitcnt vmcnt(%c9) lgkmcnt(0)"
                     : "+v"(l.sa[0]), "+v"(l.sa[1]), "+v"(l.sa[2]), "+v"(l.sa[3]), "+v"(l.sa[4]), "+v"(l.sa[5]),
                       "+v"(l.sa[6]), "+v"(l.sa[7]), "+v"(l.sb)
                     : "i"(ALLOWED) : "memory");
    else if constexpr (MS == 4)
        asm volatile("s_waitcnt vmcnt(%c5) lgkmcnt(0)"
                     : "+v"(l.sa[0]), "+v"(l.sa[1]), "+v"(l.sa[2]), "+v"(l.sa[3]), "+v"(l.sb)
                     : "i"(ALLOWED) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%c3) lgkmcnt(0)" : "+v"(l.sa[0]), "+v"(l.sa[1]), "+v"(l.sb) : "i"(ALLOWED) : "memory");
}


// RABL (timing experiments only, results are garbage): 1 = no barriers, 2 = no LDS-DMA pieces in the K loop,
// 3 = every piece re-reads K block 0 (L2-resident source: isolates HBM / L2-miss effects from issue and LDS-write cost).
// PAD: idle issue cycles appended to every MFMA step (s_nop), a pacing knob.
template <int BM, int BN, int WAVES_M, int WAVES_N, int RABL = 0, int PAD = 0>
__device__ __forceinline__ void ring_kernel_body(const GemmParams& p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16;
    constexpr int TOTAL = MS * NS, DEPTH = 3;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, A_SLOTS = 3, B_SLOTS = 2;
    constexpr int B_BASE = A_SLOTS * A_BYTES, LDS_BYTES = B_BASE + B_SLOTS * B_BYTES;
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN / 8 / NW;
    constexpr int P_STEP = TOTAL - NS;                 // barrier P sits in front of this step
    constexpr unsigned OOB = 0x80000000u;        // voffset bit that sends a buffer access out of range
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave issues the same number of LDS-DMA pieces");
    static_assert(WM % 16 == 0 && WN % 16 == 0 && NS % 2 == 0 && MS % 2 == 0 && MS >= 2, "wave tile shape");
    static_assert(WN <= 128 && 128 % WN == 0, "one SFB value per wave");
    static_assert(TOTAL >= 2 * NS && TOTAL - DEPTH >= TOTAL - NS, "the ring tail must lie within the last M-subtile");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((NW * 8) % 16 == 0 && ((NW * 8) % WN == 0 || WN % (NW * 8) == 0),
                  "the row permutation of a B piece must be lane-independent");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int num_kb = p.k / 128;
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);
    // Per-lane byte offsets of a piece's source rows (row part and 16-byte chunk) -- all in the VOFFSET, which is the
    // part of a buffer address the descriptor range-checks; only the K block offset travels in the soffset.
    const int a_voff = (wave * 8 + piece_row) * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
    const long long t_entry = p.dbg != nullptr ? __builtin_amdgcn_s_memtime() : 0;
    long long t_loop0 = 0, t_loop1 = 0;

    MaskedWalk walk;
    const int num_launched = gridDim.x;
    for (int tile_id = blockIdx.x;; tile_id += num_launched) {
        const Tile t = get_tile<BM, BN>(p, tile_id, walk);
        if (!t.valid)
            break;
        const int64_t ad_group = (p.gemm_type == kMasked) ? t.group : 0;

        float acc[MS][NS][4];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                #pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[ms][ns][r] = 0.f;

        if (t.m_end > t.m0) {
            const uint8_t* a_base = p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm;
            const uint8_t* b_base = p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn;
            const int a_rows = imin(t.m_end - t.m0, BM), b_rows = imin(p.n - t.n0, BN);
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base), 0,
                                                                  (a_rows - 1) * lda + p.k, 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base), 0,
                                                                  (b_rows - 1) * ldb + p.k, 0x00020000);
            // Scale descriptors as plain 4 x 32-bit words (inline-asm "s" operands).
            const float* sfa_group = p.sfa + ad_group * p.sfa_sg;
            const float* sfb_wave = p.sfb + static_cast<int64_t>(t.group) * p.sfb_sg +
                                    static_cast<int64_t>((t.n0 + wn * WN) / 128) * p.sfb_sn;
            const int sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kb_stride = static_cast<int>(p.sfb_sk) * 4;
            const int sfa_extent = (p.m - 1) * 4 + (num_kb - 1) * sfa_kb_stride + 4;
            const int sfb_extent = (num_kb - 1) * sfb_kb_stride + 4;
            const uint64_t sfa_addr = reinterpret_cast<uint64_t>(sfa_group), sfb_addr = reinterpret_cast<uint64_t>(sfb_wave);
            const v4i sfa_rsrc = {__builtin_amdgcn_readfirstlane(static_cast<int>(sfa_addr)),
                                  __builtin_amdgcn_readfirstlane(static_cast<int>(sfa_addr >> 32) & 0xffff),
                                  __builtin_amdgcn_readfirstlane(sfa_extent), 0x00020000};
            const v4i sfb_rsrc = {__builtin_amdgcn_readfirstlane(static_cast<int>(sfb_addr)),
                                  __builtin_amdgcn_readfirstlane(static_cast<int>(sfb_addr >> 32) & 0xffff),
                                  __builtin_amdgcn_readfirstlane(sfb_extent), 0x00020000};
            const int sfa_voff = (t.m0 + wm * WM + (lane & 15)) * 4;

            // One LDS-DMA piece of K block j: A piece q -> rows (wave + NW q) * 8 ... + 7 of A slot j % 3.
            auto issue_a_piece = [&](int slot_off, int j, int q) {
                const int unit = wave + NW * q;
                const int voff = static_cast<int>(static_cast<unsigned>(a_voff) +
                                                  (static_cast<unsigned>(q * (NW * 8) * lda) | (j < num_kb ? 0u : OOB)));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    a_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + unit * 1024), 16, voff,
                    RABL == 3 ? 0 : j * 128, 0, 0);
            };
            auto issue_b_piece = [&](int slot_off, int j, int q) {
                const int unit = wave + NW * q;
                const int voff = static_cast<int>(static_cast<unsigned>(b_voff) +
                                                  (static_cast<unsigned>(b_row_perm<WN>(q * (NW * 8)) * ldb) | (j < num_kb ? 0u : OOB)));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    b_rsrc, (__attribute__((address_space(3))) void*)(lds + B_BASE + slot_off + unit * 1024), 16, voff,
                    RABL == 3 ? 0 : j * 128, 0, 0);
            };
            auto issue_scales = [&](ScaleLanding<MS>& l, int j) {
                if constexpr (RABL == 5) {      // trace build: no scale traffic at all, constant scales
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) l.sa[ms] = 1.f;
                    l.sb = 1.f;
                    return;
                }
                const unsigned oob = j < num_kb ? 0u : OOB;
                const int jj = (RABL == 4) ? 0 : j;          // RABL 4: scales always from K block 0 (cache resident)
                issue_scale_loads<MS>(l, sfa_rsrc, static_cast<int>(static_cast<unsigned>(sfa_voff + jj * sfa_kb_stride) | oob),
                                      sfb_rsrc, static_cast<int>(static_cast<unsigned>(jj * sfb_kb_stride) | oob));
            };

            float scale[MS], scale_tail = 0.f;
            ScaleLanding<MS> land;
            v4f part[DEPTH + 1];
            #pragma unroll
            for (int i = 0; i <= DEPTH; ++i)
                part[i] = v4f{0.f, 0.f, 0.f, 0.f};

            // ---- prologue: A(0) B(0) A(1) B(1) | SF(0), full drain (the scale loads must reach their wait in straight-line
            // code: hipcc may copy their destination registers at any control-flow join in between) ----
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece(0, 0, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS; ++q) issue_b_piece(0, 0, q);
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece(A_BYTES, 1, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS; ++q) issue_b_piece(B_BYTES, 1, q);
            issue_scales(land, 0);
            wait_landing<0, MS>(land);
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                scale[ms] = land.sa[ms] * land.sb;
            raw_barrier();
            [[maybe_unused]] int trace_v = 0;                          // RABL 5: lane i = s_memtime at step i of K block 30/31
            [[maybe_unused]] long long trace_t[4] = {0, 0, 0, 0};

            // slot offsets (bytes): a_cur is being computed (and re-filled behind barrier P), *_nxt is read behind P
            int a_cur = 0, a_nxt = A_BYTES, b_nxt = B_BYTES;
            v8i bf[NS], af[2];
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                bf[ns] = load_fragment(lds + B_BASE + (wn * WN + ns * 16) * 128, frag_off);
            af[0] = load_fragment(lds + (wm * WM) * 128, frag_off);

            // Issue order per block (vmcnt counts depend on it): SF(kb+1) [top of block kb] | A(kb+2) x A_ITERS
            // [steps 0, 2, ..] | B(kb+2) x B_ITERS [behind Q] | P_kb waits vmcnt(A_ITERS + B_ITERS).
            constexpr int B_FIRST = (NS > 2 * A_ITERS ? NS : 2 * A_ITERS);
            static_assert(B_FIRST + 2 * (B_ITERS - 1) < P_STEP, "LDS-DMA pieces must be issued in front of barrier P");
            if (p.dbg != nullptr) t_loop0 = __builtin_amdgcn_s_memtime();
            for (int kb = 0; kb < num_kb; ++kb) {
                const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                const uint8_t* a_next_tile = lds + a_nxt + (wm * WM) * 128;
                const uint8_t* b_next_tile = lds + B_BASE + b_nxt + (wn * WN) * 128;
                // A(kb+2) goes into the slot that held A(kb-1): the one after a_nxt in ring order
                const int a_fill = (a_nxt == (A_SLOTS - 1) * A_BYTES) ? 0 : a_nxt + A_BYTES;
                issue_scales(land, kb + 1);         // consumed at the end of this iteration, behind P's wait

                #pragma unroll
                for (int i = 0; i < TOTAL; ++i) {
                    const int ms = i / NS, ns = i % NS;
                    const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;     // step being promoted
                    const int jms = j / NS, jns = j % NS;
                    const float jscale = (i >= DEPTH) ? scale[jms] : scale_tail;
                    if constexpr (RABL == 5) {
                        asm volatile("s_memtime %0" : "=s"(trace_t[i & 3]));
                        if (i >= 2) {
                            const int lane_sel = (kb == 30) ? (i - 2) : ((kb == 31) ? (i + 30) : 63);
                            asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(trace_v)
                                         : "s"(static_cast<int>(trace_t[(i - 2) & 3])), "s"(lane_sel));
                        }
                    }
                    if (i == P_STEP) {
                        // barrier P: block kb+1 (and its scales) landed everywhere; every read of A(kb) has returned
                        wait_landing<(RABL == 2 ? 0 : A_ITERS + B_ITERS), MS>(land);
                        if (RABL != 1) raw_barrier();
                    }
                    if (ns == 0) {
                        if (ms + 1 < MS)
                            af[(ms + 1) & 1] = load_fragment(a_tile + (ms + 1) * 2048, frag_off);
                        else
                            af[(ms + 1) & 1] = load_fragment(a_next_tile, frag_off);
                    }
                    mfma_promote_step(part[i & DEPTH], bf[ns], af[ms & 1], acc[jms][jns], jscale, part[(i + 1) & DEPTH]);
                    if constexpr (PAD > 0) asm volatile("s_nop %c0" :: "i"(PAD - 1));
                    if (ms == MS - 1)
                        bf[ns] = load_fragment(b_next_tile + ns * 2048, frag_off);
                    if (RABL != 2 && i % 2 == 0 && i / 2 < A_ITERS)
                        issue_a_piece(a_fill, kb + 2, i / 2);
                    if (RABL != 1 && i == NS - 1)
                        raw_barrier();                                          // barrier Q: B(kb) is in registers
                    if (RABL != 2 && i >= B_FIRST && (i - B_FIRST) % 2 == 0 && (i - B_FIRST) / 2 < B_ITERS)
                        issue_b_piece(b_nxt ^ B_BYTES, kb + 2, (i - B_FIRST) / 2);   // B(kb)'s slot
                }
                // scales of block kb+1 (landed before P); then their landing registers take SF(kb+2)
                scale_tail = scale[MS - 1];
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    scale[ms] = land.sa[ms] * land.sb;
                    pin_vgpr(scale[ms]);
                }
                a_cur = a_nxt;
                a_nxt = a_fill;
                b_nxt ^= B_BYTES;
            }
            if (p.dbg != nullptr) t_loop1 = __builtin_amdgcn_s_memtime();
            if constexpr (RABL == 5)
                if (p.dbg != nullptr)
                    reinterpret_cast<int*>(p.dbg + 8192)[(blockIdx.x * NW + wave) * 64 + lane] = trace_v;
            // drain: the LDS-DMA no-ops of the K tail, then the last three promotions
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            #pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                const int j = TOTAL - DEPTH + i;
                promote_only(acc[j / NS][j % NS], scale_tail, part[(TOTAL + i + 1) & DEPTH]);
            }
            __syncthreads();        // the next tile's prologue rewrites slots other waves may still be reading
        }

        v4f out[MS][NS];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                out[ms][ns] = v4f{acc[ms][ns][0], acc[ms][ns][1], acc[ms][ns][2], acc[ms][ns][3]};
        store_tile<MS, NS>(p, t, ad_group * p.d_sg, out, t.m0 + wm * WM, t.n0 + wn * WN);
        if (p.dbg != nullptr && tile_id == blockIdx.x) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_stamp(p, NW, 0, t_entry);
            dbg_stamp(p, NW, 1, t_loop0);
            dbg_stamp(p, NW, 2, t_loop1);
            dbg_stamp(p, NW, 3, __builtin_amdgcn_s_memtime());
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int RABL = 0, int PAD = 0>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_ring_kernel(const GemmParams p) {
    ring_kernel_body<BM, BN, WAVES_M, WAVES_N, RABL, PAD>(p);
}


// DABL (timing experiments only): 1 = no stagger between the wave halves, 2 = no s_setprio around the matrix segments, ...
template <int BM, int BN, int WAVES_M, int WAVES_N, int DABL = 0>
__device__ __forceinline__ void duo_abl_kernel_body(const GemmParams& p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16, HS = MS / 2;
    constexpr int TOTAL = MS * NS, SEG = HS * NS, DEPTH = 3;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, A_SLOTS = 3, B_SLOTS = 2;
    constexpr int B_BASE = A_SLOTS * A_BYTES, LDS_BYTES = B_BASE + B_SLOTS * B_BYTES;
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN / 8 / NW;
    // Early barriers (experiment): a wave in a matrix segment arrives at the segment-end barrier EB steps before its last
    // MFMA, so the partner half starts its matrix segment while this one still has EB MFMAs to issue -- the matrix pipe
    // does not idle for the barrier round trip.  (The barrier in front of a load segment only orders LDS traffic, which the
    // trailing register-only steps do not touch.)
    constexpr int EB = (DABL == 14) ? 2 : (DABL == 15 ? 4 : (DABL == 16 ? 1 : 0));
    constexpr int A_EARLY = (DABL == 9) ? 0 : (DABL == 17 ? A_ITERS : (DABL == 18 ? A_ITERS * 3 / 4 : A_ITERS / 2));
    // MP (experiment): this many LDS-DMA pieces per matrix segment ride between its MFMA steps (A pieces in M_a, B pieces in
    // M_b) instead of in L_b, the longest load segment
    constexpr int MP = (DABL == 30) ? 1 : (DABL == 31 ? 2 : 0);
    static_assert(MP == 0 || (A_ITERS - A_EARLY >= MP && B_ITERS >= MP && SEG >= 12), "pieces to move");        // A pieces issued in L_a (next to the scale loads); the rest go with B in L_b
    constexpr unsigned OOB = 0x80000000u;
    // DABL 4: matrix segments and barriers only; 5: no LDS-DMA in the loop; 6: no fragment reads in the loop; 7: no scale loads
    constexpr bool PERSIST = (DABL == 20 || DABL == 26 || DABL == 41);      // persistent launch with cross-tile prologue prefetch
    // B_MN: operand B is MN-major ([K][N], unit stride along n, row pitch b_sk): the nn / tn layouts without the re-majoring
    // pass.  LDS-DMA pieces are 4 k-rows x 256 bytes, B fragments come through the hardware transpose read, B rows keep
    // their natural order (=> 8-byte instead of 16-byte BF16 stores).  See load_fragment_tr.
    constexpr bool B_MN = (DABL == 40 || DABL == 41);
    static_assert(!B_MN || (BN == 256 && NW == 8), "MN-major B tile: 128 k-rows x 256 bytes, 32 pieces over 8 waves");
    constexpr bool TRACE = (DABL == 3 || DABL == 10 || DABL == 11), NOPRIO = (DABL != 2 && DABL != 3 && DABL != 24 && DABL != 26 && DABL != 28),
                   LOADPRIO = (DABL == 8 || DABL == 11 || DABL == 25 || DABL == 29);
    constexpr bool NO_DMA = (DABL == 4 || DABL == 5), NO_LDS_READS = (DABL == 4 || DABL == 6), NO_SCALES = (DABL == 4 || DABL == 7);
    static_assert(NW % 2 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave issues the same number of pieces");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile shape");
    static_assert(WN <= 128 && 128 % WN == 0, "one SFB value per wave");
    static_assert(SEG > DEPTH, "the promotion ring must fit in a segment");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((NW * 8) % 16 == 0 && ((NW * 8) % WN == 0 || WN % (NW * 8) == 0),
                  "the row permutation of a B piece must be lane-independent");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool upper_half = wave >= NW / 2;
    const int num_kb = p.k / 128;
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);
    // A rows are interleaved inside a wave's WM rows: LDS row position P holds tile row a_row_of(P).  For the rows of one
    // piece (P = 8 u + j) that is a_unit_row(u) + j * MS: the lane part goes into a_voff, the unit part is wave-uniform.
    auto a_unit_row = [](int u) { return (u / (WM / 8)) * WM + (u & 1) * 8 * MS + ((u % (WM / 8)) >> 1); };
    const int a_voff = piece_row * MS * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
    const long long t_entry = p.dbg != nullptr ? __builtin_amdgcn_s_memtime() : 0;
    long long t_loop0 = 0, t_loop1 = 0;

    MaskedWalk walk;
    const int num_launched = gridDim.x;
    const int sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kb_stride = static_cast<int>(p.sfb_sk) * 4;
    const int sfa_extent = (p.m - 1) * 4 + (num_kb - 1) * sfa_kb_stride + 4;
    const int sfb_extent = (num_kb - 1) * sfb_kb_stride + 4;

    // Per-piece source offsets (rows + chunk: the bounds-checked part of the address) are kernel invariants held in
    // VGPRs; the K block goes in the soffset.  Blocks past the end re-read the last K block into a dead slot -- no
    // out-of-range arithmetic in the loop, the vmcnt counts stay exact, the bytes come from L2.
    int a_piece_voff[A_ITERS], b_piece_voff[B_ITERS];
    #pragma unroll
    for (int q = 0; q < A_ITERS; ++q)
        a_piece_voff[q] = a_voff + a_unit_row(wave + NW * q) * lda;
    #pragma unroll
    for (int q = 0; q < B_ITERS; ++q)
        b_piece_voff[q] = b_voff + b_row_perm<WN>(q * (NW * 8)) * ldb;
    // MN-major B: lane l of piece u carries k-row 4u + (l >> 4), source chunk (l & 15) ^ f(k); u = wave + 8q => f lane-constant
    const int ldb_mn = static_cast<int>(p.b_sk);
    const int bmn_voff = (lane >> 4) * ldb_mn + ((((lane & 15) ^ (((4 * (wave & 1) + (lane >> 4)) & 7) | (((wave >> 2) & 1) << 3)))) << 4);
    const int tr_lane_base = (16 * (lane >> 4) + ((lane & 15) >> 1)) * 256 + (lane & 1) * 8;
    const int tr_swz = ((lane & 15) >> 1) | (((lane >> 4) & 1) << 3);

    // Addresses of one tile (plain scalars; the buffer descriptors are built from them where they are used).
    struct TileMem { const uint8_t* a_base; const uint8_t* b_base; int a_bytes, b_bytes; uint64_t sfa_addr, sfb_addr; int sfa_voff; };
    auto tile_mem = [&](const Tile& tt) {
        const int64_t adg = (p.gemm_type == kMasked) ? tt.group : 0;
        TileMem tm;
        // every descriptor input goes through readfirstlane: tile coordinates that depend on loaded values (grouped
        // layouts) are uniform in fact but not provably so, and hipcc would wrap each buffer op in a waterfall loop
        auto uniform_ptr = [](const uint8_t* ptr) {
            const uint64_t v = reinterpret_cast<uint64_t>(ptr);
            const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v));
            const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
            return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
        };
        tm.a_base = uniform_ptr(p.a + adg * p.a_sg + static_cast<int64_t>(tt.m0) * p.a_sm);
        tm.b_base = uniform_ptr(p.b + static_cast<int64_t>(tt.group) * p.b_sg + static_cast<int64_t>(tt.n0) * (B_MN ? 1 : p.b_sn));
        tm.a_bytes = __builtin_amdgcn_readfirstlane((imin(tt.m_end - tt.m0, BM) - 1) * lda + p.k);
        tm.b_bytes = __builtin_amdgcn_readfirstlane(B_MN ? (p.k - 1) * ldb_mn + (p.n - tt.n0) : (imin(p.n - tt.n0, BN) - 1) * ldb + p.k);
        tm.sfa_addr = reinterpret_cast<uint64_t>(p.sfa + adg * p.sfa_sg);
        tm.sfb_addr = reinterpret_cast<uint64_t>(p.sfb + static_cast<int64_t>(tt.group) * p.sfb_sg +
                                                 static_cast<int64_t>((tt.n0 + wn * WN) / 128) * p.sfb_sn);
        tm.sfa_voff = (tt.m0 + wm * WM + (lane & 15) * MS) * 4;
        return tm;
    };
    auto scale_rsrc = [&](uint64_t addr, int extent) {
        return v4i{__builtin_amdgcn_readfirstlane(static_cast<int>(addr)),
                   __builtin_amdgcn_readfirstlane(static_cast<int>(addr >> 32) & 0xffff),
                   __builtin_amdgcn_readfirstlane(extent), 0x00020000};
    };
    auto issue_a_piece_r = [&](const uint8_t* base, int bytes, int slot_off, int j, int q) {
        const int unit = wave + NW * q;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, bytes, 0x00020000), (__attribute__((address_space(3))) void*)(lds + slot_off + unit * 1024), 16, a_piece_voff[q],
            (DABL == 32 ? 0 : imin(j, num_kb - 1)) * 128, 0, 0);       // DABL 32 (timing): every piece re-reads K block 0 (L2 resident)
    };
    auto issue_b_piece_r = [&](const uint8_t* base, int bytes, int slot_off, int j, int q) {
        const int unit = wave + NW * q;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, bytes, 0x00020000), (__attribute__((address_space(3))) void*)(lds + B_BASE + slot_off + unit * 1024), 16,
            B_MN ? bmn_voff : b_piece_voff[q],
            B_MN ? (imin(j, num_kb - 1) * 128 + 4 * unit) * ldb_mn : (DABL == 32 ? 0 : imin(j, num_kb - 1)) * 128, 0, 0);
    };
    // Prologue pieces of a tile: A(0) B(0) A(1) B(1) into ring slots 0 / 1.  Issued at kernel entry for the first tile
    // and, in the persistent launch, for tile i+1 as soon as tile i's K loop has released the LDS -- i.e. BEFORE tile i's
    // output stores, so that the cold-start latency of a tile and its predecessor's store tail overlap.  Only LDS-DMA
    // travels ahead: a VGPR-destination load (the scales) must reach its wait in straight-line code, because hipcc is
    // free to copy the destination registers at any control-flow join in between -- before the data has arrived.
    ScaleLandingV<MS> land;
    auto issue_prologue = [&](const Tile& tt) {
        const TileMem tm = tile_mem(tt);
        #pragma unroll
        for (int q = 0; q < A_ITERS; ++q) issue_a_piece_r(tm.a_base, tm.a_bytes, 0, 0, q);
        #pragma unroll
        for (int q = 0; q < B_ITERS; ++q) issue_b_piece_r(tm.b_base, tm.b_bytes, 0, 0, q);
        #pragma unroll
        for (int q = 0; q < A_ITERS; ++q) issue_a_piece_r(tm.a_base, tm.a_bytes, A_BYTES, 1, q);
        #pragma unroll
        for (int q = 0; q < B_ITERS; ++q) issue_b_piece_r(tm.b_base, tm.b_bytes, B_BYTES, 1, q);
    };

    // Tile iteration state: (tile_id, pass); contiguous layout with BM = 2 x alignment: a tile whose halves belong to two
    // groups is walked twice.
    int tile_id = blockIdx.x, pass = 0;
    bool prefetched = false, first_tile = true;
    Tile t = get_tile<BM, BN>(p, tile_id, walk, pass);
    while (t.valid) {
        const int64_t ad_group = (p.gemm_type == kMasked) ? t.group : 0;
        Tile tn;
        bool next_prefetched = false;
        auto fetch_next = [&]() {
            if (t.second_pass) {
                pass = 1;
            } else {
                tile_id += num_launched;
                pass = 0;
            }
            tn = get_tile<BM, BN>(p, tile_id, walk, pass);
            if (PERSIST && tn.valid && tn.m_end > tn.m0) {
                // The next tile's first two K blocks and the scales of its block 0, drained HERE -- in front of this tile's
                // output stores: once stores are pending they count towards vmcnt, and waiting for "block 0 has landed" at
                // the top of the next tile would wait for (nearly) all of them.  Drained now, the next tile starts without
                // any wait and the stores overlap its first K block instead of standing between the two tiles.
                const TileMem tmn = tile_mem(tn);
                issue_prologue(tn);
                issue_scale_loads_v<MS>(land, scale_rsrc(tmn.sfa_addr, sfa_extent), tmn.sfa_voff, scale_rsrc(tmn.sfb_addr, sfb_extent), 0);
                wait_landing_v<0, MS>(land);        // the landed values stay in `land` until the next tile's L_a(0) consumes them
                next_prefetched = true;
            }
        };

        float acc[MS][NS][4];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                #pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[ms][ns][r] = 0.f;

        if (t.m_end > t.m0) {
            const TileMem tm = tile_mem(t);
            const v4i sfa_rsrc = scale_rsrc(tm.sfa_addr, sfa_extent), sfb_rsrc = scale_rsrc(tm.sfb_addr, sfb_extent);
            const int sfa_voff = tm.sfa_voff;
            auto issue_a_piece = [&](int slot_off, int j, int q) { issue_a_piece_r(tm.a_base, tm.a_bytes, slot_off, j, q); };
            auto issue_b_piece = [&](int slot_off, int j, int q) { issue_b_piece_r(tm.b_base, tm.b_bytes, slot_off, j, q); };
            auto issue_scales = [&](ScaleLandingV<MS>& l, int j) {
                const int jj = (DABL == 21) ? 0 : imin(j, num_kb - 1);   // past the end: the last block's scales again (never consumed); DABL 21 (timing): always block 0 = cache resident
                issue_scale_loads_v<MS>(l, sfa_rsrc, sfa_voff + jj * sfa_kb_stride, sfb_rsrc, jj * sfb_kb_stride);
            };

            float scale[MS], scale_tail = 0.f;
            v4f part[DEPTH + 1];
            #pragma unroll
            for (int i = 0; i <= DEPTH; ++i)
                part[i] = v4f{0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                scale[ms] = 0.f;

            // ---- block 0 and its scales must land before the first segment ----
            if (!prefetched) {
                // SF(0) is the newest vector-memory operation: a full drain, which also lands block 1 -- a fraction of a
                // microsecond once per tile.  (Straight-line from the scale loads to their wait: hipcc may copy the landing
                // registers at any control-flow join in between.)
                issue_prologue(t);
                issue_scales(land, 0);
                wait_landing_v<0, MS>(land);
            }
            raw_barrier();
            if (DABL != 1 && upper_half)
                raw_barrier();                      // the upper half runs one segment behind from here on

            [[maybe_unused]] int trace_v = 0;
            auto stamp = [&](int kb, int k) {
                if constexpr (TRACE) {      // trace build: lane 8 * (kb - 28) + k = s_memtime, for K blocks 28 .. 35
                    long long tt;
                    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt));
                    const int lane_sel = (kb >= 28 && kb < 32) ? (kb - 28) * 8 + k : 63;
                    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(trace_v) : "s"(static_cast<int>(tt)), "s"(lane_sel));
                }
            };
            int a_cur = 0, a_fill = 2 * A_BYTES, b_cur = 0;     // slots of A(kb), A(kb+2) [= A(kb-1)'s], B(kb)
            v8i bf[NS], af[HS];
            if (p.dbg != nullptr) t_loop0 = __builtin_amdgcn_s_memtime();
            if (EB > 0) raw_barrier();                          // L_a(0)'s barrier; later ones sit inside M_b

            for (int kb = 0; kb < num_kb; ++kb) {
                const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                const uint8_t* b_tile = lds + B_BASE + b_cur + (wn * WN) * 128;

                // ---------------- L_a ----------------
                stamp(kb, 0);
                if (EB == 0) raw_barrier();         // EB > 0: executed inside the previous matrix segment / before the loop
                stamp(kb, 1);
                // fragment reads first: they complete in the shadow of the slow vector-memory issue that follows
                [[maybe_unused]] FragTr bfq[NS];
                if (NO_LDS_READS ? kb == 0 : true) {
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        if constexpr (B_MN)
                            bfq[ns] = load_fragment_tr(lds + B_BASE + b_cur, tr_lane_base, ((wn * (WN / 16) + ns) ^ tr_swz) << 4);
                        else
                            bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                    }
                    #pragma unroll
                    for (int h = 0; h < HS; ++h)
                        af[h] = load_fragment(a_tile + h * 2048, frag_off);
                }
                [[maybe_unused]] long long t_in[3];
                if constexpr (TRACE) asm volatile("s_memtime %0" : "=s"(t_in[0]) :: "memory");
                scale_tail = scale[MS - 1];
                // block kb's scales landed before the previous L_b's wait (block 0: before the prologue's / the prefetch's)
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    scale[ms] = land.q[ms / 4][ms % 4] * land.sb;
                    pin_vgpr(scale[ms]);
                }
                if constexpr (TRACE) asm volatile("s_memtime %0" : "=s"(t_in[1]) :: "memory");
                if (!NO_SCALES) issue_scales(land, kb + 1);
                if (!NO_DMA) {
                    #pragma unroll
                    for (int q = 0; q < A_EARLY; ++q)
                        issue_a_piece(a_fill, kb + 2, q);
                }
                if constexpr (TRACE) asm volatile("s_memtime %0" : "=s"(t_in[2]) :: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (B_MN) {
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        bf[ns] = assemble_fragment_tr(bfq[ns]);
                }
                if constexpr (TRACE) {
                    #pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int lane_sel = (kb >= 28 && kb < 32) ? 32 + (kb - 28) * 4 + q : 63;
                        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(trace_v) : "s"(static_cast<int>(t_in[q])), "s"(lane_sel));
                    }
                }

                // ---------------- M_a ----------------
                stamp(kb, 2);
                raw_barrier();
                stamp(kb, 3);
                if (!NOPRIO && !LOADPRIO) __builtin_amdgcn_s_setprio(1);
                if (LOADPRIO) __builtin_amdgcn_s_setprio(0);
                #pragma unroll
                for (int i = 0; i < SEG; ++i) {
                    const int ns = i % NS, h = i / NS;
                    const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;
                    const float jscale = (i >= DEPTH) ? scale[j / NS] : scale_tail;
                    if (EB > 0 && i == SEG - EB) raw_barrier();
                    mfma_promote_step(part[i & DEPTH], bf[ns], af[h], acc[j / NS][j % NS], jscale, part[(i + 1) & DEPTH]);
                    if (MP > 0 && (i == 3 || (MP > 1 && i == 9)))
                        issue_a_piece(a_fill, kb + 2, A_EARLY + (i == 3 ? 0 : 1));
                }
                if (!NOPRIO && !LOADPRIO) __builtin_amdgcn_s_setprio(0);
                if (LOADPRIO) __builtin_amdgcn_s_setprio(1);

                // ---------------- L_b ----------------
                stamp(kb, 4);
                if (EB == 0) raw_barrier();
                stamp(kb, 5);
                if (!NO_LDS_READS) {
                    #pragma unroll
                    for (int h = 0; h < HS; ++h)
                        af[h] = load_fragment(a_tile + (HS + h) * 2048, frag_off);
                }
                if (!NO_DMA) {
                    #pragma unroll
                    for (int q = A_EARLY + MP; q < A_ITERS; ++q)
                        issue_a_piece(a_fill, kb + 2, q);
                    #pragma unroll
                    for (int q = 0; q < B_ITERS - MP; ++q)
                        issue_b_piece(b_cur, kb + 2, q);
                }
                // Block kb+1 and its scales: my pieces have landed.  (Persistent launch: a predecessor tile's output stores may
                // still be pending in the first K block.  They count towards vmcnt too, which can only make this wait
                // stricter -- loads retire in order among themselves, so "at most 8 operations outstanding" still implies
                // "every load but the newest 8 has landed".)
                wait_landing_v<(NO_DMA ? 0 : A_ITERS + B_ITERS - MP), MS>(land);       // MP pieces of this block come in M_b
                #pragma unroll
                for (int h = 0; h < HS; ++h)
                    asm volatile("" : "+v"(af[h]) :: "memory");

                // ---------------- M_b ----------------
                stamp(kb, 6);
                raw_barrier();
                stamp(kb, 7);
                if (!NOPRIO && !LOADPRIO) __builtin_amdgcn_s_setprio(1);
                if (LOADPRIO) __builtin_amdgcn_s_setprio(0);
                #pragma unroll
                for (int i2 = 0; i2 < SEG; ++i2) {
                    const int i = SEG + i2;
                    const int ns = i % NS, h = i2 / NS;
                    const int j = i - DEPTH;
                    if (EB > 0 && i2 == SEG - EB) raw_barrier();      // the next K block's L_a barrier
                    mfma_promote_step(part[i & DEPTH], bf[ns], af[h], acc[j / NS][j % NS], scale[j / NS], part[(i + 1) & DEPTH]);
                    if (MP > 0 && (i2 == 3 || (MP > 1 && i2 == 9)))
                        issue_b_piece(b_cur, kb + 2, B_ITERS - MP + (i2 == 3 ? 0 : 1));
                }
                if (!NOPRIO && !LOADPRIO) __builtin_amdgcn_s_setprio(0);
                if (LOADPRIO) __builtin_amdgcn_s_setprio(1);

                const int a_next = (a_cur == (A_SLOTS - 1) * A_BYTES) ? 0 : a_cur + A_BYTES;
                a_fill = a_cur;             // A(kb+3) will take the slot block kb just finished with
                a_cur = a_next;
                b_cur ^= B_BYTES;
            }
            if (DABL != 1 && !upper_half)
                raw_barrier();              // pairs with the barrier in front of the upper half's last segment
            if (p.dbg != nullptr) t_loop1 = __builtin_amdgcn_s_memtime();
            if constexpr (TRACE)
                if (p.dbg != nullptr)
                    reinterpret_cast<int*>(p.dbg + 8192)[(blockIdx.x * NW + wave) * 64 + lane] = trace_v;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the tail's re-read pieces: the ring is about to be reused
            __syncthreads();                                    // every wave is done with the LDS
            fetch_next();                                       // persistent launch: the next tile's prologue flies from here
            #pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                const int j = TOTAL - DEPTH + i;
                promote_only(acc[j / NS][j % NS], scale[MS - 1], part[(TOTAL + i + 1) & DEPTH]);
            }
        } else {
            fetch_next();
        }

        v4f out[MS][NS];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                out[ms][ns] = v4f{acc[ms][ns][0], acc[ms][ns][1], acc[ms][ns][2], acc[ms][ns][3]};
        if constexpr (DABL == 13) {
            if (out[0][0][0] == 123.456f) store_tile<MS, NS, true>(p, t, ad_group * p.d_sg, out, t.m0 + wm * WM, t.n0 + wn * WN);
        } else {
            store_tile<MS, NS, true, DABL == 12, B_MN>(p, t, ad_group * p.d_sg, out, t.m0 + wm * WM, t.n0 + wn * WN);
        }
        if (p.dbg != nullptr && first_tile && !next_prefetched) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_stamp(p, NW, 0, t_entry);
            dbg_stamp(p, NW, 1, t_loop0);
            dbg_stamp(p, NW, 2, t_loop1);
            dbg_stamp(p, NW, 3, __builtin_amdgcn_s_memtime());
        }
        t = tn;
        prefetched = next_prefetched;
        first_tile = false;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int DABL = 0>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_duo_abl_kernel(const GemmParams p) {
    duo_abl_kernel_body<BM, BN, WAVES_M, WAVES_N, DABL>(p);
}


// Hardware-scaled MFMA in the ring schedule (see the UE8M0 section of fp8_gemm_kernels.hpp for the operand semantics).
template <int BM, int BN, int WAVES_M, int WAVES_N>
__device__ __forceinline__ void e8_kernel_body(const GemmParams& p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16, TOTAL = MS * NS;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, A_SLOTS = 3, B_SLOTS = 2;
    constexpr int B_BASE = A_SLOTS * A_BYTES, LDS_BYTES = B_BASE + B_SLOTS * B_BYTES;
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN / 8 / NW;
    constexpr int P_STEP = TOTAL - NS, SCALE_LOADS = 6;
    constexpr int B_FIRST = (NS > 2 * A_ITERS ? NS : 2 * A_ITERS);
    static_assert(MS == 8 && NS == 4, "scale landing registers are written out for a 128 x 64 wave tile");
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave issues the same number of LDS-DMA pieces");
    static_assert(B_FIRST + 2 * (B_ITERS - 1) < P_STEP, "LDS-DMA pieces must be issued in front of barrier P");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int num_kb = p.k / 128;
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);
    auto a_unit_row = [](int u) { return (u / (WM / 8)) * WM + (u & 1) * 8 * MS + ((u % (WM / 8)) >> 1); };
    const int a_voff = piece_row * MS * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
    int a_piece_voff[A_ITERS], b_piece_voff[B_ITERS];
    #pragma unroll
    for (int q = 0; q < A_ITERS; ++q)
        a_piece_voff[q] = a_voff + a_unit_row(wave + NW * q) * lda;
    #pragma unroll
    for (int q = 0; q < B_ITERS; ++q)
        b_piece_voff[q] = b_voff + b_row_perm<WN>(q * (NW * 8)) * ldb;
    const int num_kq = (num_kb + 3) / 4;
    const int sfa_kq_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kq_stride = static_cast<int>(p.sfb_sk) * 4;   // bytes per packed K column

    const long long t_entry = p.dbg != nullptr ? __builtin_amdgcn_s_memtime() : 0;
    long long t_loop0 = 0, t_loop1 = 0;
    MaskedWalk walk;
    const int num_launched = gridDim.x;
    for (int tile_id = blockIdx.x;; tile_id += num_launched) {
        const Tile t = get_tile<BM, BN>(p, tile_id, walk);
        if (!t.valid)
            break;
        const int64_t ad_group = (p.gemm_type == kMasked) ? t.group : 0;

        v4f acc[MS][NS];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                acc[ms][ns] = v4f{0.f, 0.f, 0.f, 0.f};

        if (t.m_end > t.m0) {
            const uint8_t* a_base = p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm;
            const uint8_t* b_base = p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn;
            const int a_rows = imin(t.m_end - t.m0, BM), b_rows = imin(p.n - t.n0, BN);
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base), 0,
                                                                  (a_rows - 1) * lda + p.k, 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base), 0,
                                                                  (b_rows - 1) * ldb + p.k, 0x00020000);
            // packed scale words: element (row, kq) at base[kq * stride + row] (int32)
            const uint64_t sfa_addr = reinterpret_cast<uint64_t>(p.sfa + ad_group * p.sfa_sg);
            const uint64_t sfb_addr = reinterpret_cast<uint64_t>(p.sfb + static_cast<int64_t>(t.group) * p.sfb_sg);
            const v4i sfa_rsrc = {__builtin_amdgcn_readfirstlane(static_cast<int>(sfa_addr)),
                                  __builtin_amdgcn_readfirstlane(static_cast<int>(sfa_addr >> 32) & 0xffff),
                                  __builtin_amdgcn_readfirstlane((num_kq - 1) * sfa_kq_stride + p.m * 4), 0x00020000};
            const v4i sfb_rsrc = {__builtin_amdgcn_readfirstlane(static_cast<int>(sfb_addr)),
                                  __builtin_amdgcn_readfirstlane(static_cast<int>(sfb_addr >> 32) & 0xffff),
                                  __builtin_amdgcn_readfirstlane((num_kq - 1) * sfb_kq_stride + p.n * 4), 0x00020000};
            const int sfa_voff = (t.m0 + wm * WM + (lane & 15) * MS) * 4;
            int sfb_voff[NS];
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                const int i = lane & 15;
                sfb_voff[ns] = (t.n0 + wn * WN + (ns >> 1) * 32 + (i >> 2) * 8 + (ns & 1) * 4 + (i & 3)) * 4;
            }

            auto issue_a_piece = [&](int slot_off, int j, int q) {
                const int unit = wave + NW * q;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    a_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + unit * 1024), 16, a_piece_voff[q],
                    imin(j, num_kb - 1) * 128, 0, 0);
            };
            auto issue_b_piece = [&](int slot_off, int j, int q) {
                const int unit = wave + NW * q;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    b_rsrc, (__attribute__((address_space(3))) void*)(lds + B_BASE + slot_off + unit * 1024), 16,
                    b_piece_voff[q], imin(j, num_kb - 1) * 128, 0, 0);
            };
            E8Landing land;
            auto issue_scales = [&](int j) {           // the packed words that contain K block j
                const int kq = imin(j, num_kb - 1) >> 2;
                issue_e8_scale_loads(land, sfa_rsrc, sfa_voff + kq * sfa_kq_stride, sfb_rsrc, sfb_voff[0] + kq * sfb_kq_stride,
                                     sfb_voff[1] + kq * sfb_kq_stride, sfb_voff[2] + kq * sfb_kq_stride,
                                     sfb_voff[3] + kq * sfb_kq_stride);
            };
            int sa_cur[MS], sb_cur[NS];               // byte 0 = the exponent of the current K block
            auto take_scales = [&](int j) {
                const int shift = (imin(j, num_kb - 1) & 3) * 8;
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    sa_cur[ms] = static_cast<int>(static_cast<unsigned>(land.sa[ms / 4][ms % 4]) >> shift);
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    sb_cur[ns] = static_cast<int>(static_cast<unsigned>(land.sb[ns]) >> shift);
            };

            // ---- prologue: A(0) B(0) A(1) B(1) | scales of block 0, full drain ----
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece(0, 0, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS; ++q) issue_b_piece(0, 0, q);
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece(A_BYTES, 1, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS; ++q) issue_b_piece(B_BYTES, 1, q);
            issue_scales(0);
            wait_e8_landing<0>(land);
            take_scales(0);
            raw_barrier();

            int a_cur = 0, a_nxt = A_BYTES, b_nxt = B_BYTES;
            v8i bf[NS], af[2];
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                bf[ns] = load_fragment(lds + B_BASE + (wn * WN + ns * 16) * 128, frag_off);
            af[0] = load_fragment(lds + (wm * WM) * 128, frag_off);

            if (p.dbg != nullptr) t_loop0 = __builtin_amdgcn_s_memtime();
            for (int kb = 0; kb < num_kb; ++kb) {
                const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                const uint8_t* a_next_tile = lds + a_nxt + (wm * WM) * 128;
                const uint8_t* b_next_tile = lds + B_BASE + b_nxt + (wn * WN) * 128;
                const int a_fill = (a_nxt == (A_SLOTS - 1) * A_BYTES) ? 0 : a_nxt + A_BYTES;
                issue_scales(kb + 1);       // issue order per block: scales(kb+1) | A(kb+2) | B(kb+2) | P waits vmcnt(8)

                #pragma unroll
                for (int i = 0; i < TOTAL; ++i) {
                    const int ms = i / NS, ns = i % NS;
                    if (i == P_STEP) {
                        // barrier P: block kb+1 and its scale words landed everywhere; every read of A(kb) has returned
                        wait_e8_landing<A_ITERS + B_ITERS>(land);
                        raw_barrier();
                    }
                    if (ns == 0) {
                        if (ms + 1 < MS)
                            af[(ms + 1) & 1] = load_fragment(a_tile + (ms + 1) * 2048, frag_off);
                        else
                            af[(ms + 1) & 1] = load_fragment(a_next_tile, frag_off);
                    }
                    // operand roles are swapped (B rows in the A slot): the A-slot scale is the B row's, the B-slot scale the A row's
                    acc[ms][ns] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[ns], af[ms & 1], acc[ms][ns], 0, 0,
                                                                                   0, sb_cur[ns], 0, sa_cur[ms]);
                    if (ms == MS - 1)
                        bf[ns] = load_fragment(b_next_tile + ns * 2048, frag_off);
                    if (i % 2 == 0 && i / 2 < A_ITERS)
                        issue_a_piece(a_fill, kb + 2, i / 2);
                    if (i == NS - 1)
                        raw_barrier();                                          // barrier Q: B(kb) is in registers
                    if (i >= B_FIRST && (i - B_FIRST) % 2 == 0 && (i - B_FIRST) / 2 < B_ITERS)
                        issue_b_piece(b_nxt ^ B_BYTES, kb + 2, (i - B_FIRST) / 2);   // B(kb)'s slot
                }
                take_scales(kb + 1);
                a_cur = a_nxt;
                a_nxt = a_fill;
                b_nxt ^= B_BYTES;
            }
            if (p.dbg != nullptr) t_loop1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        store_tile<MS, NS, true>(p, t, ad_group * p.d_sg, acc, t.m0 + wm * WM, t.n0 + wn * WN);
        if (p.dbg != nullptr && tile_id == blockIdx.x) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_stamp(p, NW, 0, t_entry);
            dbg_stamp(p, NW, 1, t_loop0);
            dbg_stamp(p, NW, 2, t_loop1);
            dbg_stamp(p, NW, 3, __builtin_amdgcn_s_memtime());
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_e8_kernel(const GemmParams p) {
    e8_kernel_body<BM, BN, WAVES_M, WAVES_N>(p);
}


// ---------------------------------------------------------------------------------------------------------------
// FP32-scale quad kernel: the same one-wave-per-SIMD schedule for the reference's FP32 scaling factors (promotion
// acc += (sfa * sfb) * partial in the shadow of the following MFMAs, as in the 8-wave kernels).  The accumulators must be
// VALU-addressable, i.e. 128 arch VGPRs per wave => a 128 x 256 or 256 x 128 tile on four waves (wave tile 64 x 128 or
// 128 x 64).  One s_barrier per K block, no matrix-pipe hand-off between waves.  Used where the 256 x 256 tile does not fit
// the problem: grouped-contiguous layouts (BM must divide the 128-row alignment) and tile counts that quantise badly.
//
// Of the wave tile's two fragment sets the smaller one ("R": 4 fragments -- A for the 64-row wave tile, B for the 64-column
// one) stays resident in VGPRs for the whole K block, the other ("S": 8 fragments) streams through a 4-slot ring, two
// fragments ahead; ring slots 2 and 3 live in AGPRs (ds_read_b128 writes them there directly, the MFMA reads its operand
// from them), everything that is carried across the loop edge -- R and ring slots 0, 1 -- in VGPRs (a carried AGPR operand
// would be loaded into a VGPR and copied: 8 VALU moves per fragment).
// Schedule per K block (TOTAL = 32 steps; a step = one MFMA + the four FMAs of the step three back):
//   S-rows 0 .. 5 : S fragment s + 2 read at the head of row s; LDS-DMA: second half of B(kb+1), then A(kb+2)
//   barrier Z     : vmcnt(A_ITERS) => B(kb+1), A(kb+1) landed; everybody is done with A(kb)'s and B(kb)'s... slots
//   scale loads of block kb+1 (inline asm, VGPR landing), then S-rows 6, 7 R-major -- steps (r, 6), (r, 7) -- so that R
//   fragment r is dead after its pair and is re-read from block kb+1 at once; S fragments 0, 1 of block kb+1 follow;
//   LDS-DMA: first half of B(kb+2); the vmcnt(B_ITERS / 2) at the end of the block lands the scales (straight-line from
//   their issue: see "A latent race" in HISTORY.md).
// ---------------------------------------------------------------------------------------------------------------
template <bool ROWS_AGPR, bool COLS_AGPR, bool NO_FMA = false>
__device__ __forceinline__ void mfma_promote_step_q(v4f& part_new, const v8i& rows_operand, const v8i& cols_operand,
                                                    float (&c)[4], float scale, const v4f& part_old) {
#define DG_QSTEP_ASM(RC, CC)                                                                                             \
    asm volatile(                                                                                                        \
        "v_mfma_f32_16x16x128_f8f6f4 %0, %5, %6, 0\n\t"                                                                  \
        "v_fmac_f32 %1, %7, %8\n\t"                                                                                      \
        "v_fmac_f32 %2, %7, %9\n\t"                                                                                      \
        "v_fmac_f32 %3, %7, %10\n\t"                                                                                     \
        "v_fmac_f32 %4, %7, %11"                                                                                         \
        : "=&v"(part_new), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])                                                \
        : RC(rows_operand), CC(cols_operand), "v"(scale), "v"(part_old[0]), "v"(part_old[1]), "v"(part_old[2]),          \
          "v"(part_old[3])                                                                                               \
        : "memory")
    if constexpr (NO_FMA) {       // timing experiment: the bare MFMA stream (results are garbage)
        if constexpr (ROWS_AGPR) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0" : "=&v"(part_new) : "a"(rows_operand), "v"(cols_operand) : "memory");
        else if constexpr (COLS_AGPR) asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0" : "=&v"(part_new) : "v"(rows_operand), "a"(cols_operand) : "memory");
        else asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, 0" : "=&v"(part_new) : "v"(rows_operand), "v"(cols_operand) : "memory");
        asm volatile("" : "+v"(c[0]) : "v"(part_old[0]), "v"(scale));
    } else
    if constexpr (ROWS_AGPR && COLS_AGPR) DG_QSTEP_ASM("a", "a");
    else if constexpr (ROWS_AGPR) DG_QSTEP_ASM("a", "v");
    else if constexpr (COLS_AGPR) DG_QSTEP_ASM("v", "a");
    else DG_QSTEP_ASM("v", "v");
#undef DG_QSTEP_ASM
}

// QV (timing experiments, DG_EXPERIMENTS builds only; results are garbage): 1 no LDS-DMA in the loop; 2 no fragment reads;
// 3 no promotion FMAs; 4 no scale loads; 5 = 1 + 2; 6 = 1 + 2 + 3 (MFMA stream and the barrier only).
template <int BM, int BN, int WAVES_M, int WAVES_N, int QV = 0>
__device__ __forceinline__ void quad_kernel_body(const GemmParams& p) {
    constexpr int NW = 4;
    static_assert(WAVES_M * WAVES_N == NW, "one wave per SIMD");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16, TOTAL = MS * NS, DEPTH = 3;
    constexpr bool STREAM_A = MS > NS;                      // the streamed operand S; the other one (R) is resident
    constexpr int SS = STREAM_A ? MS : NS, RS = STREAM_A ? NS : MS;
    constexpr int PRE = (SS - 2) * RS, POST = 2 * RS;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, A_SLOTS = 3, B_SLOTS = 2;
    constexpr int B_BASE = A_SLOTS * A_BYTES, LDS_BYTES = B_BASE + B_SLOTS * B_BYTES;
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN / 8 / NW;
    constexpr int N_PRE = B_ITERS / 2 + A_ITERS, N_POST = B_ITERS / 2;
    constexpr bool NO_DMA = (QV == 1 || QV == 5 || QV == 6), NO_READS = (QV == 2 || QV == 5 || QV == 6), NO_FMA = (QV == 3 || QV == 6), NO_SCALES = (QV == 4);
    static_assert(SS == 8 && RS == 4, "wave tile 64 x 128 or 128 x 64: 128 accumulator registers");
    static_assert(WN <= 128 && 128 % WN == 0, "one SFB value per wave");
    static_assert(BM % (8 * NW) == 0 && BN % (16 * NW) == 0, "every wave issues the same number of pieces, B in two halves");
    static_assert(N_PRE <= PRE && N_POST <= POST, "at most one piece per step");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((NW * 8) % 16 == 0 && ((NW * 8) % WN == 0 || WN % (NW * 8) == 0),
                  "the row permutation of a B piece must be lane-independent");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int num_kb = p.k / 128;
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);
    auto a_unit_row = [](int u) { return (u / (WM / 8)) * WM + (u & 1) * 8 * MS + ((u % (WM / 8)) >> 1); };
    const int a_voff = piece_row * MS * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
    int a_piece_voff[A_ITERS], b_piece_voff[B_ITERS];
    #pragma unroll
    for (int q = 0; q < A_ITERS; ++q)
        a_piece_voff[q] = a_voff + a_unit_row(wave + NW * q) * lda;
    #pragma unroll
    for (int q = 0; q < B_ITERS; ++q)
        b_piece_voff[q] = b_voff + b_row_perm<WN>(q * (NW * 8)) * ldb;
    const int sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kb_stride = static_cast<int>(p.sfb_sk) * 4;
    const int sfa_extent = (p.m - 1) * 4 + (num_kb - 1) * sfa_kb_stride + 4;
    const int sfb_extent = (num_kb - 1) * sfb_kb_stride + 4;

    const long long t_entry = p.dbg != nullptr ? __builtin_amdgcn_s_memtime() : 0;
    long long t_loop0 = 0, t_loop1 = 0;
    MaskedWalk walk;
    const int num_launched = gridDim.x;
    auto uniform_ptr = [](const uint8_t* ptr) {
        const uint64_t v = reinterpret_cast<uint64_t>(ptr);
        const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v));
        const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
        return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
    };
    auto scale_rsrc = [](uint64_t addr, int extent) {
        return v4i{__builtin_amdgcn_readfirstlane(static_cast<int>(addr)),
                   __builtin_amdgcn_readfirstlane(static_cast<int>(addr >> 32) & 0xffff),
                   __builtin_amdgcn_readfirstlane(extent), 0x00020000};
    };

    int tile_id = blockIdx.x, pass = 0;
    while (true) {
        const Tile t = get_tile<BM, BN>(p, tile_id, walk, pass);
        if (!t.valid)
            break;
        const int64_t ad_group = (p.gemm_type == kMasked) ? t.group : 0;
        const int m_base = t.m0 + wm * WM, n_base = t.n0 + wn * WN;

        float acc[MS][NS][4];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                #pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[ms][ns][r] = 0.f;

        if (t.m_end > t.m0) {
            const uint8_t* a_base = uniform_ptr(p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm);
            const uint8_t* b_base = uniform_ptr(p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn);
            const int a_bytes = __builtin_amdgcn_readfirstlane((imin(t.m_end - t.m0, BM) - 1) * lda + p.k);
            const int b_bytes = __builtin_amdgcn_readfirstlane((imin(p.n - t.n0, BN) - 1) * ldb + p.k);
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base), 0, a_bytes, 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base), 0, b_bytes, 0x00020000);
            const v4i sfa_rsrc = scale_rsrc(reinterpret_cast<uint64_t>(p.sfa + ad_group * p.sfa_sg), sfa_extent);
            const v4i sfb_rsrc = scale_rsrc(reinterpret_cast<uint64_t>(p.sfb + static_cast<int64_t>(t.group) * p.sfb_sg +
                                                                       static_cast<int64_t>((t.n0 + wn * WN) / 128) * p.sfb_sn), sfb_extent);
            const int sfa_voff = (t.m0 + wm * WM + (lane & 15) * MS) * 4;

            auto issue_a_piece = [&](int slot_off, int j, int q) {
                if (NO_DMA) return;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    a_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + (wave + NW * q) * 1024), 16, a_piece_voff[q],
                    imin(j, num_kb - 1) * 128, 0, 0);
            };
            auto issue_b_piece = [&](int slot_off, int j, int q) {
                if (NO_DMA) return;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    b_rsrc, (__attribute__((address_space(3))) void*)(lds + B_BASE + slot_off + (wave + NW * q) * 1024), 16,
                    b_piece_voff[q], imin(j, num_kb - 1) * 128, 0, 0);
            };
            ScaleLandingV<MS> land;
            auto issue_scales = [&](int j) {
                const int jj = imin(j, num_kb - 1);             // past the end: the last block's scales again (never consumed)
                issue_scale_loads_v<MS>(land, sfa_rsrc, sfa_voff + jj * sfa_kb_stride, sfb_rsrc, jj * sfb_kb_stride);
            };

            // ---- prologue: A(0) B(0) scales(0) | A(1) B(1)[first half]; wait for the first group only ----
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece(0, 0, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS; ++q) issue_b_piece(0, 0, q);
            issue_scales(0);
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece(A_BYTES, 1, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS / 2; ++q) issue_b_piece(B_BYTES, 1, q);
            wait_landing_v<(NO_DMA ? 0 : A_ITERS + B_ITERS / 2), MS>(land);
            raw_barrier();

            int a_cur = 0, a_nxt = A_BYTES, a_fill = 2 * A_BYTES, b_cur = 0;
            const int a_wave = (wm * WM) * 128, b_wave = B_BASE + (wn * WN) * 128;          // this wave's rows inside a slot
            v8i rf[RS], sf[4];
            #pragma unroll
            for (int r = 0; r < RS; ++r)
                rf[r] = load_fragment(lds + (STREAM_A ? b_wave : a_wave) + r * 2048, frag_off);
            sf[0] = load_fragment(lds + (STREAM_A ? a_wave : b_wave), frag_off);
            sf[1] = load_fragment(lds + (STREAM_A ? a_wave : b_wave) + 2048, frag_off);

            float scale[MS], scale_prev[MS];
            v4f part[DEPTH + 1];
            #pragma unroll
            for (int i = 0; i <= DEPTH; ++i)
                part[i] = v4f{0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                scale[ms] = 0.f;

            // sequence index of a step within a K block -> (s, r): S-rows 0 .. SS-3 row-major, the last two S-rows R-major
            auto seq_s = [](int i) { return i < PRE ? i / RS : SS - 2 + ((i - PRE) & 1); };
            auto seq_r = [](int i) { return i < PRE ? i % RS : (i - PRE) >> 1; };
            // one step: MFMA of sequence index i, promotion of sequence index i - DEPTH (the first steps of a block: of the
            // previous block's last steps, at that block's scales)
            auto step = [&](int i, auto s_slot_in_agpr) {
                constexpr bool S_AGPR = decltype(s_slot_in_agpr)::value;
                const int s = seq_s(i), r = seq_r(i);
                const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;
                const int js = seq_s(j), jr = seq_r(j);
                const int jms = STREAM_A ? js : jr, jns = STREAM_A ? jr : js;
                const float jscale = (i >= DEPTH) ? scale[jms] : scale_prev[jms];
                if constexpr (STREAM_A)       // rows operand = B fragment (resident), columns operand = A fragment (streamed)
                    mfma_promote_step_q<false, S_AGPR, NO_FMA>(part[i & DEPTH], rf[r], sf[s & 3], acc[jms][jns], jscale, part[(i + 1) & DEPTH]);
                else
                    mfma_promote_step_q<S_AGPR, false, NO_FMA>(part[i & DEPTH], sf[s & 3], rf[r], acc[jms][jns], jscale, part[(i + 1) & DEPTH]);
            };

            if (p.dbg != nullptr) t_loop0 = __builtin_amdgcn_s_memtime();
            for (int kb = 0; kb < num_kb; ++kb) {
                const uint8_t* s_tile = lds + (STREAM_A ? a_cur + a_wave : b_cur + b_wave);
                const uint8_t* s_next_tile = lds + (STREAM_A ? a_nxt + a_wave : (b_cur ^ B_BYTES) + b_wave);
                const uint8_t* r_next_tile = lds + (STREAM_A ? (b_cur ^ B_BYTES) + b_wave : a_nxt + a_wave);
                // block kb's scales landed before the wait at the end of the previous block (block 0: the prologue's)
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    scale_prev[ms] = scale[ms];
                    scale[ms] = land.q[ms / 4][ms % 4] * land.sb;
                    pin_vgpr(scale[ms]);
                }
                // ---- S-rows 0 .. SS-3 ----
                #pragma unroll
                for (int i = 0; i < PRE; ++i) {
                    const int s = i / RS;
                    if (i % RS == 0 && !NO_READS)
                        sf[(s + 2) & 3] = load_fragment(s_tile + (s + 2) * 2048, frag_off);
                    step(i, std::false_type{});
                    // pieces: slot q of N_PRE rides behind step q * PRE / N_PRE: second half of B(kb+1), then A(kb+2)
                    #pragma unroll
                    for (int q = 0; q < N_PRE; ++q)
                        if (q * PRE / N_PRE == i) {
                            if (q < B_ITERS / 2)
                                issue_b_piece(b_cur ^ B_BYTES, kb + 1, B_ITERS / 2 + q);
                            else
                                issue_a_piece(a_fill, kb + 2, q - B_ITERS / 2);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- barrier Z ----
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_waitcnt(waitcnt_imm(NO_DMA ? 0 : A_ITERS, 0));
                raw_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (!NO_SCALES) issue_scales(kb + 1);               // older than every piece issued from here on
                // ---- S-rows SS-2, SS-1, R-major ----
                #pragma unroll
                for (int u = 0; u < POST; ++u) {
                    const int i = PRE + u, r = u >> 1;
                    step(i, std::false_type{});
                    if ((u & 1) && !NO_READS)
                        rf[r] = load_fragment(r_next_tile + r * 2048, frag_off);
                    if (u == 2 && !NO_READS) sf[0] = load_fragment(s_next_tile, frag_off);
                    if (u == POST / 2 + 2 && !NO_READS) sf[1] = load_fragment(s_next_tile + 2048, frag_off);
                    #pragma unroll
                    for (int q = 0; q < N_POST; ++q)
                        if (q * POST / N_POST + 1 == u)
                            issue_b_piece(b_cur, kb + 2, q);        // B(kb)'s slot: free since Z (its last reads were this block's)
                    __builtin_amdgcn_sched_barrier(0);
                }
                wait_landing_v<(NO_DMA ? 0 : N_POST), MS>(land);    // the scales of block kb+1 are in (only B(kb+2)'s first half may fly)
                const int a_free = a_cur;
                a_cur = a_nxt;
                a_nxt = a_fill;
                a_fill = a_free;
                b_cur ^= B_BYTES;
            }
            if (p.dbg != nullptr) t_loop1 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the tail's re-read pieces: the ring is about to be reused
            __syncthreads();
            #pragma unroll
            for (int i = 0; i < DEPTH; ++i) {                   // the last three steps' partials
                const int j = TOTAL - DEPTH + i;
                const int js = seq_s(j), jr = seq_r(j);
                const int jms = STREAM_A ? js : jr, jns = STREAM_A ? jr : js;
                promote_only(acc[jms][jns], scale[jms], part[(TOTAL + i + 1) & DEPTH]);
            }
        }

        if (p.d_dtype == 0 && !p.accumulate && p.d_vec_ok && n_base + WN <= p.n) {
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                #pragma unroll
                for (int g = 0; g < NS / 4; ++g) {
                    v4f quad4[4];
                    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        quad4[j] = v4f{acc[ms][4 * g + j][0], acc[ms][4 * g + j][1], acc[ms][4 * g + j][2], acc[ms][4 * g + j][3]};
                    store_rows_full_line<MS, true>(p, t, ad_group * p.d_sg, quad4, ms, m_base, n_base + 64 * g);
                }
        } else {
            auto store_group = [&](auto gc) {
                constexpr int G = decltype(gc)::value;
                v4f out[MS][4];
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        out[ms][j] = v4f{acc[ms][4 * G + j][0], acc[ms][4 * G + j][1], acc[ms][4 * G + j][2], acc[ms][4 * G + j][3]};
                store_tile<MS, 4, true>(p, t, ad_group * p.d_sg, out, m_base, n_base + 64 * G);
            };
            store_group(std::integral_constant<int, 0>{});
            if constexpr (NS == 8)
                store_group(std::integral_constant<int, 1>{});
        }
        if (p.dbg != nullptr && tile_id == static_cast<int>(blockIdx.x) && pass == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_stamp(p, NW, 0, t_entry);
            dbg_stamp(p, NW, 1, t_loop0);
            dbg_stamp(p, NW, 2, t_loop1);
            dbg_stamp(p, NW, 3, __builtin_amdgcn_s_memtime());
        }
        if (t.second_pass) {
            pass = 1;
        } else {
            pass = 0;
            tile_id += num_launched;
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int QV = 0>
__global__ __launch_bounds__(256)
void dg_fp8_gemm_quad_kernel(const GemmParams p) {
    quad_kernel_body<BM, BN, WAVES_M, WAVES_N, QV>(p);
}

}  // namespace dg
