// Quad kernel: the hardware-scaled (packed UE8M0) FP8 GEMM built around what hardware scaling frees on CDNA4.
//
// With v_mfma_scale_f32_16x16x128_f8f6f4 the matrix core applies 2^(ea-127) * 2^(eb-127) itself and accumulates IN PLACE
// across K blocks: there is no FP32 promotion pass and no need for VALU-addressable accumulators.  So this kernel runs
// FOUR waves per 256 x 256 tile -- one wave per SIMD with the whole 512-entry register file -- each owning a 128 x 128
// wave tile (256 accumulator registers, which hipcc may keep in AGPRs).  Against the 8-wave kernels (wave tile 128 x 64):
//   * fragment traffic out of LDS is 128 KiB per K block instead of 192 KiB (every A and B fragment feeds 8 MFMAs);
//   * one s_barrier per K block instead of four, and no hand-off of the matrix pipe between two waves of a SIMD: a wave's
//     64 MFMAs per K block issue back to back, everything else (32 ds_read_b128, 16 LDS-DMA pieces) rides in their shadow,
//     about 1 filler instruction per 32-cycle MFMA gap;
//   * the byte of a packed scale word is picked by the MFMA's op_sel, the K loop is unrolled over the four K blocks of a
//     word: no shifts, one set of scale loads per four K blocks.
//
// Reference semantics: deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_gemm_1d1d.cuh:34 (block-scaled UMMA with UE8M0
// scales accumulating in tensor memory), host driver csrc/jit_kernels/impls/sm100_fp8_fp4_gemm_1d1d.hpp:93,161,244.
//
// Schedule of one K block kb for a wave (MFMA step (ms, ns): acc[ms][ns] += B_frag[ns] x A_frag[ms]):
//   rows 0..5   : steps (ms, 0..7); at the head of row r the A fragment r + 2 is read into a 4-deep register ring;
//                 LDS-DMA: the second half of B(kb+1) and all of A(kb+2) (this wave's 12 pieces, one per fourth step; giving
//                 each wave its own step of the four -- four specialised copies of the loop -- made hipcc spill inside the
//                 loop, 231 us instead of 97, and the shared schedule shows no issue cost for the pieces: with or without
//                 them the loop runs 2.35-2.44 k cycles per K block)
//   barrier Z   : s_waitcnt vmcnt(8) (only A(kb+2) may still fly) + lgkmcnt(0), s_barrier  =>  A(kb+1) and B(kb+1) have
//                 landed for everybody, and everybody is done with A(kb)'s LDS slot and (since the previous block) B(kb)'s
//   rows 6..7   : ns-major -- steps (6, ns), (7, ns) -- so that B fragment ns is dead after its pair and is re-read from
//                 B(kb+1) at once; the first two A fragments of block kb+1 follow; LDS-DMA: the first half of B(kb+2)
//                 into B(kb)'s slot (A ring: 3 slots, B ring: 2 slots, 160 KiB)
// LDS image, row permutations and the epilogue are those of the other fast kernels (fp8_gemm_kernels.hpp).
#pragma once
#include "fp8_gemm_kernels.hpp"

namespace dg {

struct E8LandingQ { v4i sa[2]; int sb[8]; };
// The same words for wave tiles whose rows keep their NATURAL order (MN-major operands read in place, MNK below): the lane's A rows are 16 apart,
// so every word is a load of its own.
struct E8LandingQN { int sa[8]; int sb[8]; };
template <typename L> __device__ __forceinline__ int e8_sa(const L& l, int ms) {
    if constexpr (std::is_same_v<L, E8LandingQN>) return l.sa[ms]; else return l.sa[ms / 4][ms % 4];
}
template <typename L> __device__ __forceinline__ void e8_set_sa(L& l, int ms, int v) {
    if constexpr (std::is_same_v<L, E8LandingQN>) l.sa[ms] = v; else l.sa[ms / 4][ms % 4] = v;
}

// Packed scale words of one K quad: MS consecutive words of the lane's MS interleaved A rows (one or two dwordx4) and one
// word per N-subtile for its B rows.  K quad in the soffset, the N-subtile in the immediate offset.  NOT valid until the wait.
template <int MS, int NS = 8>
__device__ __forceinline__ void issue_e8q_scale_loads(E8LandingQ& l, const v4i& sfa_rsrc, int sfa_voff, int sfa_soff,
                                                      const v4i& sfb_rsrc, int sfb_voff, int sfb_soff) {
    static_assert((MS == 8 || MS == 4) && (NS == 8 || (NS == 4 && MS == 8)), "unrolled by hand");
    if constexpr (NS == 4)
        asm volatile(
            "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
            "buffer_load_dwordx4 %0, %6, %7, %8 offen\n\t"
            "buffer_load_dwordx4 %1, %6, %7, %8 offen offset:16\n\t"
            "buffer_load_dword %2, %9, %10, %11 offen\n\t"
            "buffer_load_dword %3, %9, %10, %11 offen offset:16\n\t"
            "buffer_load_dword %4, %9, %10, %11 offen offset:128\n\t"
            "buffer_load_dword %5, %9, %10, %11 offen offset:144"
            : "=&v"(l.sa[0]), "=&v"(l.sa[1]), "=&v"(l.sb[0]), "=&v"(l.sb[1]), "=&v"(l.sb[2]), "=&v"(l.sb[3])
            : "v"(sfa_voff), "s"(sfa_rsrc), "s"(sfa_soff), "v"(sfb_voff), "s"(sfb_rsrc), "s"(sfb_soff)
            : "memory");
    else if constexpr (MS == 8)
        asm volatile(
            "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
            "buffer_load_dwordx4 %0, %10, %11, %12 offen\n\t"
            "buffer_load_dwordx4 %1, %10, %11, %12 offen offset:16\n\t"
            "buffer_load_dword %2, %13, %14, %15 offen\n\t"
            "buffer_load_dword %3, %13, %14, %15 offen offset:16\n\t"
            "buffer_load_dword %4, %13, %14, %15 offen offset:128\n\t"
            "buffer_load_dword %5, %13, %14, %15 offen offset:144\n\t"
            "buffer_load_dword %6, %13, %14, %15 offen offset:256\n\t"
            "buffer_load_dword %7, %13, %14, %15 offen offset:272\n\t"
            "buffer_load_dword %8, %13, %14, %15 offen offset:384\n\t"
            "buffer_load_dword %9, %13, %14, %15 offen offset:400"
            : "=&v"(l.sa[0]), "=&v"(l.sa[1]), "=&v"(l.sb[0]), "=&v"(l.sb[1]), "=&v"(l.sb[2]), "=&v"(l.sb[3]), "=&v"(l.sb[4]),
              "=&v"(l.sb[5]), "=&v"(l.sb[6]), "=&v"(l.sb[7])
            : "v"(sfa_voff), "s"(sfa_rsrc), "s"(sfa_soff), "v"(sfb_voff), "s"(sfb_rsrc), "s"(sfb_soff)
            : "memory");
    else
        asm volatile(
            "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
            "buffer_load_dwordx4 %0, %9, %10, %11 offen\n\t"
            "buffer_load_dword %1, %12, %13, %14 offen\n\t"
            "buffer_load_dword %2, %12, %13, %14 offen offset:16\n\t"
            "buffer_load_dword %3, %12, %13, %14 offen offset:128\n\t"
            "buffer_load_dword %4, %12, %13, %14 offen offset:144\n\t"
            "buffer_load_dword %5, %12, %13, %14 offen offset:256\n\t"
            "buffer_load_dword %6, %12, %13, %14 offen offset:272\n\t"
            "buffer_load_dword %7, %12, %13, %14 offen offset:384\n\t"
            "buffer_load_dword %8, %12, %13, %14 offen offset:400"
            : "=&v"(l.sa[0]), "=&v"(l.sb[0]), "=&v"(l.sb[1]), "=&v"(l.sb[2]), "=&v"(l.sb[3]), "=&v"(l.sb[4]),
              "=&v"(l.sb[5]), "=&v"(l.sb[6]), "=&v"(l.sb[7])
            : "v"(sfa_voff), "s"(sfa_rsrc), "s"(sfa_soff), "v"(sfb_voff), "s"(sfb_rsrc), "s"(sfb_soff)
            : "memory");
}

template <int MS, int NS = 8>
__device__ __forceinline__ void tie_e8q_landing(E8LandingQ& l) {
    if constexpr (NS == 4)
        asm volatile("" : "+v"(l.sa[0]), "+v"(l.sa[1]), "+v"(l.sb[0]), "+v"(l.sb[1]), "+v"(l.sb[2]), "+v"(l.sb[3]) :: "memory");
    else if constexpr (MS == 8)
        asm volatile("" : "+v"(l.sa[0]), "+v"(l.sa[1]), "+v"(l.sb[0]), "+v"(l.sb[1]), "+v"(l.sb[2]), "+v"(l.sb[3]), "+v"(l.sb[4]),
                          "+v"(l.sb[5]), "+v"(l.sb[6]), "+v"(l.sb[7]) :: "memory");
    else
        asm volatile("" : "+v"(l.sa[0]), "+v"(l.sb[0]), "+v"(l.sb[1]), "+v"(l.sb[2]), "+v"(l.sb[3]), "+v"(l.sb[4]),
                          "+v"(l.sb[5]), "+v"(l.sb[6]), "+v"(l.sb[7]) :: "memory");
}

// ONE of the loads of issue_e8q_scale_loads (MS == 8, NS == 8: index 0, 1 = the two dwordx4 of the A rows, 2 .. 9 = the word of N-subtile
// index - 2), for the granularity-32 loop, which spreads a K block's ten scale loads over the MFMA gaps of the block in front of it.
// `index` is a constant after unrolling; the immediates must be literals.
__device__ __forceinline__ void issue_e8q_scale_load_one(E8LandingQ& l, int index, const v4i& sfa_rsrc, int sfa_voff, int sfa_soff,
                                                         const v4i& sfb_rsrc, int sfb_voff, int sfb_soff) {
#define DG_E8Q_SB(i, off) asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:" #off : "=&v"(l.sb[i]) : "v"(sfb_voff), "s"(sfb_rsrc), "s"(sfb_soff) : "memory")
    switch (index) {
    case 0: asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=&v"(l.sa[0]) : "v"(sfa_voff), "s"(sfa_rsrc), "s"(sfa_soff) : "memory"); break;
    case 1: asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=&v"(l.sa[1]) : "v"(sfa_voff), "s"(sfa_rsrc), "s"(sfa_soff) : "memory"); break;
    case 2: DG_E8Q_SB(0, 0); break;
    case 3: DG_E8Q_SB(1, 16); break;
    case 4: DG_E8Q_SB(2, 128); break;
    case 5: DG_E8Q_SB(3, 144); break;
    case 6: DG_E8Q_SB(4, 256); break;
    case 7: DG_E8Q_SB(5, 272); break;
    case 8: DG_E8Q_SB(6, 384); break;
    default: DG_E8Q_SB(7, 400); break;
    }
#undef DG_E8Q_SB
}

// Natural row order (E8LandingQN): word of A row m0 + 16 ms + i and of B row n0 + 16 ns + i for the lane's i = lane & 15 -- sixteen dword loads, the
// M- / N-subtile in the immediate offset.  `index` 0 .. 7: A subtile, 8 .. 15: B subtile; a constant after unrolling.
__device__ __forceinline__ void issue_e8n_scale_load_one(E8LandingQN& l, int index, const v4i& sfa_rsrc, int sfa_voff, int sfa_soff,
                                                         const v4i& sfb_rsrc, int sfb_voff, int sfb_soff) {
#define DG_E8N_SA(i, off) asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:" #off : "=&v"(l.sa[i]) : "v"(sfa_voff), "s"(sfa_rsrc), "s"(sfa_soff) : "memory")
#define DG_E8N_SB(i, off) asm volatile("buffer_load_dword %0, %1, %2, %3 offen offset:" #off : "=&v"(l.sb[i]) : "v"(sfb_voff), "s"(sfb_rsrc), "s"(sfb_soff) : "memory")
    switch (index) {
    case 0: DG_E8N_SA(0, 0); break;    case 1: DG_E8N_SA(1, 64); break;   case 2: DG_E8N_SA(2, 128); break;  case 3: DG_E8N_SA(3, 192); break;
    case 4: DG_E8N_SA(4, 256); break;  case 5: DG_E8N_SA(5, 320); break;  case 6: DG_E8N_SA(6, 384); break;  case 7: DG_E8N_SA(7, 448); break;
    case 8: DG_E8N_SB(0, 0); break;    case 9: DG_E8N_SB(1, 64); break;   case 10: DG_E8N_SB(2, 128); break; case 11: DG_E8N_SB(3, 192); break;
    case 12: DG_E8N_SB(4, 256); break; case 13: DG_E8N_SB(5, 320); break; case 14: DG_E8N_SB(6, 384); break; default: DG_E8N_SB(7, 448); break;
    }
#undef DG_E8N_SA
#undef DG_E8N_SB
}
__device__ __forceinline__ void tie_e8n_landing(E8LandingQN& l) {
    asm volatile("" : "+v"(l.sa[0]), "+v"(l.sa[1]), "+v"(l.sa[2]), "+v"(l.sa[3]), "+v"(l.sa[4]), "+v"(l.sa[5]), "+v"(l.sa[6]), "+v"(l.sa[7]),
                      "+v"(l.sb[0]), "+v"(l.sb[1]), "+v"(l.sb[2]), "+v"(l.sb[3]), "+v"(l.sb[4]), "+v"(l.sb[5]), "+v"(l.sb[6]), "+v"(l.sb[7]) :: "memory");
}

// The scaled MFMA accumulating in place, as inline asm: with the builtin hipcc treats every accumulator update as a new
// value, gives results and inputs different registers and rotates 256 registers back at the loop end through thousands of
// v_accvgpr moves and scratch spills.  "+a" pins each accumulator to one AGPR quad for the whole K loop.  J = byte of the
// packed scale words (op_sel / op_sel_hi of both scale operands, encoding checked against the builtin's output).
// Nothing pads hazards inside asm: consecutive steps never touch the same accumulator; the callers keep >= 12 wait states
// between the last MFMA and the first read of an accumulator, and a few between a VALU write of a scale register and here.
// (The A / B operands stay in VGPRs: fed from AGPRs the MFMA measured ~1.5x slower.)
template <int J>
__device__ __forceinline__ void mfma_e8_inplace(v4f& acc, const v8i& rows_operand, const v8i& cols_operand, int rows_scale,
                                                int cols_scale) {
#ifdef DG_QUAD_NOSCALE     // timing probe (results are garbage): the plain MFMA in the same stream -- what does the scale operand pair cost?
    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc) : "v"(rows_operand), "v"(cols_operand), "v"(rows_scale), "v"(cols_scale) : "memory");
    return;
#endif
    if constexpr (J == 0)
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0]"
                     : "+a"(acc) : "v"(rows_operand), "v"(cols_operand), "v"(rows_scale), "v"(cols_scale) : "memory");
    else if constexpr (J == 1)
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,1,0] op_sel_hi:[0,0,0]"
                     : "+a"(acc) : "v"(rows_operand), "v"(cols_operand), "v"(rows_scale), "v"(cols_scale) : "memory");
    else if constexpr (J == 2)
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[1,1,0]"
                     : "+a"(acc) : "v"(rows_operand), "v"(cols_operand), "v"(rows_scale), "v"(cols_scale) : "memory");
    else
        asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[1,1,0] op_sel_hi:[1,1,0]"
                     : "+a"(acc) : "v"(rows_operand), "v"(cols_operand), "v"(rows_scale), "v"(cols_scale) : "memory");
}

// QV (timing experiments, DG_EXPERIMENTS builds only; results are garbage):
//   0 production; 1 no LDS-DMA in the loop; 2 no fragment reads in the loop; 3 neither (matrix stream + barrier);
//   5 no barrier in the loop; 6 (STAGED) every staged load re-reads K block 0 (cache hits: what does memory latency cost?);
//   7 every LDS-DMA piece re-reads K block 0 (same instructions, cache hits only).
// BM = 256: wave tile 128 x 128 (MS = 8), the dense / large contiguous form; BM = 128: wave tile 64 x 128 (MS = 4) for the
// grouped layouts whose M alignment is 128 rows (contiguous, psum, masked) and for small dense problems.
// STAGED: the operand bytes go global -> VGPR -> ds_write_b128 instead of through LDS-DMA.  Why: a buffer_load ... lds of 1 KiB costs
// its wave ~83 ns of LDS-DMA throughput whatever the source (tools/ubench/fill_rate.hip: 4 waves per CU fill at 50 GB/s from L2 through
// LDS-DMA, 80 GB/s through registers; 8 waves 80 / 103), and this kernel's 16 pieces per wave and K block at 83 ns are 1.33 us --
// MORE than the 1.1 us its 64 MFMAs take: with the LDS-DMA pieces the loop is fill-bound (81.7 us for C2; 66.3 us with the pieces
// compiled out, profiles/r03_fill/NOTES.md; hipBLASLt's 4-wave kernel sits on the same 16 x 86 ns = 1.38 us per K block).
// Schedule: the same piece positions and LDS slots as the LDS-DMA form; position i of the piece stream WRITES the piece that form
// would have issued there (loaded half a K block earlier into one of POS / 2 staging registers of 16 bytes per lane) and
// then LOADS the piece of position i + POS / 2 into the freed register -- so every piece is in LDS no later than before, and its global
// load is in flight half a K block longer.
// WAVES_N = 4 ("octo"): the same single-barrier schedule on EIGHT waves (2 x 4, wave tile 128 x 64, 128 accumulators in AGPRs: two
// waves per SIMD fit the 512-entry register file).  Twice the waves issue the LDS-DMA pieces (8 per wave and K block: 0.82 us of fill
// instead of 1.31), at the price of 192 instead of 128 KiB of fragment reads and two waves sharing each matrix pipe.
// K_TAIL (128-row form, dense): K need not be a multiple of 128 (whole 16-byte chunks, K > 128 -- the dgrad shapes K = 2112, 576 with packed
// scales; the reference's SM100 kernels take any K through TMA zero-fill, csrc/jit_kernels/impls/sm100_fp8_fp4_gemm_1d1d.hpp:93).  The partial
// last block is an ordinary block of the loop whose pieces carry a per-lane offset bias that pushes the chunks at and beyond K out of the
// descriptor's range: an out-of-range LDS-DMA lane writes ZEROS into the LDS (the duo kernels' tail mechanism; tools/ubench/lds_dma_oob_probe.hip),
// so neither the row padding nor the next row's bytes reach the matrix core.  Its scale byte is the block's own, picked as for any block.
// TABSK (round 5): the remainder walk of the packed-scale contiguous tiling (table_mode 2) with every tile cut along K into table_pieces()
// ranges of whole K quads, one work item per (tile, piece) -- 64 .. 96 remainder tiles do not fill 256 CUs and each would stream its group's whole
// weight panel alone.  A piece accumulates its range in place as always and writes its FP32 partial tile, row-major [BM][BN], to the caller's
// workspace (slab (tile * pieces + piece)); dg_e8_tab_reduce_kernel adds the pieces in piece order and stores BF16.  Fewer than two pieces (no
// workspace, many tiles): whole tiles, stored directly.
// G32 (round 6): scale granularity 32 along K -- the reference's SM100 MX recipe for FP8 x FP8 (csrc/apis/gemm.hpp:311-312,
// csrc/apis/layout.hpp:48-58; per_token_cast_to_fp8(..., gran_k = 32), deep_gemm/utils/math.py:26-38) and the NATIVE block size of
// v_mfma_scale_f32_16x16x128_f8f6f4.  A packed word then holds the four exponents of ONE 128-K block (byte j = K bytes [32 j, 32 j + 32)), one word
// per row and K block: element (row, kb) at base[kb * stride + row].  What the hardware does with the scale operand, measured (tools/g32_probe.py,
// profiles/r06_probe/g32_scale_byte_mapping.log): lane (r, g) holds the K bytes 16 g .. 16 g + 15 (registers 0-3) and 64 + 16 g .. (registers 4-7)
// of row r -- the chunk pair (g, g + 4) the fragment reads of every kernel here already use is the matrix core's NATURAL K order, not a
// permutation -- and the scale supplied by lane group g applies to MX block g of the row, K bytes [32 g, 32 g + 32): the first (g < 2) or second
// register half of lane groups 2 (g & 1) and 2 (g & 1) + 1, NOT to the lane's own 32 bytes.  So the fragments stay as they are and lane group g
// takes byte g of the row's word.  One change against the gran-128 kernel: the words of block kb + 1 are loaded at the top of block kb (older than
// its pieces: landed by barrier Z's counted wait) and, behind the block's last MFMA, shifted down by 8 g per lane (16 VALU operations per K block)
// so that every lane's byte sits in byte 0: op_sel stays 0.
// KG (round 6, second half): the K-grouped GEMM with packed UE8M0 scales -- the reference's SM100 form of k_grouped_fp8_gemm_tn_contiguous
// (csrc/apis/gemm.hpp:299-346, impls/sm100_fp8_fp4_gemm_1d1d.cuh with GemmType::KGroupedContiguous; scheduler/gemm.cuh:74-85,238-261).  Recipe
// (1, 1, gran_k): one exponent per ROW of A and per ROW of B and gran_k K bytes -- which is the MX format the scaled MFMA takes as it is: no FP32
// promotion at all, the whole group's K range accumulates in the matrix core.  Group g: D[g] (FP32) += A[:, K_g] B[:, K_g]^T over its K range of the
// K-major operands ([m, sum_k] / [n, sum_k]: the host layer re-majors the reference's MN-major tensors once); tiles in group-major order, one per
// workgroup.  Scale words as the reference packs them (impls/smxx_layout.cuh:148-246): group g owns ceil(ceil(k_g / gran_k) / 4) packed rows of
// [packed_sf_k, mn] counted from the end of the group before it; byte j of the group's packed row r = scale block 4 r + j of the group.  Granularity 128:
// K block kb of the group reads byte kb & 3 of row kb >> 2; granularity 32: row kb, lane group g takes byte g (G32 above).  Either way the loop is the
// G32 loop: block kb + 1's words are loaded under block kb and shifted down so that the byte sits in byte 0 (a group's K extent is any multiple of
// 32: whole K quads cannot be assumed, and a fifth copy of the block body does not fit -- see the K tail note below).
// K ranges: kg_prefix (host-side extents) or, kg_psum, the device-side psum layout with K alignment m_alignment (group starts at the previous end
// rounded up to the alignment; the k-columns between a group's end and the next start hold zeros by the layout's contract).  A partial last block:
// the 16-byte chunks at and beyond the group's aligned end are pushed out of the descriptor's range (tail_bias, the K_TAIL mechanism) and land as zeros.
// MNK (with KG, 256 x 256): the operands are the reference's MN-major tensors as they are -- a [total_k, m], b [total_k, n], unit stride along m / n, row
// pitches a_sk / b_sk -- no re-majoring pass (0.19 of 1.23 ms at 8 x 4096 x 7168 x ~4096).  The LDS image of an operand's K block is [128 k][256 mn
// bytes] with the 16-byte chunks of row k stored at chunk ^ f(k) (load_fragment_tr's layout, fp8_gemm_kernels.hpp); an LDS-DMA piece is 4 k-rows x 256
// bytes; a fragment is four ds_read_b64_tr_b8 (the hardware transpose read: the same 32 K slots per lane as a K-major fragment) instead of two
// ds_read_b128; A and B rows keep their natural order, so a lane's scale words are sixteen separate loads (E8LandingQN) and the epilogue maps
// accumulator (ms, ns) to rows m0 + 16 ms + i, columns n0 + 16 ns + 4 g.  A group's k-rows beyond its aligned end lie outside the descriptor (zeros).
template <int BM, int BN, int QV = 0, bool STAGED = false, int WAVES_N = 2, bool K_TAIL = false, int HS = 0, bool TABSK = false, bool G32 = false, bool KG = false,
          bool MNK = false>
__device__ __forceinline__ void quad_e8_kernel_body(const GemmParams& p) {
    static_assert(!MNK || (KG && BM == 256 && BN == 256 && DG_M0_SHARE), "MNK: the K-grouped 256 x 256 form");
    static_assert(!G32 || (QV == 0 && !STAGED && WAVES_N == 2 && !K_TAIL && HS == 0 && !TABSK), "G32: the two production four-wave forms");
    static_assert(!KG || (QV == 0 && !STAGED && WAVES_N == 2 && !K_TAIL && HS == 0 && !TABSK), "KG: the two production four-wave forms");
    constexpr bool G32L = G32 || KG;        // the loop that shifts every block's scale bytes into byte 0 (no op_sel, no K quads)
    constexpr bool TAIL = K_TAIL || KG;     // a partial last K block, masked through the descriptor range
    static_assert(!TABSK || (BM == 128 && BN == 256 && QV == 0 && !STAGED && WAVES_N == 2 && !K_TAIL && HS == 0), "TABSK: the 128-row production form");
    static_assert(!K_TAIL || (BM == 128 && !STAGED && QV == 0 && WAVES_N == 2), "K tail: the 128-row production form");
    static_assert(HS == 0 || (BM == 256 && BN == 256 && QV == 0 && !STAGED && WAVES_N == 2 && !K_TAIL), "HS: the 256 x 256 four-wave form");
    constexpr int NW = 2 * WAVES_N;
    constexpr int WM = BM / 2, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16;
    constexpr int PRE = (MS - 2) * NS, POST = 2 * NS;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, A_SLOTS = HS ? 2 : 3, B_SLOTS = 2;
    constexpr int B_BASE = A_SLOTS * A_BYTES, LDS_BYTES = B_BASE + B_SLOTS * B_BYTES;
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN / 8 / NW;
    constexpr int N_PRE = B_ITERS / 2 + A_ITERS, N_POST = B_ITERS / 2;
    constexpr bool M0S = DG_M0_SHARE && !STAGED && A_ITERS % 4 == 0 && B_ITERS % 4 == 0 && (B_ITERS / 2) % 4 == 0;
    constexpr bool NO_DMA = (QV == 1 || QV == 3), NO_READS = (QV == 2 || QV == 3), NO_BARRIER = (QV == 5), HOT_LOADS = (QV == 6), HOT_DMA = (QV == 7);
    static_assert((NS == 8 && (MS == 8 || MS == 4)) || (NS == 4 && MS == 8), "wave tiles 128 x 128, 64 x 128 or (eight waves) 128 x 64");
    constexpr int PRE_STRIDE = PRE / N_PRE, POST_STRIDE = POST / N_POST;
    static_assert(B_ITERS % 2 == 0 && PRE % N_PRE == 0 && POST % N_POST == 0 && PRE_STRIDE >= 2 && POST_STRIDE >= 2,
                  "one piece per PRE_STRIDE / POST_STRIDE steps");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int num_kb = K_TAIL ? (p.k + 127) / 128 : p.k / 128, num_kq = (num_kb + 3) / 4;      // (TABSK: of the work item's K range; KG: of the tile's group)
    int k_tail = K_TAIL ? (p.k & 127) : 0;
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    [[maybe_unused]] const int g32_shift = (lane >> 4) * 8;         // G32: lane group g supplies the scale of MX block g = byte g of the row's word
    auto read_fragment = [&](const uint8_t* tile_rows) { return load_fragment(tile_rows, frag_off); };
    // MNK: the transpose read of 16-row subtile `sub` (0 .. 15 of the operand's 256 rows) out of the [128 k][256] image at `tile`
    [[maybe_unused]] const int tr_lane_base = (16 * (lane >> 4) + ((lane & 15) >> 1)) * 256 + (lane & 1) * 8;
    [[maybe_unused]] const int tr_swz = ((lane & 15) >> 1) | (((lane >> 4) & 1) << 3);
    // (inline asm with hand-placed lgkmcnt waits, as in pipe_pc_kernel_body: behind the BUILTIN form of the read hipcc puts an s_waitcnt vmcnt(0) --
    //  it cannot tell the read from the LDS-DMA pieces in flight -- and the K loop ran 4.4 us per block instead of 1.3)
    //  The four 8-byte results are put together into the MFMA operand AT ONCE: staged in their own registers until the wait (FragTr, as pipe_pc does
    //  with its four + two fragments) the sixteen fragments of this kernel do not fit -- 170 scratch operations in the K loop.  That is only correct
    //  if hipcc allocates the four results INTO the operand's registers (no copy in front of the wait); it does, and tests/test_codegen.py's
    //  landing check (no instruction touches a load's registers before its wait) holds the build to it.
    [[maybe_unused]] auto read_tr_now = [&](const uint8_t* tile, int chunk_off) {
        v2i_t q0, q1, q2, q3;
        const int addr = static_cast<int>(reinterpret_cast<uintptr_t>(tile)) + tr_lane_base + chunk_off;
        asm volatile("ds_read_b64_tr_b8 %0, %1" : "=&v"(q0) : "v"(addr) : "memory");
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:2048" : "=&v"(q1) : "v"(addr) : "memory");
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:16384" : "=&v"(q2) : "v"(addr) : "memory");
        asm volatile("ds_read_b64_tr_b8 %0, %1 offset:18432" : "=&v"(q3) : "v"(addr) : "memory");
        return v8i{q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
    };
    [[maybe_unused]] auto read_a_tr = [&](int slot, int ms) { return read_tr_now(lds + slot, ((wm * (BM / 32) + ms) ^ tr_swz) << 4); };
    [[maybe_unused]] auto read_b_tr = [&](int slot, int ns) {
        return read_tr_now(lds + (A_SLOTS * A_BYTES) + slot, ((wn * (BN / WAVES_N / 16) + ns) ^ tr_swz) << 4);
    };
    // K-major images: fragment ms of A / ns of B out of the ring slot at byte offset `slot` (B: relative to B_BASE)
    auto frag_a = [&](int slot, int ms) { return load_fragment(lds + slot + (wm * (BM / 2) + ms * 16) * 128, frag_off); };
    auto frag_b = [&](int slot, int ns) { return load_fragment(lds + (A_SLOTS * A_BYTES) + slot + (wn * (BN / WAVES_N) + ns * 16) * 128, frag_off); };
    const int lda = static_cast<int>(MNK ? p.a_sk : p.a_sm), ldb = static_cast<int>(MNK ? p.b_sk : p.b_sn);
    // A rows interleaved inside a wave's WM rows (LDS row position ms * 16 + i holds tile row i * MS + ms): a lane's MS row
    // scales are MS consecutive words of the MN-major scale tensor.  See duo_kernel_body.
    auto a_unit_row = [](int u) { return (u / (WM / 8)) * WM + (u & 1) * 8 * MS + ((u % (WM / 8)) >> 1); };
    const int a_voff = piece_row * MS * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
    int a_piece_voff[STAGED ? 1 : A_ITERS], b_piece_voff[STAGED ? 1 : B_ITERS];
    int a_piece_soff[STAGED ? A_ITERS : 1], b_piece_soff[STAGED ? B_ITERS : 1];     // STAGED: the (wave-uniform) row part in the soffset
    if constexpr (STAGED) {
        #pragma unroll
        for (int q = 0; q < A_ITERS; ++q)
            a_piece_soff[q] = __builtin_amdgcn_readfirstlane(a_unit_row(wave + NW * q) * lda);
        #pragma unroll
        for (int q = 0; q < B_ITERS; ++q)
            b_piece_soff[q] = __builtin_amdgcn_readfirstlane(b_row_perm<WN>(q * (NW * 8)) * ldb);
    } else if constexpr (MNK) {
        // unit u = 4 k-rows x 256 bytes at LDS offset 1024 u: lane l lands at row 4 u + (l >> 4), chunk position l & 15, and so carries source chunk
        // (l & 15) ^ f(k) of k-row k = 4 u + (l >> 4); u = wave * 8 + q (groups of four share one M0, as below)
        #pragma unroll
        for (int q = 0; q < A_ITERS; ++q) {
            const int k_in = 4 * (wave * A_ITERS + q) + (lane >> 4);
            const int f = (k_in & 7) | (((k_in >> 4) & 1) << 3);
            const int chunk = (((lane & 15) ^ f) << 4) + M0_SHARE_BIAS - (q & 3) * 1024;
            a_piece_voff[q] = k_in * lda + chunk;
            b_piece_voff[q] = k_in * ldb + chunk;
        }
    } else if constexpr (M0S) {
        // a wave owns A_ITERS (B_ITERS) CONSECUTIVE units: groups of four pieces share one M0 (DG_LDS_DMA_PIECE_SUB)
        #pragma unroll
        for (int q = 0; q < A_ITERS; ++q)
            a_piece_voff[q] = a_voff + a_unit_row(wave * A_ITERS + q) * lda + M0_SHARE_BIAS - (q & 3) * 1024;
        #pragma unroll
        for (int q = 0; q < B_ITERS; ++q)
            b_piece_voff[q] = b_row_perm<WN>((wave * B_ITERS + q) * 8 + piece_row) * ldb + src_chunk * 16 + M0_SHARE_BIAS - (q & 3) * 1024;
    } else {
        #pragma unroll
        for (int q = 0; q < A_ITERS; ++q)
            a_piece_voff[q] = a_voff + a_unit_row(wave + NW * q) * lda;
        #pragma unroll
        for (int q = 0; q < B_ITERS; ++q)
            b_piece_voff[q] = b_voff + b_row_perm<WN>(q * (NW * 8)) * ldb;
    }
    constexpr int POS = N_PRE + N_POST, DEPTH = POS / 2;            // piece positions per K block; staging registers (each serves two positions)
    static_assert(POS % 2 == 0, "a staging register serves two positions per K block");
    const int lane16 = lane * 16;
    [[maybe_unused]] int tail_bias = (K_TAIL && k_tail != 0 && src_chunk * 16 >= k_tail) ? 0x40000000 : 0;       // (KG: per tile)
    const int sfa_kq_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kq_stride = static_cast<int>(p.sfb_sk) * 4;   // bytes per K quad

    const long long t_entry = p.dbg != nullptr ? DG_STAMP_CLOCK() : 0;
    long long t_loop0 = 0, t_loop1 = 0;
    MaskedWalk walk;
    if (p.table_mode != 0)          // (round 5: the group-relative tiling of the contiguous layout, launch_e8_contiguous_tabled; every lane is active here)
        walk.table_mask = contiguous_tile_mask(p.layout, p.m, p.table_mode, &walk.block_group), walk.have_block_groups = true;
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

    [[maybe_unused]] int sk_tiles = 0, sk_pieces = 1, sk_piece = 0, sk_kq0 = 0;
    if constexpr (TABSK) {
        sk_tiles = table_count(p, walk) * p.num_n_tiles;
        sk_pieces = table_pieces(p, sk_tiles);
    }
    int tile_id = blockIdx.x, pass = 0;
    while (true) {
        int tile = tile_id;
        if constexpr (TABSK) {
            if (tile_id >= sk_tiles * sk_pieces)
                break;
            tile = tile_id % sk_tiles;
            sk_piece = tile_id / sk_tiles;
            const int total_kq = p.k / 512;                       // (host: k % 512 == 0)
            sk_kq0 = sk_piece * total_kq / sk_pieces;
            num_kq = (sk_piece + 1) * total_kq / sk_pieces - sk_kq0;
            num_kb = 4 * num_kq;
        }
        const Tile t = get_tile<BM, BN>(p, tile, walk, pass);
        if (!t.valid)
            break;
        const int64_t ad_group = (p.gemm_type == kMasked) ? t.group : 0;
        const int64_t d_group = KG ? t.group : ad_group;            // (KG: every group is a full M x N output)
        const int m_base = t.m0 + wm * WM, n_base = t.n0 + wn * WN;
        [[maybe_unused]] int kg_k0 = 0, kg_sf_row0 = 0, kg_k_mask = 0;
        if constexpr (KG) {
            // the group's K range [k0, k0 + extent), the extent up to which the operand bytes are the group's own or the layout's zeros (k_mask)
            // and the first of its packed scale rows: one scalar pass over the groups in front of it (<= 128 of them, once per tile)
            const int gran4 = G32 ? 128 : 512;                       // K bytes per packed scale row
            int prev_end = 0, rows = 0, k0 = 0, extent = 0, mask = 0;
            for (int g = 0; g <= t.group; ++g) {
                int start, end, aligned_end;
                if (p.kg_psum) {
                    const int a = p.m_alignment;                     // (host: a multiple of 32)
                    start = (prev_end + a - 1) / a * a;
                    end = imin(p.layout[g], p.k);
                    aligned_end = imin((end + a - 1) / a * a, p.k);
                    prev_end = end;
                } else {
                    start = p.kg_prefix[g];
                    end = aligned_end = p.kg_prefix[g + 1];
                }
                const int ext = imax(end - start, 0);
                if (g < t.group)
                    rows += (ext + gran4 - 1) / gran4;
                else
                    k0 = start, extent = ext, mask = imax(aligned_end - start, 0);
            }
            kg_k0 = __builtin_amdgcn_readfirstlane(k0);
            kg_sf_row0 = __builtin_amdgcn_readfirstlane(rows);
            kg_k_mask = __builtin_amdgcn_readfirstlane(mask);
            extent = __builtin_amdgcn_readfirstlane(extent);
            num_kb = (extent + 127) / 128;
            num_kq = (num_kb + 3) / 4;
            // chunks of the last block at and beyond the aligned end: out of range (zeros); none when the aligned end covers the block
            k_tail = kg_k_mask >= num_kb * 128 ? 0 : kg_k_mask - (num_kb - 1) * 128;
            tail_bias = (!MNK && k_tail != 0 && src_chunk * 16 >= k_tail) ? 0x40000000 : 0;     // (MNK: the k-rows beyond the end are out of range by themselves)
        }
        auto advance = [&] {
            if (t.second_pass) {
                pass = 1;
            } else {
                pass = 0;
                tile_id += num_launched;
            }
        };
        if (KG && num_kb <= 0) {            // an empty group: D[g] stays as it is (c was folded into d by the host layer)
            advance();
            continue;
        }
        if (t.m_end <= t.m0) {
            // nothing to compute (padding rows of a contiguous layout): zero rows only.  Kept apart from the main path so that
            // the accumulators there have ONE definition chain (a join with this path would be resolved by copying all of them
            // out of the AGPRs, spills included).
            if (!TABSK || sk_piece == 0) {              // (TABSK: the zero rows of a padding block are written once, by its first piece)
                v4f zero[MS][4];
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        zero[ms][j] = v4f{0.f, 0.f, 0.f, 0.f};
                store_tile<MS, 4, true>(p, t, ad_group * p.d_sg, zero, m_base, n_base);
                if constexpr (NS == 8)
                    store_tile<MS, 4, true>(p, t, ad_group * p.d_sg, zero, m_base, n_base + 64);
            }
            advance();
            continue;
        }

        v4f acc[MS][NS];        // (zeroed below, AFTER the prologue's loads have been issued: the 128 accumulator writes fly under the memory latency)

        {
            // (TABSK: the piece's K range starts at K quad sk_kq0 -- operand bases, scale bases and extents move, every index below is relative)
            // (KG: the group's column range of the K-major operands; b_sg / sfb_sg are 0 -- one B for the launch)
            const int k_ext = TABSK ? num_kb * 128 : KG ? imin(kg_k_mask, num_kb * 128) : p.k;
            const uint8_t* a_base = uniform_ptr(MNK ? p.a + static_cast<int64_t>(kg_k0) * lda + t.m0
                                                    : p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm + (TABSK ? sk_kq0 * 512 : 0) + (KG ? kg_k0 : 0));
            const uint8_t* b_base = uniform_ptr(MNK ? p.b + static_cast<int64_t>(kg_k0) * ldb + t.n0
                                                    : p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn + (TABSK ? sk_kq0 * 512 : 0) + (KG ? kg_k0 : 0));
            // (MNK: the descriptor ends with the last k-row's valid bytes; a lane past M (N) inside an earlier row reads the next k-row's head --
            //  bytes that only reach rows / columns which are never stored)
            const int a_bytes = __builtin_amdgcn_readfirstlane(MNK ? (k_ext - 1) * lda + (p.m - t.m0) : (imin(t.m_end - t.m0, BM) - 1) * lda + k_ext);
            const int b_bytes = __builtin_amdgcn_readfirstlane(MNK ? (k_ext - 1) * ldb + (p.n - t.n0) : (imin(p.n - t.n0, BN) - 1) * ldb + k_ext);
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base) - (M0S ? M0_SHARE_BIAS : 0), 0, a_bytes + (M0S ? M0_SHARE_BIAS : 0), 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base) - (M0S ? M0_SHARE_BIAS : 0), 0, b_bytes + (M0S ? M0_SHARE_BIAS : 0), 0x00020000);
            // packed scale words: element (row, kq) at base[kq * stride + row] (int32); rows of the whole A (masked: of the group)
            const int num_sf = G32 ? num_kb : num_kq;       // rows of the scale tensors along K: one per K quad (G32: per K block)
            const v4i sfa_rsrc = scale_rsrc(reinterpret_cast<uint64_t>(p.sfa + ad_group * p.sfa_sg + (TABSK ? sk_kq0 * p.sfa_sk : 0) + (KG ? kg_sf_row0 * p.sfa_sk : 0)),
                                            (num_sf - 1) * sfa_kq_stride + p.m * 4);
            const v4i sfb_rsrc = scale_rsrc(reinterpret_cast<uint64_t>(p.sfb + static_cast<int64_t>(t.group) * p.sfb_sg + (TABSK ? sk_kq0 * p.sfb_sk : 0) + (KG ? kg_sf_row0 * p.sfb_sk : 0)),
                                            (num_sf - 1) * sfb_kq_stride + p.n * 4);
            const int sfa_voff = MNK ? (t.m0 + wm * WM + (lane & 15)) * 4 : (t.m0 + wm * WM + (lane & 15) * MS) * 4;
            // B row of N-subtile ns, MFMA row slot i = lane & 15: wave_n0 + (ns >> 1) * 32 + (i >> 2) * 8 + (ns & 1) * 4 + (i & 3)
            const int sfb_voff = MNK ? (t.n0 + wn * WN + (lane & 15)) * 4 : (t.n0 + wn * WN + ((lane & 15) >> 2) * 8 + (lane & 3)) * 4;

            auto issue_a_piece = [&](int slot_off, int j, int q) {
                if (NO_DMA) return;
                if constexpr (STAGED)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        a_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + (wave + NW * q) * 1024), 16, a_voff,
                        a_piece_soff[q] + (HOT_DMA ? 0 : imin(j, num_kb - 1)) * 128, 0, 0);
                else if constexpr (M0S)
                    DG_LDS_DMA_PIECE_SUB(a_rsrc, lds + slot_off + (wave * A_ITERS + (q & ~3)) * 1024,
                                         a_piece_voff[q] + (TAIL && j >= num_kb - 1 ? tail_bias : 0),
                                         (HOT_DMA ? 0 : imin(j, num_kb - 1)) * (MNK ? 128 * lda : 128), q, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        a_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + (wave + NW * q) * 1024), 16,
                        a_piece_voff[q] + (TAIL && j >= num_kb - 1 ? tail_bias : 0), (HOT_DMA ? 0 : imin(j, num_kb - 1)) * 128, 0, 0);
            };
            auto issue_b_piece = [&](int slot_off, int j, int q) {
                if (NO_DMA) return;
                if constexpr (STAGED)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        b_rsrc, (__attribute__((address_space(3))) void*)(lds + B_BASE + slot_off + (wave + NW * q) * 1024), 16,
                        b_voff, b_piece_soff[q] + (HOT_DMA ? 0 : imin(j, num_kb - 1)) * 128, 0, 0);
                else if constexpr (M0S)
                    DG_LDS_DMA_PIECE_SUB(b_rsrc, lds + B_BASE + slot_off + (wave * B_ITERS + (q & ~3)) * 1024,
                                         b_piece_voff[q] + (TAIL && j >= num_kb - 1 ? tail_bias : 0),
                                         (HOT_DMA ? 0 : imin(j, num_kb - 1)) * (MNK ? 128 * ldb : 128), q, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        b_rsrc, (__attribute__((address_space(3))) void*)(lds + B_BASE + slot_off + (wave + NW * q) * 1024), 16,
                        b_piece_voff[q] + (TAIL && j >= num_kb - 1 ? tail_bias : 0), (HOT_DMA ? 0 : imin(j, num_kb - 1)) * 128, 0, 0);
            };
            // STAGED: position `pos` (0 .. POS - 1: N_PRE in rows 0 .. MS-3, N_POST in the last two rows; >= POS: the next block's) of
            // block kb is: second half of B(kb+1) | A(kb+2) | first half of B(kb+2)
            v4i stage[STAGED ? DEPTH : 1];
            auto stage_load = [&](int r, int pos, int kb) {
                if (pos >= POS) { pos -= POS; ++kb; }
                if (HOT_LOADS) kb = -2;
                if (pos < B_ITERS / 2)
                    stage[r] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(
                        b_rsrc, b_voff, b_piece_soff[B_ITERS / 2 + pos] + imax(imin(kb + 1, num_kb - 1), 0) * 128, 0));
                else if (pos < N_PRE)
                    stage[r] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(
                        a_rsrc, a_voff, a_piece_soff[pos - B_ITERS / 2] + imax(imin(kb + 2, num_kb - 1), 0) * 128, 0));
                else
                    stage[r] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(
                        b_rsrc, b_voff, b_piece_soff[pos - N_PRE] + imax(imin(kb + 2, num_kb - 1), 0) * 128, 0));
            };
            auto stage_write = [&](int r, int lds_off) {
                *reinterpret_cast<v4i*>(lds + lds_off + lane16) = stage[r];
            };
            using Landing = std::conditional_t<MNK, E8LandingQN, E8LandingQ>;
            Landing cur, nxt;
            auto issue_scales = [&](Landing& l, int kq) {
                const int q = imin(kq, num_sf - 1);
                if constexpr (MNK) {
                    asm volatile("s_nop 4" ::: "memory");       // SGPR operands written by VALU just before (see issue_e8q_scale_loads)
                    #pragma unroll
                    for (int i = 0; i < 16; ++i)
                        issue_e8n_scale_load_one(l, i, sfa_rsrc, sfa_voff, q * sfa_kq_stride, sfb_rsrc, sfb_voff, q * sfb_kq_stride);
                } else {
                    issue_e8q_scale_loads<MS, NS>(l, sfa_rsrc, sfa_voff, q * sfa_kq_stride, sfb_rsrc, sfb_voff, q * sfb_kq_stride);
                }
            };
            auto tie_landing = [&](Landing& l) {
                if constexpr (MNK) tie_e8n_landing(l); else tie_e8q_landing<MS, NS>(l);
            };
            // G32: every lane's own byte of the landed words into byte 0 (lane group g: bits [8 g, 8 g + 8)); KG at granularity 128: byte kb & 3
            // of the word, the same for every lane.  Row of the scale tensors and shift of K block kb:
            [[maybe_unused]] auto sf_row = [&](int kb) { return G32 ? kb : kb >> 2; };
            [[maybe_unused]] auto sf_shift = [&](int kb) { return G32 ? g32_shift : (kb & 3) * 8; };
            // One scale word of K block `kb` shifted down, as an instruction that STAYS where it is written: left as plain C++ the shifts of the
            // single-body loop are sunk to the top of the next block -- sixteen VALU slots in front of its first MFMA, 2.5 % of C2 (ISA checked).
            [[maybe_unused]] auto shifted_word = [&](int src, int kb) {
                int out;
                if constexpr (G32) {
                    asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(out) : "v"(g32_shift), "v"(src));
                } else {
                    const int sh = __builtin_amdgcn_readfirstlane((kb & 3) * 8);
                    asm volatile("v_lshrrev_b32 %0, %1, %2" : "=v"(out) : "s"(sh), "v"(src));
                }
                return out;
            };
            [[maybe_unused]] auto moved_word = [&](int src) {
                int out;
                asm volatile("v_mov_b32 %0, %1" : "=v"(out) : "v"(src));
                return out;
            };
            [[maybe_unused]] auto shift_down = [&](Landing& dst, const Landing& src, int sh) {
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    e8_set_sa(dst, ms, static_cast<int>(static_cast<unsigned>(e8_sa(src, ms)) >> sh));
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    dst.sb[ns] = static_cast<int>(static_cast<unsigned>(src.sb[ns]) >> sh);
            };

            if constexpr (HS != 0) {
                // ---- the register-resident schedule (round 5; the loop structure of the fastest third-party 256 x 256 FP8 kernel on this
                // part, profiles/r03_ceiling/NOTES.md (d)): ALL sixteen fragments of a K block live in registers (128 VGPRs), so an LDS
                // buffer is free as soon as its second-half fragments have been read -- two buffers per operand carry a prefetch distance of
                // ~1.5 K blocks (A 3-slot / B 2-slot ring of the default schedule: 0.6 - 1), and fragment reads and LDS-DMA pieces sit in
                // SEPARATE phases of the block (a piece issued beside fragment reads costs its wave 100-185 cycles, alone 60).
                //   phase 1  (a0..3 x b0..3)  reads b4..7 of B[u]                     | lgkmcnt(0), barrier 1: B[u] is free
                //   phase 2  (a0..3 x b4..7)  pieces B(kb+2) -> B[u]; reads a4..7     | lgkmcnt(0), barrier 2: A[u] is free
                //   phase 3  (a4..7 x b0..3)  pieces A(kb+2) -> A[u]                  | vmcnt(16): my pieces of block kb+1 are in; barrier 3
                //   phase 4  (a4..7 x b4..7)  reads b0..3, a0..3 of block kb+1 from the other buffers
                // HS == 1: one piece per MFMA gap (the first eight of a phase); HS == 2: one per two gaps. ----
                static_assert(A_ITERS == 8 && B_ITERS == 8 && MS == 8 && NS == 8, "two M0 groups of four pieces per operand and wave");
                #pragma unroll
                for (int q = 0; q < A_ITERS; ++q) issue_a_piece(0, 0, q);
                #pragma unroll
                for (int q = 0; q < B_ITERS; ++q) issue_b_piece(0, 0, q);
                issue_scales(cur, 0);
                #pragma unroll
                for (int q = 0; q < A_ITERS; ++q) issue_a_piece(A_BYTES, 1, q);
                #pragma unroll
                for (int q = 0; q < B_ITERS; ++q) issue_b_piece(B_BYTES, 1, q);
                asm volatile("" ::: "memory");
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        acc[ms][ns] = v4f{0.f, 0.f, 0.f, 0.f};
                        asm volatile("" : "+a"(acc[ms][ns]));
                    }
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_waitcnt(waitcnt_imm(A_ITERS + B_ITERS, 15));
                tie_e8q_landing<MS, NS>(cur);
                raw_barrier();
                int a_cur = 0, b_cur = 0;
                v8i bf[NS], af[MS];
                #pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bf[i] = load_fragment(lds + B_BASE + (wn * WN + i * 16) * 128, frag_off);
                    af[i] = load_fragment(lds + (wm * WM + i * 16) * 128, frag_off);
                }
                if (p.dbg != nullptr) t_loop0 = DG_STAMP_CLOCK();
                asm volatile("s_nop 7" ::: "memory");               // zero-initialised accumulators (VALU writes) -> first MFMA
                auto block = [&](auto jc, auto load_next, const E8LandingQ& w, int kb) {
                    constexpr int J = decltype(jc)::value;
                    constexpr bool LOAD_NEXT = decltype(load_next)::value;
                    constexpr int GAP = HS;                          // MFMA gaps per piece
                    const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                    const uint8_t* b_tile = lds + B_BASE + b_cur + (wn * WN) * 128;
                    const uint8_t* a_next_tile = lds + (a_cur ^ A_BYTES) + (wm * WM) * 128;
                    const uint8_t* b_next_tile = lds + B_BASE + (b_cur ^ B_BYTES) + (wn * WN) * 128;
                    if (LOAD_NEXT) issue_scales(nxt, (kb >> 2) + 1);    // older than every piece of this block: in by its vmcnt(16)
                    // ---- phase 1 ----
                    #pragma unroll
                    for (int step = 0; step < 16; ++step) {
                        const int ms = step >> 2, ns = step & 3;
                        mfma_e8_inplace<J>(acc[ms][ns], bf[ns], af[ms], w.sb[ns], w.sa[ms / 4][ms % 4]);
                        if (step < 4) bf[4 + step] = load_fragment(b_tile + (4 + step) * 2048, frag_off);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
                    raw_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- phase 2 ----
                    #pragma unroll
                    for (int step = 0; step < 16; ++step) {
                        const int ms = step >> 2, ns = 4 + (step & 3);
                        mfma_e8_inplace<J>(acc[ms][ns], bf[ns], af[ms], w.sb[ns], w.sa[ms / 4][ms % 4]);
                        if constexpr (GAP == 1) {
                            if (step < 8) issue_b_piece(b_cur, kb + 2, step);
                            else if (step < 12) af[step - 4] = load_fragment(a_tile + (step - 4) * 2048, frag_off);
                        } else {
                            if (step % 2 == 0) issue_b_piece(b_cur, kb + 2, step / 2);
                            else if (step >= 7 && step < 15) {          // steps 7, 9, 11, 13: exactly the four fragments of rows 4 .. 7
                                static_assert(MS == 8, "af[4 .. 7] are the fragments read here");
                                af[4 + (step - 7) / 2] = load_fragment(a_tile + (4 + (step - 7) / 2) * 2048, frag_off);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
                    raw_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- phase 3 ----
                    #pragma unroll
                    for (int step = 0; step < 16; ++step) {
                        const int ms = 4 + (step >> 2), ns = step & 3;
                        mfma_e8_inplace<J>(acc[ms][ns], bf[ns], af[ms], w.sb[ns], w.sa[ms / 4][ms % 4]);
                        if constexpr (GAP == 1) {
                            if (step < 8) issue_a_piece(a_cur, kb + 2, step);
                        } else {
                            if (step % 2 == 0) issue_a_piece(a_cur, kb + 2, step / 2);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(A_ITERS + B_ITERS, 15));     // everything but this block's sixteen pieces
                    if (LOAD_NEXT) tie_e8q_landing<MS, NS>(nxt);
                    raw_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- phase 4 ----
                    #pragma unroll
                    for (int step = 0; step < 16; ++step) {
                        const int ms = 4 + (step >> 2), ns = 4 + (step & 3);
                        mfma_e8_inplace<J>(acc[ms][ns], bf[ns], af[ms], w.sb[ns], w.sa[ms / 4][ms % 4]);
                        if (step < 4) bf[step] = load_fragment(b_next_tile + step * 2048, frag_off);
                        else if (step < 8) af[step - 4] = load_fragment(a_next_tile + (step - 4) * 2048, frag_off);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    a_cur ^= A_BYTES;
                    b_cur ^= B_BYTES;
                };
                using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
                using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
                using Yes = std::true_type; using No = std::false_type;
                for (int kb = 0; kb + 4 <= num_kb; kb += 4) {
                    block(I0{}, No{}, cur, kb);
                    block(I1{}, Yes{}, cur, kb + 1);
                    block(I2{}, No{}, cur, kb + 2);
                    block(I3{}, No{}, cur, kb + 3);
                    cur = nxt;
                    asm volatile("s_nop 3" ::: "memory");           // VALU-written scale registers -> MFMA
                }
                if (p.dbg != nullptr) t_loop1 = DG_STAMP_CLOCK();
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            } else {
            // ---- prologue: A(0) B(0) words(0) | A(1) B(1)[first half]; wait for the first group only ----
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece(0, 0, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS; ++q) issue_b_piece(0, 0, q);
            // (the shifted-scale loop of the 256-row forms takes a block's words from `nxt`, see G32_DEFER in block())
            constexpr bool SPREAD_FORM = G32L && MS == 8 && NS == 8 && PRE_STRIDE == 4 && POST == 16;
#ifndef DG_RMW_NT
#define DG_RMW_NT 0                 // (1 / 2 / 3: non-temporal loads / stores / both in the FP32 reduce-add epilogue -- measured 1-8 % slower on the K-grouped
                                    //  call, profiles/r06_probe/rmw_epilogue_nontemporal_negative.log)
#endif
#ifndef DG_QUAD_SPREAD
#define DG_QUAD_SPREAD 0            // (1: tuning builds.  Measured neutral, same box, product against variant: dense_ue8m0 80.0-80.8 us either way,
                                    //  contiguous_ue8m0 136.2-137.7 either way -- profiles/r06_probe/quad_spread_neutral.log: the ten moves and the
                                    //  s_nop between two K quads are not where the op_sel loop loses time)
#endif
            constexpr bool QUAD_SPREAD_FORM = DG_QUAD_SPREAD && !G32L && QV == 0 && !STAGED && MS == 8 && NS == 8 && PRE_STRIDE == 4 && POST == 16;
            issue_scales(SPREAD_FORM ? nxt : cur, 0);
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece(A_BYTES, 1, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS / 2; ++q) issue_b_piece(B_BYTES, 1, q);
            asm volatile("" ::: "memory");
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    acc[ms][ns] = v4f{0.f, 0.f, 0.f, 0.f};
                    asm volatile("" : "+a"(acc[ms][ns]));          // materialised HERE (hipcc sinks plain zero-initialisation to the first MFMA: behind the wait)
                }
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(waitcnt_imm(NO_DMA ? 0 : A_ITERS + B_ITERS / 2, 0));
            if constexpr (SPREAD_FORM) {
                tie_landing(nxt);
                shift_down(cur, nxt, sf_shift(0));      // (`nxt` keeps the raw words: block 0 redoes its four deferred ones from there)
            } else {
                tie_landing(cur);
                if constexpr (G32)
                    shift_down(cur, cur, g32_shift);    // (KG at granularity 128: block 0 is byte 0)
                if constexpr (QUAD_SPREAD_FORM)
                    nxt = cur;                          // (QUAD_SPREAD: block 0 of the first quad moves four words out of `nxt`)
            }
            raw_barrier();
            if constexpr (STAGED) {
                #pragma unroll
                for (int r = 0; r < DEPTH; ++r)
                    stage_load(r, r, 0);                        // the first DEPTH positions of block 0
            }

            // slots (byte offsets): A(kb), A(kb+1), A(kb+2) [= where A(kb+2) is filled]; B(kb), B(kb+1)
            int a_cur = 0, a_nxt = A_BYTES, a_fill = 2 * A_BYTES, b_cur = 0;
            v8i bf[NS], af[4];
            // MNK: the transpose reads are asm (the compiler knows neither their latency nor their counter): every first use of a fragment sits
            // behind a counted lgkmcnt wait.  The reads of a wave return in order; per block they are issued as
            //   rows 0 .. 5: A(ms + 2) at the head of row ms | barrier Z (lgkmcnt 0) | last two rows: B'(ns) behind step 2 ns + 1, A'(0) at step 4, A'(1) at step 10
            // (' = of the next block), four reads each.
            if constexpr (MNK) {
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    bf[ns] = read_b_tr(0, ns);
                af[0] = read_a_tr(0, 0);
                af[1] = read_a_tr(0, 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                bf[ns] = frag_b(0, ns);
            af[0] = frag_a(0, 0);
            af[1] = frag_a(0, 1);
            }

            if (p.dbg != nullptr) t_loop0 = DG_STAMP_CLOCK();
            asm volatile("s_nop 7" ::: "memory");               // zero-initialised accumulators (VALU writes) -> first MFMA
            // One K block with byte J of the scale words `w`.  LOAD_NEXT: this block issues the loads of the next K quad's
            // words (block J == 1 of a whole quad); TIE_NEXT: they are waited for at this block's barrier (J == 2).
            // G32, 256-row form: `land` receives block kb + 1's words -- one load per MFMA gap in steps 2 .. 15 (all of them older than the A pieces,
            // the only operations barrier Z's counted wait leaves in flight), tied at barrier Z, shifted IN PLACE in the gaps of the last two rows
            // (one word per step) into `w`'s own registers as they fall free: one block body, one register set in use.
            auto block = [&](auto jc, auto load_next, auto tie_next, Landing& w, int kb, Landing& land) {
                constexpr int J = decltype(jc)::value;
                constexpr bool G32_SPREAD = SPREAD_FORM;
                // QUAD_SPREAD (round 6, the op_sel loop of the 256-row form): the next K quad's words move from `land` into w's registers inside
                // block J == 3 as those fall free (and the last four in the first gaps of the next quad's block J == 0) instead of ten moves and an
                // s_nop between two quads, right behind the MFMAs that read the destination registers -- the schedule of the shifted-scale loop
                constexpr bool QUAD_SPREAD = QUAD_SPREAD_FORM;
                constexpr bool LOAD_NEXT = decltype(load_next)::value, TIE_NEXT = decltype(tie_next)::value;
                const int b_next_slot = b_cur ^ B_BYTES;
                [[maybe_unused]] const int g32_q = imin(sf_row(kb + 1), num_sf - 1);
                [[maybe_unused]] const int g32_sh = sf_shift(kb + 1);
                if constexpr (G32L && !G32_SPREAD)
                    issue_scales(land, sf_row(kb + 1));     // block kb + 1's words: older than every piece of this block, in by barrier Z's counted wait
                // ---- rows 0 .. MS-3 ----
                #pragma unroll
                for (int step = 0; step < PRE; ++step) {
                    // boustrophedon order (round 5): at a row change the B fragment stays, so one operand of EVERY MFMA equals its predecessor's --
                    // on this power-limited part worth 0.4-0.5 us of 79 (same box, 3 of 4 pairs: profiles/r05_probe/quad_serpentine_order_ab.log);
                    // the accumulators are independent: same bits
                    const int ms = step / NS, ns = (ms & 1) ? NS - 1 - step % NS : step % NS;
                    if constexpr (MNK) {
                        if (step % NS == 0 && ms + 2 < MS)
                            af[(ms + 2) & 3] = read_a_tr(a_cur, ms + 2);
                        if (step == 0)          // behind A(2): the last 15 reads may still fly -- A(2), B'(7), B'(6), three of B'(5); B'(0 .. 4), A'(0), A'(1) are in
                            asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
                        else if (step == 5)
                            asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");        // B'(6), B'(7), A(2) behind B'(5)
                        else if (step == 6)
                            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                        else if (step == 7)
                            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                        else if (step % NS == 0 && ms >= 2)
                            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");         // A(ms + 1), A(ms + 2) behind A(ms)
                    } else {
                    if (step % NS == 0 && !NO_READS)
                        af[(ms + 2) & 3] = frag_a(a_cur, ms + 2);
                    }
                    mfma_e8_inplace<J>(acc[ms][ns], bf[ns], af[ms & 3], w.sb[ns], e8_sa(w, ms));
                    // pieces: one per PRE_STRIDE steps: second half of B(kb+1), then A(kb+2)
                    if (step % PRE_STRIDE == 1) {
                        const int q = step / PRE_STRIDE;
                        if constexpr (STAGED) {
                            if (q < B_ITERS / 2)
                                stage_write(q % DEPTH, B_BASE + (b_cur ^ B_BYTES) + (wave + NW * (B_ITERS / 2 + q)) * 1024);
                            else
                                stage_write(q % DEPTH, a_fill + (wave + NW * (q - B_ITERS / 2)) * 1024);
                            if constexpr (PRE_STRIDE < 4)
                                stage_load(q % DEPTH, q + DEPTH, kb);
                        } else if (q < B_ITERS / 2) {
                            issue_b_piece(b_cur ^ B_BYTES, kb + 1, B_ITERS / 2 + q);
                        } else {
                            issue_a_piece(a_fill, kb + 2, q - B_ITERS / 2);
                        }
                    }
                    if constexpr (STAGED && PRE_STRIDE >= 4)        // the refill of the staging register two MFMA gaps behind its write:
                        if (step % PRE_STRIDE == 3)                 // one filler per gap (both in one gap held up the next MFMA)
                            stage_load((step / PRE_STRIDE) % DEPTH, step / PRE_STRIDE + DEPTH, kb);
                    if constexpr (G32_SPREAD) {
                        // G32_DEFER: the four words the previous block's LAST steps still read (the A words of rows 6, 7 and the B words of N-subtiles
                        // 6, 7) are shifted into w's registers HERE, out of `land`, which still holds this block's raw words -- its reload starts in gap 2
                        // with sa[0]; sa[1] follows in gap 3 (behind the shift there), sb[6] / sb[7] in gaps 14 / 15.  First readers: steps 6, 7 and the last two rows.  Nothing waits
                        // between two blocks (shifted behind the previous block's last MFMA they cost an s_nop + three VALU slots of every block: 2.5 %)
                        // (every write of a w register keeps >= 3 MFMAs behind the last MFMA that read it as its scale operand)
                    }
                    if constexpr (QUAD_SPREAD && J == 0) {          // (the first quad of a tile: `land` was set to the same words by the prologue)
                        if (step == 2) e8_set_sa(w, 6, moved_word(e8_sa(land, 6)));
                        if (step == 3) e8_set_sa(w, 7, moved_word(e8_sa(land, 7)));
                        if (step == 4) w.sb[6] = moved_word(land.sb[6]);
                        if (step == 5) w.sb[7] = moved_word(land.sb[7]);
                    }
                    if constexpr (G32_SPREAD) {
                        if (step == 2) e8_set_sa(w, 6, shifted_word(e8_sa(land, 6), kb));
                        if (step == 3) e8_set_sa(w, 7, shifted_word(e8_sa(land, 7), kb));
                        if (step == 4) w.sb[6] = shifted_word(land.sb[6], kb);
                        if (step == 5) w.sb[7] = shifted_word(land.sb[7], kb);
                        // gaps 2, 3, 4, 6, 7, 10, 11, 12, 14, 15 (pieces sit in gaps 1, 5, 9, 13, ...; fragment reads in gaps 0, 8, ...)
                        if constexpr (MNK) {
                            // sixteen loads, two per gap, all of them in front of the first A piece (gap 17): older than what barrier Z leaves in flight
                            constexpr int kSlotsN[8] = {6, 7, 10, 11, 12, 14, 15, 16};
                            #pragma unroll
                            for (int i = 0; i < 8; ++i)
                                if (step == kSlotsN[i]) {
                                    issue_e8n_scale_load_one(land, 2 * i, sfa_rsrc, sfa_voff, g32_q * sfa_kq_stride, sfb_rsrc, sfb_voff, g32_q * sfb_kq_stride);
                                    issue_e8n_scale_load_one(land, 2 * i + 1, sfa_rsrc, sfa_voff, g32_q * sfa_kq_stride, sfb_rsrc, sfb_voff, g32_q * sfb_kq_stride);
                                }
                        } else {
                        constexpr int kSlots[10] = {2, 3, 4, 6, 7, 10, 11, 12, 14, 15};
                        #pragma unroll
                        for (int i = 0; i < 10; ++i)
                            if (step == kSlots[i])
                                issue_e8q_scale_load_one(land, i, sfa_rsrc, sfa_voff, g32_q * sfa_kq_stride, sfb_rsrc, sfb_voff, g32_q * sfb_kq_stride);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- barrier Z ----
                asm volatile("" ::: "memory");
                if constexpr (STAGED) {
                    // every piece of block kb+1 has been WRITTEN (lgkmcnt 0); the next quad's words, when tied here, were issued one
                    // block earlier with N_POST + N_PRE staged loads behind them
                    if (TIE_NEXT) __builtin_amdgcn_s_waitcnt(waitcnt_imm(N_POST + N_PRE, 0));
                    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else {
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(NO_DMA ? 0 : A_ITERS, 0));
                }
                if (TIE_NEXT) tie_landing(nxt);                     // the next K quad's words (issued one block earlier) are in
                if (G32L) tie_landing(land);                        // G32 / KG: the next block's words, issued in this block
                if (!NO_BARRIER) raw_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (LOAD_NEXT) issue_scales(nxt, (kb >> 2) + 1);   // older than every piece issued from here on
                // ---- rows MS-2, MS-1, ns-major ----
                #pragma unroll
                for (int step = 0; step < POST; ++step) {
                    const int ns = step >> 1, ms = MS - 2 + ((step & 1) ^ (ns & 1));
                    mfma_e8_inplace<J>(acc[ms][ns], bf[ns], af[ms & 3], w.sb[ns], e8_sa(w, ms));
                    if constexpr (MNK) {
                        if (step & 1) bf[ns] = read_b_tr(b_next_slot, ns);
                        if (step == POST / 4) af[0] = read_a_tr(a_nxt, 0);
                        if (step == (POST * 5) / 8) af[1] = read_a_tr(a_nxt, 1);
                    } else {
                    if ((step & 1) && !NO_READS)
                        bf[ns] = frag_b(b_next_slot, ns);
                    if (step == POST / 4 && !NO_READS) af[0] = frag_a(a_nxt, 0);
                    if (step == (POST * 5) / 8 && !NO_READS) af[1] = frag_a(a_nxt, 1);
                    }
                    if (step % POST_STRIDE == 1) {
                        if constexpr (STAGED) {
                            const int pos = N_PRE + step / POST_STRIDE;
                            stage_write(pos % DEPTH, B_BASE + b_cur + (wave + NW * (step / POST_STRIDE)) * 1024);
                            if constexpr (POST_STRIDE < 4)
                                stage_load(pos % DEPTH, pos + DEPTH, kb);
                        } else {
                            issue_b_piece(b_cur, kb + 2, step / POST_STRIDE);   // B(kb)'s slot: its fragments have been in registers since the last block
                        }
                    }
                    if constexpr (STAGED && POST_STRIDE >= 4)
                        if (step % POST_STRIDE == 3)
                            stage_load((N_PRE + step / POST_STRIDE) % DEPTH, N_PRE + step / POST_STRIDE + DEPTH, kb);
                    if constexpr (G32_SPREAD) {
                        // one landed word per gap: its byte into byte 0 -- and into w's OWN register, as soon as this block is done with it (the A
                        // words of rows 0 .. 5 died with the rows above; the B word of N-subtile ns after step 2 ns + 1): the loop has ONE block
                        // body and one register set for the words in use (see the loop below for why)
                        if (step < 4)
                            e8_set_sa(w, step, shifted_word(e8_sa(land, step), kb + 1));
                        else if (step == 5 || step == 7)
                            e8_set_sa(w, 4 + (step - 5) / 2, shifted_word(e8_sa(land, 4 + (step - 5) / 2), kb + 1));
                        else if (step % 2 == 0 && step <= 14)           // sb[0] @ 4, sb[1] @ 6, ... sb[5] @ 14 (last read at 2 ns + 1)
                            w.sb[step / 2 - 2] = shifted_word(land.sb[step / 2 - 2], kb + 1);
                    }
                    if constexpr (QUAD_SPREAD && J == 3) {
                        if (step < 4)
                            e8_set_sa(w, step, moved_word(e8_sa(land, step)));
                        else if (step == 5 || step == 7)
                            e8_set_sa(w, 4 + (step - 5) / 2, moved_word(e8_sa(land, 4 + (step - 5) / 2)));
                        else if (step % 2 == 0 && step <= 14)
                            w.sb[step / 2 - 2] = moved_word(land.sb[step / 2 - 2]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (G32L && !G32_SPREAD) {
                    asm volatile("s_nop 4" ::: "memory");           // the last MFMAs' scale operands -> VALU writes of the same registers
                    shift_down(w, land, g32_sh);
                    asm volatile("s_nop 3" ::: "memory");           // VALU-written scale registers -> the next block's MFMAs
                }
                // (G32_SPREAD: the last four words are deferred into the next block's first gaps: no wait here)
                const int a_free = a_cur;
                a_cur = a_nxt;
                a_nxt = a_fill;
                a_fill = a_free;
                b_cur ^= B_BYTES;
            };
            using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
            using Yes = std::true_type; using No = std::false_type;
            int kb = 0;
            if constexpr (G32L) {
                // one word per row and K block, every lane's byte shifted into byte 0.  ONE block body in a plain loop: `cur` holds the words in
                // use, `nxt` receives the next block's and is shifted back into `cur` as its registers fall free.  (Until the K-grouped kernels
                // the loop alternated the two sets over two bodies plus a third for an odd block count; with three bodies the accumulators meet at
                // join points, hipcc resolves those with accumulator copies it places where it likes -- e.g. in front of the odd block -- and a
                // copy taken there misses that block's MFMAs: seen as acc[2][7] of every tile with an odd K-block count losing its last block once
                // the FP32 reduce-add epilogue changed the register pressure.  One body, one definition chain, no copies.)
                for (; kb < num_kb; ++kb)
                    block(I0{}, No{}, No{}, cur, kb, nxt);
            } else {
            for (; kb + 4 <= num_kb; kb += 4) {                 // whole K quads: byte select by op_sel, no shifts
                block(I0{}, No{}, No{}, cur, kb, nxt);
                block(I1{}, Yes{}, No{}, cur, kb + 1, nxt);
                block(I2{}, No{}, Yes{}, cur, kb + 2, nxt);
                block(I3{}, No{}, No{}, cur, kb + 3, nxt);
                if constexpr (!QUAD_SPREAD_FORM) {              // (QUAD_SPREAD: moved inside the blocks)
                    cur = nxt;
                    asm volatile("s_nop 3" ::: "memory");       // VALU-written scale registers -> MFMA
                }
            }
            // K tail (k % 512 != 0): up to three more blocks out of the last quad's words (loaded by the last whole quad, or by
            // the prologue when there is none), the byte shifted down by VALU
            if constexpr (MS == 4)          // (the 256-row form takes whole quads only -- host check: a fifth copy of its 64-step
                                            // block body pushes hipcc into keeping the accumulators in memory)
            for (; kb < num_kb; ++kb) {
                E8LandingQ w;
                const int shift = (kb & 3) * 8;
                #pragma unroll
                for (int q = 0; q < MS / 4; ++q)
                    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                        w.sa[q][e] = static_cast<int>(static_cast<unsigned>(cur.sa[q][e]) >> shift);
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    w.sb[ns] = static_cast<int>(static_cast<unsigned>(cur.sb[ns]) >> shift);
                asm volatile("s_nop 3" ::: "memory");
                block(I0{}, No{}, No{}, w, kb, nxt);
            }
            }   // (!G32L)
            if (p.dbg != nullptr) t_loop1 = DG_STAMP_CLOCK();
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");   // last MFMA -> accumulator reads; the tail's re-read pieces
            if constexpr (MNK)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the last block's asm fragment reads: their registers are free game for the epilogue
            __syncthreads();
            }   // (default schedule)
        }
        if (TABSK && sk_pieces >= 2) {
            // the piece's FP32 partial tile, row-major [BM][BN], into its workspace slab (accumulator -> (row, column) as store_tile's interleaved
            // rows and permuted columns); one M-subtile at a time, as below
            float* slab = reinterpret_cast<float*>(static_cast<uint8_t*>(p.sk_workspace) + 4096) +
                          (static_cast<int64_t>(tile) * sk_pieces + sk_piece) * (BM * BN);
            const int lg = lane >> 4;
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    asm volatile("" : "+a"(acc[ms][ns]));
                    const v4f v = acc[ms][ns];
                    const int row = wm * WM + (lane & 15) * MS + ms;
                    const int col = wn * WN + (ns >> 2) * 64 + lg * 8 + ((ns & 3) >> 1) * 32 + (ns & 1) * 4;
                    *reinterpret_cast<v4f*>(slab + row * BN + col) = v;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (!MNK && p.d_dtype == 0 && !p.accumulate && p.d_vec_ok && n_base + WN <= p.n) {
            // BF16 full-line stores, one M-subtile at a time: 32 accumulator registers leave the AGPRs, are packed, exchanged and
            // stored before the next 32 are touched (left alone hipcc reads all of them up front and spills half).
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
                #pragma unroll
                for (int g = 0; g < NS / 4; ++g) {
                    // (the empty asm re-defines the four accumulators as AGPR values HERE: the copies into VGPRs that ordinary
                    // code needs are made right behind an asm output, i.e. at this point and not at the loop exit)
                    asm volatile("" : "+a"(acc[ms][4 * g]), "+a"(acc[ms][4 * g + 1]), "+a"(acc[ms][4 * g + 2]), "+a"(acc[ms][4 * g + 3]));
                    const v4f quad4[4] = {acc[ms][4 * g], acc[ms][4 * g + 1], acc[ms][4 * g + 2], acc[ms][4 * g + 3]};
                    store_rows_full_line<MS, true>(p, t, d_group * p.d_sg, quad4, ms, m_base, n_base + 64 * g);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (p.d_dtype != 0 && p.accumulate && p.d_vec_ok && n_base + WN <= p.n && p.head_lr == 0 && t.m_end > t.m_begin && t.zero_to <= t.zero_from) {
            // FP32 reduce-add (D += A B^T: the weight-gradient forms -- dense recipe (1, 1, gran_k) with packed scales, the K-grouped GEMM), round 6:
            // the old values of FOUR M-subtiles x all N-subtiles (4 * NS loads of 16 bytes per lane, 4 * NS * 4 VGPRs) are in flight before the
            // first add -- MS / 4 memory round trips per wave tile instead of 2 MS through store_tile (one per M-subtile and half: 16 trips of ~2 us
            // were 35 us of a 117 us call at 4096 x 4096 x 7168 and half of the K-grouped call).  Loads from a row clamped into the tile's rows
            // (always inside D), stores predicated; an accumulator leaves its AGPRs right in front of its add.
            static_assert(MS % 4 == 0, "four M-subtiles per round trip");
            float* dbase = reinterpret_cast<float*>(p.d) + d_group * p.d_sg;
            const int lg = lane >> 4;
            int coff[NS];
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                coff[ns] = MNK ? n_base + ns * 16 + lg * 4 : n_base + (ns >> 2) * 64 + lg * 8 + ((ns & 3) >> 1) * 32 + (ns & 1) * 4;
            // (MNK: natural order -- accumulator (ms, ns) holds rows m_base + 16 ms + i, columns n_base + 16 ns + 4 g .. + 3)
            auto row_of = [&](int ms) { return MNK ? m_base + ms * 16 + (lane & 15) : m_base + (lane & 15) * MS + ms; };
            #pragma unroll
            for (int mb = 0; mb < MS; mb += 4) {
                v4f old[4][NS];
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = row_of(mb + u);
                    const float* src = dbase + static_cast<int64_t>(imin(imax(row, t.m_begin), t.m_end - 1)) * p.d_sm;
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
#if DG_RMW_NT & 1            // (tuning builds: non-temporal policy on the old values' loads (1) / the stores (2))
                        old[u][ns] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(src + coff[ns]));
#else
                        old[u][ns] = *reinterpret_cast<const v4f*>(src + coff[ns]);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
                #pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int row = row_of(mb + u);
                    float* dst = dbase + static_cast<int64_t>(row) * p.d_sm;
                    const bool live = row >= t.m_begin && row < t.m_end;
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        asm volatile("" : "+a"(acc[mb + u][ns]));
                        const v4f v = acc[mb + u][ns] + old[u][ns];
                        if (live)
#if DG_RMW_NT & 2
                            __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst + coff[ns]));
#else
                            *reinterpret_cast<v4f*>(dst + coff[ns]) = v;
#endif
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // every other output form through the shared store_tile, in two halves of four N-subtiles (64 columns): the column
            // map of subtile ns is n_base + (ns >> 1) * 32 + ..., so subtiles 4..7 are subtiles 0..3 of a tile 64 columns further right
            auto store_half = [&](auto gc) {
                constexpr int G = decltype(gc)::value;
                v4f out[MS][4];
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        asm volatile("" : "+a"(acc[ms][4 * G + j]));
                        out[ms][j] = acc[ms][4 * G + j];
                    }
                if constexpr (MNK)      // natural rows and columns: subtiles 4 G .. 4 G + 3 are the 64 columns from n_base + 64 G
                    store_tile<MS, 4, false, false, true>(p, t, d_group * p.d_sg, out, m_base, n_base + 64 * G);
                else
                store_tile<MS, 4, true>(p, t, d_group * p.d_sg, out, m_base, n_base + 64 * G);
            };
            store_half(std::integral_constant<int, 0>{});
            if constexpr (NS == 8)
                store_half(std::integral_constant<int, 1>{});
        }
        if (p.dbg != nullptr && tile_id == static_cast<int>(blockIdx.x) && pass == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_stamp(p, NW, 0, t_entry);
            dbg_stamp(p, NW, 1, t_loop0);
            dbg_stamp(p, NW, 2, t_loop1);
            dbg_stamp(p, NW, 3, DG_STAMP_CLOCK());
        }
        advance();
    }
}

template <int BM, int BN, int QV = 0, bool STAGED = false, int WAVES_N = 2, bool K_TAIL = false, int HS = 0, bool TABSK = false, bool G32 = false, bool KG = false,
          bool MNK = false>
__global__ __launch_bounds__(128 * WAVES_N)
void dg_fp8_gemm_quad_e8_kernel(const GemmParams p) {
    quad_e8_kernel_body<BM, BN, QV, STAGED, WAVES_N, K_TAIL, HS, TABSK, G32, KG, MNK>(p);
}

// Second phase of the TABSK remainder walk: one workgroup per (remainder tile, 32-row quarter) adds the tile's partial slabs in piece order
// (bit-repeatable) and stores the rows that belong to the group as BF16.  The grid is an upper bound; tile count and pieces come from the same
// device-side tile list as in the first phase.  Padding blocks were zero-filled by the first phase.
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_e8_tab_reduce_kernel(const GemmParams p) {
    constexpr int BM = 128, BN = 256;
    MaskedWalk walk;
    walk.table_mask = contiguous_tile_mask(p.layout, p.m, p.table_mode, &walk.block_group), walk.have_block_groups = true;
    const int tiles = table_count(p, walk) * p.num_n_tiles, pieces = table_pieces(p, tiles);
    const int tile = blockIdx.x >> 2, quarter = blockIdx.x & 3;
    if (pieces < 2 || tile >= tiles)
        return;
    const Tile t = get_tile<BM, BN>(p, tile, walk, 0);
    if (!t.valid || t.m_end <= t.m0)
        return;
    const float* slab = reinterpret_cast<const float*>(static_cast<const uint8_t*>(p.sk_workspace) + 4096) + static_cast<int64_t>(tile) * pieces * (BM * BN);
    const int c4 = threadIdx.x & 63, r0 = quarter * 32 + (threadIdx.x >> 6);
    const int col = t.n0 + c4 * 4;
    uint16_t* d = reinterpret_cast<uint16_t*>(p.d);
    #pragma unroll 2
    for (int r = r0; r < quarter * 32 + 32; r += 4) {
        const int row = t.m0 + r;
        if (row >= t.m_end)
            break;
        v4f sum = *reinterpret_cast<const v4f*>(slab + r * BN + c4 * 4);
        for (int s = 1; s < pieces; ++s)
            sum += *reinterpret_cast<const v4f*>(slab + static_cast<int64_t>(s) * (BM * BN) + r * BN + c4 * 4);
        uint16_t* dst = d + static_cast<int64_t>(row) * p.d_sm + col;
        if (col + 4 <= p.n && p.d_vec_ok) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf16(sum[0], sum[1]), pack_bf16(sum[2], sum[3]));
        } else {
            #pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < p.n)
                    dst[e] = static_cast<uint16_t>(pack_bf16(sum[e], 0.f) & 0xffffu);
        }
    }
}
#endif

}  // namespace dg
