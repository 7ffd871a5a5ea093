// CDNA4 (gfx950) device code of the FP8 blockwise-scaled GEMM path.
//
// What it computes (reference semantics: deep_gemm/include/deep_gemm/impls/sm90_fp8_gemm_1d2d.cuh:283-347,416-418):
//   D[m,n] = cast( sum_kb (sfa[m,kb] * sfb[n/128,kb]) * (sum_{k in 128-block kb} A[m,k] * B[n,k]) )
// How it is built for MI355X (nothing here mirrors the reference's TMA/WGMMA structure):
//   * one v_mfma_f32_16x16x128_f8f6f4 consumes exactly one 128-K scale block of a 16x16 output tile, so the FP32
//     promotion is `acc += (sfa*sfb) * mfma(...)` with a zero C operand -- 4 VALU FMAs per 32-cycle MFMA;
//   * operand roles are swapped (B rows feed the MFMA's A slot, A rows its B slot) so that in the C/D register map
//     (col = lane & 15, row = 4 * (lane >> 4) + reg) every lane owns ONE m: one SFA value per lane per 16-row subtile,
//     SFB is wave-uniform;
//   * LDS tiles are [row][128 B] with the 16-byte chunk index XOR-ed by (row & 7); lane (r, g) of a fragment reads
//     chunks g and g+4 of row r (a K permutation shared by both operands, so the contraction is unchanged), which
//     makes both ds_read_b128 of a fragment bank-conflict free;
//   * the fast path fills LDS with global_load_lds_dwordx4 (LDS-DMA): 8 lanes cover one full 128-byte line of a row,
//     the swizzle is applied on the per-lane SOURCE address (the LDS image of a wave instruction is lane-linear);
//   * B-tile rows are stored permuted so that after the last K block every lane holds 8 consecutive n of one m per pair
//     of N-subtiles and the four lanes of a row write 64 contiguous bytes per store instruction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace dg {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

enum GemmType : int { kNormal = 0, kContiguous = 1, kContiguousPsum = 2, kMasked = 3, kKGrouped = 4 };
constexpr int kMaxKGroups = 64;

struct GemmParams {
    const uint8_t* a;
    const float* sfa;
    const uint8_t* b;
    const float* sfb;
    void* d;
    const int32_t* layout;          // grouped_layout (contiguous) or masked_m (masked)
    int m, n, k, num_groups;        // masked: m = m_max (rows per group)
    int64_t a_sg, a_sm, a_sk;       // strides in elements (= bytes for FP8)
    int64_t b_sg, b_sn, b_sk;
    int64_t sfa_sg, sfa_sm, sfa_sk;
    int64_t sfb_sg, sfb_sn, sfb_sk;
    int64_t d_sg, d_sm;
    int sfb_gran_n;                 // 128 or 1
    // K-grouped launch (gemm_type kKGrouped, per-column-SFB kernel only): group g owns K range [kg_prefix[g], kg_prefix[g+1])
    // and writes D[g]; kg_blocks != 0: the operands are the groups' K-major [m, k_g] / [n, k_g] matrices stored one after
    // another (row stride k_g); 0: column ranges of one K-major matrix (row strides a_sm / b_sn).
    int kg_blocks;
    int kg_prefix[kMaxKGroups + 1];
    // kg_psum != 0: the K ranges are read on the device instead -- `layout[g]` is group g's END along K (the reference's psum layout,
    // scheduler/gemm.cuh:74-85), the group starts at the previous end rounded up to 128 and whole 128-blocks are computed (the rows
    // between an end and the next multiple of 128 hold zeros by the layout's contract); k = rows of the operands (a multiple of 128)
    int kg_psum;
    int head_lr, head_mid, head_right;  // epilogue column map of fp8_gemm_nt_skip_head_mid: D column of GEMM column n is
                                    // n + (n + head_right) / head_lr * head_mid (head_lr = left + right; 0 = identity)
    int d_dtype;                    // 0 bf16, 1 fp32
    int accumulate;
    int gemm_type;
    int m_alignment;                // contiguous layouts
    int num_m_tiles, num_n_tiles;   // per group
    int group_m;                    // tile-order swizzle: m-tiles per L2 group
    int d_vec_ok;                   // 16-byte aligned D rows => vector stores
    int d_nt;                       // non-temporal policy on the full-line BF16 output stores (large outputs: set by the host)
    long long* dbg;                 // tuning aid (normally null): per wave {kernel entry, K loop begin, K loop end, after stores} s_memtime ticks
    // K-split tail of a persistent launch (duo kernels built with SPLITK): tiles [sk_first_tile, sk_first_tile + sk_tiles) are cut
    // into sk_factor K pieces, each written as an FP32 partial tile to sk_workspace + 4 KiB: [sk_tiles][sk_factor][BM * BN]; a second
    // kernel (dg_split_k_reduce_kernel, same stream) sums them in piece order.  The first 4 KiB of the workspace hold the tile tables
    // of the contiguous layout's group-relative tiling (see tile_table below); nothing else depends on their contents.
    void* sk_workspace;
    int sk_first_tile, sk_tiles, sk_factor;
    // Tile table of the contiguous layout (device memory, written by dg_build_contiguous_tile_table_kernel in front of the GEMM on the
    // same stream; nullptr = the fixed tile grid): tile_table[0] = number of M tiles of this launch, tile_table[1 + i] = first row of M
    // tile i (BM rows that belong to ONE group, or a block of padding rows).  With a table and SPLITK EVERY tile of the launch is cut
    // along K: the tile count is only known on the device, so are the pieces -- min(sk_factor, sk_capacity / tiles), see table_pieces().
    const int32_t* tile_table;
    int sk_capacity;                // FP32 partial tiles the workspace holds (table launches)
    // table_mode 1 / 2: the same tiling WITHOUT the table kernel, for layouts of at most 64 blocks of 128 rows (M <= 8192): every
    // workgroup derives the tile list itself from one 64-lane load of the blocks' group ids (contiguous_tile_mask: the first blocks of
    // the 256-row tiles (1) or the 128-row remainders and padding blocks (2) as a bit mask).  tile_table stays nullptr.
    int table_mode;
    int skinny_cols;                // skinny kernel with two N-subtiles: columns per workgroup (17 .. 32, multiple of 4); 0 = 16 x NSUB
    // K split in two WITHOUT the reduction kernel (table launches of the contiguous layout): piece 0 of a tile writes its FP32 partial and
    // raises the tile's flag (uint32 at sk_workspace[tile], release at agent scope) to this value; piece 1 -- dispatched later, or run
    // later by the same persistent workgroup -- waits for it, adds the partial to its own accumulators (the same two-operand sum as the
    // reduction kernel's), stores the tile and clears the flag.  0 = off.  The value is a NaN bit pattern that changes per launch: nothing
    // else that ever lands in the workspace header (tile tables, zeros) equals it.
    unsigned sk_exchange;
};

// The clock of the debug stamps: the shader-clock counter (s_memtime: per CU, NOT comparable between CUs -- good for durations inside a wave) or, in
// -DDG_STAMP_REALTIME tuning builds, the chip-wide 100 MHz counter (s_memrealtime: comparable between workgroups, 10 ns resolution; tools/c2_end_time_histogram.py)
#ifdef DG_STAMP_REALTIME
#define DG_STAMP_CLOCK() static_cast<long long>(__builtin_amdgcn_s_memrealtime())
#else
#define DG_STAMP_CLOCK() __builtin_amdgcn_s_memtime()
#endif
__device__ __forceinline__ void dbg_stamp(const GemmParams& p, int waves_per_block, int slot, long long t) {
    if (p.dbg != nullptr && (threadIdx.x & 63) == 0)
        p.dbg[(static_cast<long long>(blockIdx.x) * waves_per_block + (threadIdx.x >> 6)) * 4 + slot] = t;
}

struct Tile {
    int m0, n0;       // first row (within the group's A/D for masked, global otherwise) / first column
    int group;        // B / SFB group (masked: also A / SFA / D group)
    int m_end;        // rows >= m_end are not computed from A (clamped loads)
    int zero_from;    // rows in [zero_from, zero_to) are stored as zeros; rows outside [m_begin, m_end) and that range are skipped
    int m_begin;      // rows < m_begin are not stored by this pass (two-pass tiles of the contiguous layout)
    int zero_to;
    bool valid;
    bool second_pass; // the tile needs another pass (its two alignment-sized halves belong to different groups)
};

// Values that are wave-uniform in fact but not provably so (tile coordinates derived from loaded group sizes ...): hipcc wraps every
// buffer operation whose descriptor is built from them in a waterfall loop (v_readfirstlane x 4, two 64-bit compares, exec save / restore,
// a branch: ~12 instructions per LDS-DMA piece).  Routing them through readfirstlane makes the descriptor an SGPR value.
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* uniform_pointer(T* ptr) {
    const uint64_t v = reinterpret_cast<uint64_t>(ptr);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<int>(v));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<int>(v >> 32));
    return reinterpret_cast<T*>((static_cast<uint64_t>(hi) << 32) | lo);
}
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// Block b runs on XCD b % 8 (observed, speed only): give each XCD a contiguous chunk of the tile order, then walk
// the tiles in groups of `group_m` m-tiles so that a chunk is a compact rectangle sharing A and B panels in its L2.
// a / b for 0 <= a < 2^22, 0 < b < 2^22 (tile counts): one reciprocal and a one-step correction instead of the ~40-instruction integer
// division sequence -- three of those sat between a workgroup's entry and its first load (~1.1 k cycles, tools/prologue_stamps.py).
// (a + 0.5) / b is at least 0.5 / b away from an integer and the float error is below a * 2^-22 / b, so the truncation is already exact;
// the correction makes it independent of the reciprocal's accuracy.
__device__ __forceinline__ int div_small(int a, int b) {
    int q = static_cast<int>((static_cast<float>(a) + 0.5f) * __builtin_amdgcn_rcpf(static_cast<float>(b)));
    const int rem = a - q * b;
    q += (rem >= b ? 1 : 0) - (rem < 0 ? 1 : 0);
    return __builtin_amdgcn_readfirstlane(q);              // (tile ids are wave-uniform; the float detour hides that from the compiler)
}

__device__ __forceinline__ void swizzled_tile(int bid, int nwg, int num_m_tiles, int num_n_tiles, int group_m,
                                              int& mt, int& nt) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int per_group = group_m * num_n_tiles;
    const bool small = nwg < (1 << 22);                   // (more tiles than that: > 17 GB of output even with the smallest tile)
    const int grp = small ? div_small(lin, per_group) : lin / per_group, in_grp = lin - grp * per_group;
    const int first_m = grp * group_m;
    const int h = imin(num_m_tiles - first_m, group_m);
    nt = small ? div_small(in_grp, h) : in_grp / h;
    mt = first_m + in_grp - nt * h;
}

// K pieces per tile of a table launch with SPLITK: the host's wish (sk_factor) capped by what the workspace holds for the tile count the
// table reports; below 2 the tiles are computed whole and stored directly.  The duo kernel and the reduction kernel both call this.
__device__ __forceinline__ int table_pieces(const GemmParams& p, int tiles) {
    if (tiles <= 0 || p.sk_factor < 2)
        return 1;
    // one round of pieces over the workgroup slots of the launch (sk_tiles carries the slot count on a table launch): 64 tiles on 256
    // CUs -> 4 pieces, 80 -> 3, 96 -> 2; more tiles than half the slots -> whole tiles
    const int pieces = imin(imin(p.sk_factor, p.sk_tiles / tiles), p.sk_capacity / tiles);
    return pieces >= 2 ? pieces : 1;
}

// Maps a linear tile id to a tile for every GEMM type (reference scheduler semantics:
// deep_gemm/include/deep_gemm/scheduler/gemm.cuh:156-237, :311-319).  `state` carries the masked-layout walk.
// block_group (round 5): lane b holds the group id of the layout's 128-row block b (the very load the tile mask is built from), so that a table
// tile's group is a lane read instead of a SECOND global load that depends on the first (a tile's first LDS-DMA piece waits for its group id:
// the B base; tools/prologue_stamps.py put 12.4 k ticks in front of the first load of a table launch against 7 k of a dense one).
struct MaskedWalk { int group = 0; int cum_m_tiles = 0; unsigned long long table_mask = 0; int block_group = 0; bool have_block_groups = false; };

__device__ __forceinline__ bool table_launch(const GemmParams& p) { return p.tile_table != nullptr || p.table_mode != 0; }

// The tile list of dg_build_contiguous_tile_table_kernel as a bit mask over the layout's (at most 64) blocks of 128 rows, computed by a
// whole wave: a run of blocks of one group is cut into 256-row tiles from ITS first block (mode 1: bit = first block of such a tile); an odd
// run leaves its last block, and every block of padding rows (-1) stands alone (mode 2).  Call with all 64 lanes active.
__device__ __forceinline__ unsigned long long contiguous_tile_mask(const int32_t* __restrict__ layout, int m, int mode, int* block_group = nullptr) {
    const int lane = threadIdx.x & 63, nb = (m + 127) / 128;
    const int g = lane < nb ? layout[lane * 128] : -2;
    if (block_group != nullptr)
        *block_group = g;
    const int g_prev = __shfl_up(g, 1, 64);
    const bool starts = lane == 0 || g != g_prev || g < 0;
    const unsigned long long start_mask = __ballot(starts);
    const unsigned long long at_or_below = start_mask & ((2ull << lane) - 1ull);
    const int pos = lane - (63 - __builtin_clzll(at_or_below));                           // position inside the run (lane 0 always starts one)
    const bool next_same = g >= 0 && lane + 1 < nb && !((start_mask >> (lane + 1)) & 1ull);
    const bool big_first = g >= 0 && (pos & 1) == 0 && next_same;
    const bool rem = lane < nb && (g < 0 || ((pos & 1) == 0 && !next_same));
    return __ballot(mode == 1 ? big_first : rem);
}

__device__ __forceinline__ int kth_set_bit(unsigned long long mask, int k) {
    for (int i = 0; i < k; ++i)
        mask &= mask - 1ull;
    return __builtin_ctzll(mask);
}
__device__ __forceinline__ int table_count(const GemmParams& p, const MaskedWalk& w) {
    return p.tile_table != nullptr ? p.tile_table[0] : __builtin_popcountll(w.table_mask);
}
__device__ __forceinline__ int table_first_row(const GemmParams& p, const MaskedWalk& w, int i) {
    return p.tile_table != nullptr ? p.tile_table[1 + i] : kth_set_bit(w.table_mask, i) * 128;
}

template <int BM, int BN>
__device__ __forceinline__ Tile get_tile(const GemmParams& p, int tile_id, MaskedWalk& walk, int pass = 0) {
    Tile t;
    t.valid = true;
    t.second_pass = false;
    if (p.gemm_type == kMasked) {
        // Persistent walk over groups; masked_m lives on the device (README: the CPU never learns the counts).
        int nmt;
        while (true) {
            if (walk.group >= p.num_groups) { t.valid = false; return t; }
            nmt = (imin(p.layout[walk.group], p.m) + BM - 1) / BM;
            if (tile_id < (walk.cum_m_tiles + nmt) * p.num_n_tiles)
                break;
            walk.cum_m_tiles += nmt;
            ++walk.group;
        }
        const int local = tile_id - walk.cum_m_tiles * p.num_n_tiles;
        const int nt = local / nmt, mt = local - nt * nmt;     // m fastest: the tiles sharing a B panel run together
        t.group = walk.group;
        t.m0 = mt * BM;
        t.n0 = nt * BN;
        t.m_end = imin(p.layout[walk.group], p.m);
        t.zero_from = t.zero_to = t.m0 + BM;
        t.m_begin = t.m0;
        return t;
    }
    if (table_launch(p)) {
        // contiguous layout through a tile table: group-relative M tiles (no tile straddles two groups), see GemmParams::tile_table
        const int count = table_count(p, walk), total = count * p.num_n_tiles;
        if (tile_id >= total) { t.valid = false; return t; }
        int mt, nt;
        swizzled_tile(tile_id, total, count, p.num_n_tiles, p.group_m, mt, nt);
        t.m0 = table_first_row(p, walk, mt);
        t.n0 = nt * BN;
        t.m_begin = t.m0;
        t.m_end = imin(t.m0 + BM, p.m);
        t.zero_from = t.zero_to = t.m_end;
        const int g = walk.have_block_groups ? __builtin_amdgcn_readlane(walk.block_group, __builtin_amdgcn_readfirstlane(t.m0 >> 7)) : p.layout[t.m0];
        if (g < 0) { t.m_end = t.m0; t.zero_from = t.m0; t.zero_to = imin(t.m0 + BM, p.m); }
        t.group = imax(g, 0);
        return t;
    }
    const int num_tiles = p.num_m_tiles * p.num_n_tiles;
    int kgroup = 0;
    if (p.gemm_type == kKGrouped) {                 // group-major tile order: every group is a full M x N output
        kgroup = tile_id / num_tiles;
        if (kgroup >= p.num_groups) { t.valid = false; return t; }
        tile_id -= kgroup * num_tiles;
    }
    if (tile_id >= num_tiles) { t.valid = false; return t; }
    int mt, nt;
    swizzled_tile(tile_id, num_tiles, p.num_m_tiles, p.num_n_tiles, p.group_m, mt, nt);
    t.m0 = mt * BM;
    t.n0 = nt * BN;
    t.group = kgroup;
    t.m_end = p.m;
    t.zero_from = t.zero_to = imin(t.m0 + BM, p.m);    // zero rows never reach past the end of D (m need not be a multiple of BM)
    t.m_begin = t.m0;
    if (p.gemm_type == kContiguous && BM > p.m_alignment) {
        // A tile of two alignment-sized halves (BM == 2 * m_alignment, checked by the host): the halves may belong to
        // different groups (=> two passes over the tile, one per half, each with its own B) or be padding (=> zeros).
        const int half = p.m_alignment, mid = t.m0 + half;
        const int g0 = p.layout[t.m0], g1 = mid < p.m ? p.layout[mid] : -2;       // -2: the half does not exist
        const int end1 = imin(mid + half, p.m);
        const bool c0 = g0 >= 0, c1 = g1 >= 0, has1 = g1 != -2;
        if (c0 && c1 && g0 != g1) {
            // two different groups: pass 0 -> first half, pass 1 -> second half
            t.second_pass = (pass == 0);
            t.group = pass == 0 ? g0 : g1;
            t.m_begin = pass == 0 ? t.m0 : mid;
            t.m_end = pass == 0 ? mid : end1;
        } else if (c0 && c1) {                      // one group owns both halves
            t.group = g0;
            t.m_end = end1;
        } else if (c0) {                            // second half is padding (zero rows) or does not exist
            t.group = g0;
            t.m_end = imin(mid, p.m);               // (m need not be a multiple of the alignment: nothing is stored past row m)
            t.zero_from = t.m_end;
            t.zero_to = has1 ? end1 : t.m_end;
        } else if (c1) {                            // first half is padding
            t.group = g1;
            t.m_begin = mid;
            t.m_end = end1;
            t.zero_from = t.m0;
            t.zero_to = mid;
        } else {                                    // nothing to compute
            t.group = 0;
            t.m_begin = t.m_end = t.m0;
            t.zero_from = t.m0;
            t.zero_to = has1 ? end1 : imin(mid, p.m);
        }
    } else if (p.gemm_type == kContiguous) {
        const int g = p.layout[t.m0];
        if (g < 0) { t.m_end = t.m0; t.zero_from = t.m0; }
        t.group = imax(g, 0);
    } else if (p.gemm_type == kContiguousPsum) {
        // Group g owns rows [align(end[g-1], alignment), end[g]); the gap up to the next aligned start is zero-filled.
        int start = 0;
        t.m_end = t.m0;                                    // rows past the last group: left untouched
        for (int g = 0; g < p.num_groups; ++g) {
            const int end = p.layout[g];
            const int next = (end + p.m_alignment - 1) / p.m_alignment * p.m_alignment;
            if (t.m0 >= start && t.m0 < next) {
                t.group = g;
                t.m_end = imax(imin(end, p.m), t.m0);
                t.zero_from = t.m_end;
                break;
            }
            start = next;
        }
    }
    return t;
}

__device__ __forceinline__ v4f mfma_fp8_k128(const v8i& rows_operand, const v8i& cols_operand) {
    // Zero scale operands select the unscaled encoding (v_mfma_f32_16x16x128_f8f6f4, cbsz = blgp = 0 => e4m3 x e4m3).
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(rows_operand, cols_operand, zero, 0, 0, 0, 0, 0, 0);
}

// LDS tile geometry shared by every kernel: row r of a tile occupies bytes [r*128, r*128+128); logical 16-byte chunk c
// of that row is stored at chunk position c ^ (r & 7).
__device__ __forceinline__ int lds_chunk_offset(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// Fragment of 16 rows starting at `tile_rows` (row index multiple of 16): lane (r = lane & 15, g = lane >> 4) gets the
// 32 bytes {chunk g, chunk g + 4} of row r.
__device__ __forceinline__ v8i load_fragment(const uint8_t* tile_rows, int frag_off) {
    const v4i lo = *reinterpret_cast<const v4i*>(tile_rows + frag_off);
    const v4i hi = *reinterpret_cast<const v4i*>(tile_rows + (frag_off ^ 64));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// B-tile row permutation: LDS row position p (within the BN-row tile) holds global column n0 + perm(p), chosen so that
// the MFMA row slot i = 4 * lg + r of N-subtile ns lands on column wave_n0 + (ns >> 1) * 32 + lg * 8 + (ns & 1) * 4 + r:
// a lane then holds 8 consecutive BF16 outputs (16 bytes) per pair of N-subtiles, and the four lanes lg = 0..3 of a row
// write 64 CONTIGUOUS bytes with one store instruction (full 64-byte sectors instead of 16-byte pieces at a 32-byte
// stride: the epilogue of one tile per CU is not overlapped with anything, its store efficiency is wall time).
template <int WN>
__device__ __forceinline__ int b_row_perm(int p) {
    const int w = p / WN, q = p % WN, ns = q >> 4, i = q & 15;
    if (WN % 32 != 0 && ns == WN / 16 - 1)      // (WN = 112, the 256 x 224 tile: the unpaired seventh subtile keeps its natural column order)
        return w * WN + ns * 16 + i;
    return w * WN + (ns >> 1) * 32 + (i >> 2) * 8 + (ns & 1) * 4 + (i & 3);
}

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v;
    v[0] = static_cast<__bf16>(lo);
    v[1] = static_cast<__bf16>(hi);
    return __builtin_bit_cast(uint32_t, v);
}
__device__ __forceinline__ float bf16_lo(uint32_t v) { return __builtin_bit_cast(float, v << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t v) { return __builtin_bit_cast(float, v & 0xffff0000u); }
__device__ __forceinline__ float round_bf16(float x) { return bf16_lo(pack_bf16(x, 0.f)); }

// Output column of GEMM column n (reference: epilogue/transform.cuh:15-22, EpilogueHeadSplits::apply_index_n).  Runs of 8
// columns starting at a multiple of 8 stay contiguous when left, mid and right are multiples of 8 (the vector paths).
__device__ __forceinline__ int d_col(const GemmParams& p, int n) {
    return p.head_lr > 0 ? n + (n + p.head_right) / p.head_lr * p.head_mid : n;
}

// Full-line BF16 stores of one M-subtile (16 rows) x four N-subtiles (64 columns starting at n_base): the output tail of a
// CU is bound by the NUMBER of store requests (one per 64-byte run: ~5 cycles each), not by bytes.  Lanes r and r + 8 of a
// 16-lane row swap halves through DPP (row_ror:8) so that one store instruction writes 8 rows x 128 contiguous bytes
// instead of 16 rows x 64: half the requests for the same bytes.  Caller: BF16 output, no accumulation, 16-byte aligned
// rows, n_base + 64 <= n.  acc4[j][r] = D[row(ms, lane & 15)][n_base + (j >> 1) * 32 + lg * 8 + (j & 1) * 4 + r].
template <int MS, bool INTERLEAVED_ROWS>
__device__ __forceinline__ void store_rows_full_line_packed(const GemmParams& p, const Tile& t, int64_t d_group_off,
                                                            const uint32_t (&w0)[4], const uint32_t (&w1)[4], int ms, int m_base, int n_base) {
    // w0 / w1: the lane's 8 BF16 columns n_base + lg * 8 .. + 7 (w0) and n_base + 32 + lg * 8 .. + 7 (w1) of its row
    const int lane = threadIdx.x & 63, lg = lane >> 4;
    const bool lo = (lane & 8) == 0;
    const int r7 = lane & 7;
    const int col = n_base + ((lane >> 3) & 1) * 32 + lg * 8;
    uint32_t x[4], y[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t send = lo ? w1[j] : w0[j];
        uint32_t recv;
        // s_nop 1: VALU write -> DPP read of the same VGPR needs two wait states (hipcc pads nothing inside asm)
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(recv) : "v"(send));
        x[j] = lo ? w0[j] : recv;
        y[j] = lo ? recv : w1[j];
    }
    #pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int i = r7 + half * 8;
        const int row = INTERLEAVED_ROWS ? m_base + i * MS + ms : m_base + ms * 16 + i;
        const bool compute_row = row >= t.m_begin && row < t.m_end;
        const bool zero_row = row >= t.zero_from && row < t.zero_to;
        if (!compute_row && !zero_row)
            continue;
        const uint32_t* v = half == 0 ? x : y;
        uint16_t* drow = reinterpret_cast<uint16_t*>(p.d) + d_group_off + static_cast<int64_t>(row) * p.d_sm;
        // Large outputs leave with the non-temporal policy: a GEMM's D is written once and read by nobody in this launch; streaming it
        // past the L2's replacement (and out of the kernel-end write-back) was worth 5-6 % of the whole C2 call on BOTH kernel families
        // (97.2 -> 92.2 us, 86.2 -> 80.9 us; sc1 write-through alone: nothing) -- what hipBLASLt's kernels do ("NTD").  Small outputs
        // (the consumer kernel finds them in the L2) keep the default policy: GemmParams::d_nt is set by the host from the output size.
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        const u4 data = zero_row ? u4{0u, 0u, 0u, 0u} : u4{v[0], v[1], v[2], v[3]};
        u4* addr = reinterpret_cast<u4*>(drow + d_col(p, col));
#ifdef DG_NO_NT
        if (false)
#else
        if (p.d_nt)     // (inline asm: hipcc merges a __builtin_nontemporal_store with the plain store of the other branch and drops the hint;
                        //  s_nop 1: an asm dwordx4 store's data registers must not be rewritten by the next instruction)
#endif
            asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" :: "v"(addr), "v"(data) : "memory");
        else
            *addr = data;
    }
}

template <int MS, bool INTERLEAVED_ROWS>
__device__ __forceinline__ void store_rows_full_line(const GemmParams& p, const Tile& t, int64_t d_group_off,
                                                     const v4f (&acc4)[4], int ms, int m_base, int n_base) {
    uint32_t w0[4], w1[4];
    #pragma unroll
    for (int j = 0; j < 2; ++j) {
        w0[2 * j] = pack_bf16(acc4[j][0], acc4[j][1]);
        w0[2 * j + 1] = pack_bf16(acc4[j][2], acc4[j][3]);
        w1[2 * j] = pack_bf16(acc4[2 + j][0], acc4[2 + j][1]);
        w1[2 * j + 1] = pack_bf16(acc4[2 + j][2], acc4[2 + j][3]);
    }
    store_rows_full_line_packed<MS, INTERLEAVED_ROWS>(p, t, d_group_off, w0, w1, ms, m_base, n_base);
}

// The same stores from the NATURAL column order of the MN-major-B kernels (acc4[ns][r] = D[row][n_base + ns * 16 + lg * 4 + r]): the
// packed words first change lanes so that every lane holds the 8 + 8 consecutive columns the full-line form wants.  With column
// bits c5 c4 c3 c2 = (ns1 ns0 | lg1 lg0) before and (h | lg1 lg0 | j) after, that is a rotation of (ns0, lg1, lg0): one
// v_permlane32_swap (register bit <-> lane bit 5) and one v_permlane16_swap (register bit <-> lane bit 4) per register pair --
// 8 swaps per 16 rows instead of 4x the store requests (measured: the 8-byte-per-lane epilogue took 18.0 k cycles per tile
// against 9.9 k, tools/c3_diag.py).
template <int MS, bool INTERLEAVED_ROWS>
__device__ __forceinline__ void store_rows_full_line_natural(const GemmParams& p, const Tile& t, int64_t d_group_off,
                                                             const v4f (&acc4)[4], int ms, int m_base, int n_base) {
    typedef unsigned int u2 __attribute__((ext_vector_type(2)));
    uint32_t wn[4][2];
    #pragma unroll
    for (int ns = 0; ns < 4; ++ns) {
        wn[ns][0] = pack_bf16(acc4[ns][0], acc4[ns][1]);
        wn[ns][1] = pack_bf16(acc4[ns][2], acc4[ns][3]);
    }
    #pragma unroll
    for (int h = 0; h < 2; ++h)
        #pragma unroll
        for (int d = 0; d < 2; ++d) {
            // lanes 32..63 of the even register <-> lanes 0..31 of the odd one, then odd 16-lane rows of the even <-> even rows of the odd
            const u2 a = __builtin_amdgcn_permlane32_swap(wn[2 * h][d], wn[2 * h + 1][d], false, false);
            const u2 b = __builtin_amdgcn_permlane16_swap(a[0], a[1], false, false);
            wn[2 * h][d] = b[0];
            wn[2 * h + 1][d] = b[1];
        }
    const uint32_t w0[4] = {wn[0][0], wn[0][1], wn[1][0], wn[1][1]}, w1[4] = {wn[2][0], wn[2][1], wn[3][0], wn[3][1]};
    store_rows_full_line_packed<MS, INTERLEAVED_ROWS>(p, t, d_group_off, w0, w1, ms, m_base, n_base);
}

// FP32 reduce-add of a whole wave tile (recipe (1, 1, 128) wgrad / K-grouped GEMMs: D += A B^T): the old values of FOUR subtile rows
// are fetched before the first add, so a wave makes MS / 4 memory round trips instead of MS -- the tail of these kernels is the latency
// of those trips (a 256 x 256 FP32 tile: 8 trips of ~2 us against a K loop of 80-120 us), not bytes.  Loads are unconditional from a
// row clamped into the tile's computed range (always inside D); only the stores are predicated.  COL(ns) = first of the lane's four
// consecutive columns of N-subtile ns.
template <int MS, int NS, bool INTERLEAVED_ROWS, typename ColFn>
__device__ __forceinline__ void reduce_add_tile_fp32(const GemmParams& p, const Tile& t, int64_t d_group_off, v4f (&acc)[MS][NS],
                                                     int m_base, ColFn col_of) {
    static_assert(MS % 4 == 0, "four subtile rows per round trip");
    const int lane = threadIdx.x & 63;
    float* dbase = reinterpret_cast<float*>(p.d) + d_group_off;
    int coff[NS];
    #pragma unroll
    for (int ns = 0; ns < NS; ++ns)
        coff[ns] = d_col(p, col_of(ns));
    #pragma unroll
    for (int mb = 0; mb < MS; mb += 4) {
        v4f old[4][NS];
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = INTERLEAVED_ROWS ? m_base + (lane & 15) * MS + mb + u : m_base + (mb + u) * 16 + (lane & 15);
            const float* src = dbase + static_cast<int64_t>(imin(imax(row, t.m_begin), t.m_end - 1)) * p.d_sm;
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                old[u][ns] = *reinterpret_cast<const v4f*>(src + coff[ns]);
        }
        #pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = INTERLEAVED_ROWS ? m_base + (lane & 15) * MS + mb + u : m_base + (mb + u) * 16 + (lane & 15);
            const bool compute_row = row >= t.m_begin && row < t.m_end;
            const bool zero_row = row >= t.zero_from && row < t.zero_to;
            float* dst = dbase + static_cast<int64_t>(row) * p.d_sm;
            if (zero_row) {
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    *reinterpret_cast<v4f*>(dst + coff[ns]) = v4f{0.f, 0.f, 0.f, 0.f};
            } else if (compute_row) {
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    *reinterpret_cast<v4f*>(dst + coff[ns]) = acc[mb + u][ns] + old[u][ns];
            }
        }
    }
}

// BF16 reduce-add of a whole wave tile (D = bf16(D + bf16(A B^T)), the reference's BF16 accumulation, tests/generators.py:138-140): the old
// values of TWO subtile rows per memory round trip (MS / 2 trips per wave tile instead of one per 16-byte vector, 2 MS); loads from
// a row clamped into the tile's computed range, stores predicated.  n_lane: the lane's first column (+ h * 32 for pair h of N-subtiles).
template <int MS, int NS, bool INTERLEAVED_ROWS>
__device__ __forceinline__ void reduce_add_tile_bf16(const GemmParams& p, const Tile& t, int64_t d_group_off, v4f (&acc)[MS][NS],
                                                     int m_base, int n_lane) {
    static_assert(MS % 2 == 0 && NS % 2 == 0, "two subtile rows per round trip, 8 columns per lane and pair of N-subtiles");
    const int lane = threadIdx.x & 63;
    uint16_t* dbase = reinterpret_cast<uint16_t*>(p.d) + d_group_off;
    int coff[NS / 2];
    #pragma unroll
    for (int h = 0; h < NS / 2; ++h)
        coff[h] = d_col(p, n_lane + h * 32);
    #pragma unroll
    for (int mb = 0; mb < MS; mb += 2) {
        uint4 old[2][NS / 2];
        #pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = INTERLEAVED_ROWS ? m_base + (lane & 15) * MS + mb + u : m_base + (mb + u) * 16 + (lane & 15);
            const uint16_t* src = dbase + static_cast<int64_t>(imin(imax(row, t.m_begin), t.m_end - 1)) * p.d_sm;
            #pragma unroll
            for (int h = 0; h < NS / 2; ++h)
                old[u][h] = *reinterpret_cast<const uint4*>(src + coff[h]);
        }
        #pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int row = INTERLEAVED_ROWS ? m_base + (lane & 15) * MS + mb + u : m_base + (mb + u) * 16 + (lane & 15);
            const bool compute_row = row >= t.m_begin && row < t.m_end;
            const bool zero_row = row >= t.zero_from && row < t.zero_to;
            uint16_t* dst = dbase + static_cast<int64_t>(row) * p.d_sm;
            if (zero_row) {
                #pragma unroll
                for (int h = 0; h < NS / 2; ++h)
                    *reinterpret_cast<uint4*>(dst + coff[h]) = make_uint4(0u, 0u, 0u, 0u);
            } else if (compute_row) {
                #pragma unroll
                for (int h = 0; h < NS / 2; ++h) {
                    const uint32_t o[4] = {old[u][h].x, old[u][h].y, old[u][h].z, old[u][h].w};
                    uint32_t w[4];
                    #pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const v4f v = acc[mb + u][2 * h + j];
                        w[2 * j] = pack_bf16(v[0], v[1]);
                        w[2 * j + 1] = pack_bf16(v[2], v[3]);
                    }
                    #pragma unroll
                    for (int j = 0; j < 4; ++j)
                        w[j] = pack_bf16(bf16_lo(o[j]) + bf16_lo(w[j]), bf16_hi(o[j]) + bf16_hi(w[j]));
                    *reinterpret_cast<uint4*>(dst + coff[h]) = make_uint4(w[0], w[1], w[2], w[3]);
                }
            }
        }
    }
}

// Epilogue.  acc[ms][ns][r] = D[m = m_base + ms*16 + (lane & 15)][n = n_base + lg*4*NS + ns*4 + r].
// accumulate => reduce-add in D's dtype (reference: epilogue/sm100_store_cd.cuh:121-129).
// INTERLEAVED_ROWS: acc[ms] belongs to row m_base + (lane & 15) * MS + ms instead (the duo kernel's A-row permutation).
// NATURAL_COLS: acc[ms][ns][r] belongs to column n_base + ns*16 + lg*4 + r (B rows in their natural order in the LDS
// image: the MN-major operand path); only the FP32 vector path and the element-wise paths exist for it.
// ms_only >= 0: only that M-subtile is stored (the K-split reduction kernel works on one subtile row per workgroup).
// BATCH_RMW: FP32 reduce-add through reduce_add_tile_fp32 (the kernels whose every call accumulates; costs 64 live registers).
template <int MS, int NS, bool INTERLEAVED_ROWS = false, bool NT_STORE = false, bool NATURAL_COLS = false, bool BATCH_RMW = false>
__device__ __forceinline__ void store_tile(const GemmParams& p, const Tile& t, int64_t d_group_off, v4f (&acc)[MS][NS],
                                           int m_base, int n_base, int ms_only = -1) {
    const int lane = threadIdx.x & 63, lg = lane >> 4;
    if constexpr (NATURAL_COLS) {
        if constexpr (NS == 4) {
            if (p.d_dtype == 0 && !p.accumulate && p.d_vec_ok && n_base + 64 <= p.n) {
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    if (ms_only < 0 || ms == ms_only)
                        store_rows_full_line_natural<MS, INTERLEAVED_ROWS>(p, t, d_group_off, acc[ms], ms, m_base, n_base);
                return;
            }
        }
        const bool full = n_base + NS * 16 <= p.n;
        if constexpr (BATCH_RMW && MS % 4 == 0) {
            if (p.d_dtype != 0 && p.accumulate && full && p.d_vec_ok && ms_only < 0 && t.m_end > t.m_begin) {
                reduce_add_tile_fp32<MS, NS, INTERLEAVED_ROWS>(p, t, d_group_off, acc, m_base,
                                                               [&](int ns) { return n_base + ns * 16 + lg * 4; });
                return;
            }
        }
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms) {
            if (ms_only >= 0 && ms != ms_only)
                continue;
            const int row = INTERLEAVED_ROWS ? m_base + (lane & 15) * MS + ms : m_base + ms * 16 + (lane & 15);
            const bool compute_row = row >= t.m_begin && row < t.m_end;
            const bool zero_row = row >= t.zero_from && row < t.zero_to;
            if (!compute_row && !zero_row)
                continue;
            if (p.d_dtype == 0 && full && p.d_vec_ok) {
                // BF16: a lane's 4 consecutive columns of a subtile = one 8-byte store (the permuted order's 16-byte stores
                // need 8 consecutive columns per lane, which the natural B row order does not give)
                uint16_t* drow = reinterpret_cast<uint16_t*>(p.d) + d_group_off + static_cast<int64_t>(row) * p.d_sm;
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    uint2* dst = reinterpret_cast<uint2*>(drow + d_col(p, n_base + ns * 16 + lg * 4));
                    uint32_t w0 = zero_row ? 0u : pack_bf16(acc[ms][ns][0], acc[ms][ns][1]);
                    uint32_t w1 = zero_row ? 0u : pack_bf16(acc[ms][ns][2], acc[ms][ns][3]);
                    if (p.accumulate && !zero_row) {
                        const uint2 old = *dst;
                        w0 = pack_bf16(bf16_lo(old.x) + bf16_lo(w0), bf16_hi(old.x) + bf16_hi(w0));
                        w1 = pack_bf16(bf16_lo(old.y) + bf16_lo(w1), bf16_hi(old.y) + bf16_hi(w1));
                    }
                    *dst = make_uint2(w0, w1);
                }
                continue;
            }
            if (p.d_dtype != 0 && full && p.d_vec_ok) {
                float* drow = reinterpret_cast<float*>(p.d) + d_group_off + static_cast<int64_t>(row) * p.d_sm;
                v4f old[NS];
                if (p.accumulate && !zero_row) {
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        old[ns] = *reinterpret_cast<const v4f*>(drow + d_col(p, n_base + ns * 16 + lg * 4));
                }
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    v4f v = zero_row ? v4f{0.f, 0.f, 0.f, 0.f} : acc[ms][ns];
                    if (p.accumulate && !zero_row) v += old[ns];
                    *reinterpret_cast<v4f*>(drow + d_col(p, n_base + ns * 16 + lg * 4)) = v;
                }
                continue;
            }
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int col = n_base + ns * 16 + lg * 4 + r;
                    if (col >= p.n)
                        continue;
                    const int64_t off = d_group_off + static_cast<int64_t>(row) * p.d_sm + d_col(p, col);
                    if (p.d_dtype == 0) {
                        uint16_t* d16 = reinterpret_cast<uint16_t*>(p.d);
                        float v = zero_row ? 0.f : round_bf16(acc[ms][ns][r]);
                        if (p.accumulate && !zero_row)
                            v = v + bf16_lo(static_cast<uint32_t>(d16[off]));
                        d16[off] = static_cast<uint16_t>(pack_bf16(v, 0.f) & 0xffffu);
                    } else {
                        float* d32 = reinterpret_cast<float*>(p.d);
                        const float v = zero_row ? 0.f : acc[ms][ns][r];
                        d32[off] = (p.accumulate && !zero_row) ? d32[off] + v : v;
                    }
                }
        }
        return;
    }
    const int n_lane = n_base + lg * 8;                     // + (ns >> 1) * 32 + (ns & 1) * 4 + r
    const bool full_n = (n_lane + (NS / 2 - 1) * 32 + 8 <= p.n) && (NS % 2 == 0);

    // Full-line stores (BF16, 64-column wave tile entirely inside N, no accumulation): the output tail of a CU is bound by
    // the NUMBER of store requests (one per 64-byte run: ~5 cycles each), not by bytes.  Lanes r and r + 8 of a 16-lane row
    // swap halves through DPP (row_ror:8) so that one store instruction writes 8 rows x 128 contiguous bytes instead of
    // 16 rows x 64: half the requests for the same bytes.
    if constexpr (NS == 4) {
        if (p.d_dtype == 0 && !p.accumulate && p.d_vec_ok && n_base + 64 <= p.n) {
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                if (ms_only < 0 || ms == ms_only)
                    store_rows_full_line<MS, INTERLEAVED_ROWS>(p, t, d_group_off, acc[ms], ms, m_base, n_base);
            return;
        }
    }
    if constexpr (!NT_STORE && MS % 2 == 0 && NS % 2 == 0) {
        if (p.d_dtype == 0 && p.accumulate && n_base + NS * 16 <= p.n && p.d_vec_ok && ms_only < 0 && t.m_end > t.m_begin) {
            reduce_add_tile_bf16<MS, NS, INTERLEAVED_ROWS>(p, t, d_group_off, acc, m_base, n_lane);
            return;
        }
    }
    if constexpr (BATCH_RMW && MS % 4 == 0 && NS % 2 == 0) {
        if (p.d_dtype != 0 && p.accumulate && n_base + NS * 16 <= p.n && p.d_vec_ok && ms_only < 0 && t.m_end > t.m_begin) {
            reduce_add_tile_fp32<MS, NS, INTERLEAVED_ROWS>(p, t, d_group_off, acc, m_base,
                                                           [&](int ns) { return n_lane + (ns >> 1) * 32 + (ns & 1) * 4; });
            return;
        }
    }
    #pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        if (ms_only >= 0 && ms != ms_only)
            continue;
        const int row = INTERLEAVED_ROWS ? m_base + (lane & 15) * MS + ms : m_base + ms * 16 + (lane & 15);
        const bool compute_row = row >= t.m_begin && row < t.m_end;
        const bool zero_row = row >= t.zero_from && row < t.zero_to;
        if (!compute_row && !zero_row)
            continue;
        if (p.d_dtype == 0) {
            uint16_t* drow = reinterpret_cast<uint16_t*>(p.d) + d_group_off + static_cast<int64_t>(row) * p.d_sm;
            if (full_n && p.d_vec_ok) {
                #pragma unroll
                for (int h = 0; h < NS / 2; ++h) {
                    uint32_t w[4];
                    #pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const v4f v = acc[ms][2 * h + j];
                        w[2 * j] = zero_row ? 0u : pack_bf16(v[0], v[1]);
                        w[2 * j + 1] = zero_row ? 0u : pack_bf16(v[2], v[3]);
                    }
                    uint4* dst = reinterpret_cast<uint4*>(drow + d_col(p, n_lane + h * 32));
                    if (p.accumulate && !zero_row) {
                        const uint4 old = *dst;
                        const uint32_t o[4] = {old.x, old.y, old.z, old.w};
                        #pragma unroll
                        for (int j = 0; j < 4; ++j)
                            w[j] = pack_bf16(bf16_lo(o[j]) + bf16_lo(w[j]), bf16_hi(o[j]) + bf16_hi(w[j]));
                    }
                    if constexpr (NT_STORE) {
                        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
                        __builtin_nontemporal_store(u4{w[0], w[1], w[2], w[3]}, reinterpret_cast<u4*>(dst));
                    } else {
                        *dst = make_uint4(w[0], w[1], w[2], w[3]);
                    }
                }
            } else {
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    #pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int col = n_lane + (ns >> 1) * 32 + (ns & 1) * 4 + r;
                        if (col < p.n) {
                            const int dc = d_col(p, col);
                            float v = zero_row ? 0.f : round_bf16(acc[ms][ns][r]);
                            if (p.accumulate && !zero_row)
                                v = v + bf16_lo(static_cast<uint32_t>(drow[dc]));
                            drow[dc] = static_cast<uint16_t>(pack_bf16(v, 0.f) & 0xffffu);
                        }
                    }
            }
        } else {
            float* drow = reinterpret_cast<float*>(p.d) + d_group_off + static_cast<int64_t>(row) * p.d_sm;
            if (full_n && p.d_vec_ok && p.accumulate && !zero_row) {
                // reduce-add: all of the row's loads first, then the adds and stores -- one memory round trip per row
                // instead of one per 16-byte vector (hipcc cannot hoist a load over the previous vector's store itself)
                v4f old[NS];
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    old[ns] = *reinterpret_cast<const v4f*>(drow + d_col(p, n_lane + (ns >> 1) * 32 + (ns & 1) * 4));
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    *reinterpret_cast<v4f*>(drow + d_col(p, n_lane + (ns >> 1) * 32 + (ns & 1) * 4)) = acc[ms][ns] + old[ns];
                continue;
            }
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                v4f v = acc[ms][ns];
                if (zero_row) v = v4f{0.f, 0.f, 0.f, 0.f};
                const int col = n_lane + (ns >> 1) * 32 + (ns & 1) * 4;
                if (full_n && p.d_vec_ok) {
                    v4f* dst = reinterpret_cast<v4f*>(drow + d_col(p, col));
                    if (p.accumulate && !zero_row) v += *dst;
                    *dst = v;
                } else {
                    #pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < p.n) {
                            const int dc = d_col(p, col + r);
                            drow[dc] = (p.accumulate && !zero_row) ? drow[dc] + v[r] : v[r];
                        }
                }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Fast path, hand-scheduled: same tiles / LDS image / LDS-DMA staging as dg_fp8_gemm_fast_kernel, but the MFMA +
// FP32-promotion stream is written as one inline-asm statement per 16x16x128 step:
//     v_mfma  part[i & 3] = Bfrag[ns] x Afrag[ms]            (zero C operand: one 128-K scale block)
//     4 x v_fmac  acc[step i-3] += scale[step i-3] * part[(i-3) & 3]
// i.e. the promotion of step i-3 rides in the shadow of MFMA i (three MFMAs = 96 matrix-pipe cycles separate a
// result from its first VALU read, which also satisfies the 12-wait-state XDL-write -> VALU-read rule without
// s_nop padding).  The ring runs across K blocks: the last three steps of block kb are promoted during the first
// three MFMAs of block kb+1 with block kb's scales.  hipcc, left alone, emits mfma / s_nop 11 / fma per step.
// ---------------------------------------------------------------------------------------------------------------
// (Negative result, round 3: the four promotion FMAs as TWO v_pk_fma_f32 -- half the VALU issue slots -- run C2 at 110 us instead of
// 91.5 us, C3 36.2 instead of 30.0: packed FP32 FMAs issue at half rate behind a matrix instruction on this part.  -DDG_PK_FMA builds it.)
// Cache policy of the in-kernel K-split exchange (GemmParams::sk_exchange): 17 = sc0 | sc1 on the partial tile's stores and loads (written
// through / read past the per-XCD L2) with relaxed flag accesses; 0 = ordinary accesses with release / acquire at agent scope (a writeback
// and an invalidate of the WHOLE L2 per workgroup: measured 151 us against 142 for C4 with the reduction kernel).
#ifndef DG_SK_XCHG_AUX
#define DG_SK_XCHG_AUX 17
#endif
#define DG_SK_XCHG_ORDER_REL (DG_SK_XCHG_AUX == 0 ? __ATOMIC_RELEASE : __ATOMIC_RELAXED)
#define DG_SK_XCHG_ORDER_ACQ (DG_SK_XCHG_AUX == 0 ? __ATOMIC_ACQUIRE : __ATOMIC_RELAXED)
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mfma_promote_step(v4f& part_new, const v8i& rows_operand, const v8i& cols_operand,
                                                  float (&c)[4], float scale, const v4f& part_old) {
#ifndef DG_PK_FMA
    asm volatile(
        "v_mfma_f32_16x16x128_f8f6f4 %0, %5, %6, 0\n\t"
        "v_fmac_f32 %1, %7, %8\n\t"
        "v_fmac_f32 %2, %7, %9\n\t"
        "v_fmac_f32 %3, %7, %10\n\t"
        "v_fmac_f32 %4, %7, %11"
        : "=&v"(part_new), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
        : "v"(rows_operand), "v"(cols_operand), "v"(scale), "v"(part_old[0]), "v"(part_old[1]), "v"(part_old[2]),
          "v"(part_old[3])
        : "memory");
#else
    v2f c01 = {c[0], c[1]}, c23 = {c[2], c[3]};
    const v2f s2 = {scale, scale}, p01 = {part_old[0], part_old[1]}, p23 = {part_old[2], part_old[3]};
    asm volatile(
        "v_mfma_f32_16x16x128_f8f6f4 %0, %3, %4, 0\n\t"
        "v_pk_fma_f32 %1, %5, %6, %1\n\t"
        "v_pk_fma_f32 %2, %5, %7, %2"
        : "=&v"(part_new), "+v"(c01), "+v"(c23)
        : "v"(rows_operand), "v"(cols_operand), "v"(s2), "v"(p01), "v"(p23)
        : "memory");
    c[0] = c01[0]; c[1] = c01[1]; c[2] = c23[0]; c[3] = c23[1];
#endif
}

// Forces `x` to be materialised in a VGPR at this point of the instruction stream (scheduling fence for one value).
__device__ __forceinline__ void pin_vgpr(float& x) { asm volatile("" : "+v"(x)); }

__device__ __forceinline__ void promote_only(float (&c)[4], float scale, const v4f& part_old) {
    // Drain step.  s_nop 3 keeps >= 12 wait states between the last MFMA and the first read of its result even if
    // every intervening instruction issues back to back.
#ifndef DG_PK_FMA
    asm volatile(
        "s_nop 3\n\t"
        "v_fmac_f32 %0, %4, %5\n\t"
        "v_fmac_f32 %1, %4, %6\n\t"
        "v_fmac_f32 %2, %4, %7\n\t"
        "v_fmac_f32 %3, %4, %8"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
        : "v"(scale), "v"(part_old[0]), "v"(part_old[1]), "v"(part_old[2]), "v"(part_old[3])
        : "memory");
#else
    v2f c01 = {c[0], c[1]}, c23 = {c[2], c[3]};
    const v2f s2 = {scale, scale}, p01 = {part_old[0], part_old[1]}, p23 = {part_old[2], part_old[3]};
    asm volatile(
        "s_nop 3\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0\n\t"
        "v_pk_fma_f32 %1, %2, %4, %1"
        : "+v"(c01), "+v"(c23)
        : "v"(s2), "v"(p01), "v"(p23)
        : "memory");
    c[0] = c01[0]; c[1] = c01[1]; c[2] = c23[0]; c[3] = c23[1];
#endif
}

// SPREAD: LDS-DMA piece placement: 0 = the whole next stage at the head of the K block, n = one piece every n MFMA steps.
template <int BM, int BN, int WAVES_M, int WAVES_N, int SPREAD>
__device__ __forceinline__ void pipe_kernel_body(const GemmParams& p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16;
    constexpr int TOTAL = MS * NS, DEPTH = 3;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_UNITS = BM / 8, B_UNITS = BN / 8;
    constexpr int A_ITERS = (A_UNITS + NW - 1) / NW, B_ITERS = (B_UNITS + NW - 1) / NW;
    static_assert(WM % 16 == 0 && WN % 16 == 0 && NS % 2 == 0, "wave tile must be a multiple of 16 x 32");
    static_assert(WN <= 128 && 128 % WN == 0, "one SFB value per wave");
    static_assert(TOTAL >= DEPTH + 1, "pipeline ring needs at least 4 steps per K block");
    static_assert(SPREAD == 0 || SPREAD * (A_ITERS + B_ITERS) <= TOTAL, "not enough steps to spread the LDS-DMA pieces");
    static_assert((NW * 8) % 16 == 0 && ((NW * 8) % WN == 0 || WN % (NW * 8) == 0),
                  "the row permutation of a B piece must be lane-independent: perm(p0 + slab) = perm(p0) + perm(slab)");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * STAGE_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int num_kb = p.k / 128;
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);

    // Per-lane byte offsets of the LDS-DMA source pattern inside the first 8*NW-row slab of a tile; the slab index and
    // the K block go into the (wave-uniform) soffset of the buffer instruction.
    const int a_voff = (wave * 8 + piece_row) * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
    const long long t_entry = p.dbg != nullptr ? DG_STAMP_CLOCK() : 0;
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
            // Buffer descriptors bound each tile to its valid rows: out-of-range lanes of an edge tile fetch nothing
            // (those rows / columns are never stored), so no per-lane clamping is needed.
            const uint8_t* a_base = uniform_pointer(p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm);
            const uint8_t* b_base = uniform_pointer(p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn);
            const int a_rows = uniform_int(imin(t.m_end - t.m0, BM)), b_rows = uniform_int(imin(p.n - t.n0, BN));
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base), 0,
                                                                  (a_rows - 1) * lda + p.k, 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base), 0,
                                                                  (b_rows - 1) * ldb + p.k, 0x00020000);
            const float* sfa_group = uniform_pointer(p.sfa + ad_group * p.sfa_sg);
            const int sfa_rows = (p.gemm_type == kMasked) ? p.m : p.m;
            const int sfa_extent = (static_cast<int>(p.sfa_sm) * (sfa_rows - 1) + static_cast<int>(p.sfa_sk) * (num_kb - 1) + 1) * 4;
            const auto sfa_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(sfa_group), 0, sfa_extent, 0x00020000);
            const int sfa_voff = (t.m0 + wm * WM + (lane & 15)) * static_cast<int>(p.sfa_sm) * 4;
            const int sfa_ms_stride = 16 * static_cast<int>(p.sfa_sm) * 4, sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4;
            const float* sfb_wave = p.sfb + static_cast<int64_t>(t.group) * p.sfb_sg +
                                    static_cast<int64_t>((t.n0 + wn * WN) / 128) * p.sfb_sn;

            // One LDS-DMA piece (1 KiB = 8 rows x 128 B) of the stage; pieces 0..A_ITERS-1 belong to A, the rest to B.
            auto issue_piece = [&](int stage, int kb, int q) {
                uint8_t* stage_base = lds + stage * STAGE_BYTES;
                if (q < A_ITERS) {
                    const int unit = wave + NW * q;
                    if (A_UNITS % NW == 0 || unit < A_UNITS)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            a_rsrc, (__attribute__((address_space(3))) void*)(stage_base + unit * 1024), 16, a_voff,
                            q * (NW * 8) * lda + kb * 128, 0, 0);
                } else {
                    const int j = q - A_ITERS, unit = wave + NW * j;
                    if (B_UNITS % NW == 0 || unit < B_UNITS)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            b_rsrc, (__attribute__((address_space(3))) void*)(stage_base + A_BYTES + unit * 1024), 16,
                            b_voff, b_row_perm<WN>(j * (NW * 8)) * ldb + kb * 128, 0, 0);
                }
            };
            auto issue_stage = [&](int stage, int kb) {
                #pragma unroll
                for (int q = 0; q < A_ITERS + B_ITERS; ++q)
                    issue_piece(stage, kb, q);
            };
            auto load_sfa = [&](int ms, int kb) -> float {
                return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    sfa_rsrc, sfa_voff, ms * sfa_ms_stride + kb * sfa_kb_stride, 0));
            };

            // scale[ms]: SFA(kb) * SFB(kb), formed at the top of block kb from values fetched one block earlier.
            float scale[MS], sa_nxt[MS], scale_tail = 0.f;
            float sb_cur, sb_nxt = 0.f;
            v4f part[DEPTH + 1];
            #pragma unroll
            for (int i = 0; i <= DEPTH; ++i)
                part[i] = v4f{0.f, 0.f, 0.f, 0.f};

            issue_stage(0, 0);
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                sa_nxt[ms] = load_sfa(ms, 0);
            sb_cur = sfb_wave[0];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            v8i bf[NS], af[2];
            if (p.dbg != nullptr) t_loop0 = DG_STAMP_CLOCK();
            for (int kb = 0; kb < num_kb; ++kb) {
                const int cur = kb & 1;
                const bool has_next = kb + 1 < num_kb;
                // Form the block's scales BEFORE any new LDS-DMA is in flight: hipcc waits vmcnt(0) at the first use of an
                // ordinary load result, which would drain the prefetch if it happened after the DMA issue.
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    scale[ms] = sa_nxt[ms] * sb_cur;
                    pin_vgpr(scale[ms]);
                }
                // Next block's SFA (unconditional: past the last K block the buffer descriptor bounds the access).
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    sa_nxt[ms] = load_sfa(ms, kb + 1);
                if (has_next) {
                    if constexpr (SPREAD == 0)
                        issue_stage(cur ^ 1, kb + 1);
                    sb_nxt = sfb_wave[static_cast<int64_t>(kb + 1) * p.sfb_sk];
                }

                const uint8_t* a_tile = lds + cur * STAGE_BYTES + (wm * WM) * 128;
                const uint8_t* b_tile = lds + cur * STAGE_BYTES + A_BYTES + (wn * WN) * 128;
                bf[0] = load_fragment(b_tile, frag_off);
                af[0] = load_fragment(a_tile, frag_off);

                #pragma unroll
                for (int i = 0; i < TOTAL; ++i) {
                    const int ms = i / NS, ns = i % NS;
                    const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;     // step being promoted
                    const int jms = j / NS, jns = j % NS;
                    const float jscale = (i >= DEPTH) ? scale[jms] : scale_tail;     // i < DEPTH: previous block's tail
                    // Fragment reads ride one step ahead of their first use: B subtile ns+1 during the first M-subtile,
                    // A subtile ms+1 at the head of subtile ms.
                    if (ms == 0 && ns + 1 < NS)
                        bf[ns + 1] = load_fragment(b_tile + (ns + 1) * 2048, frag_off);
                    if (ns == 0 && ms + 1 < MS)
                        af[(ms + 1) & 1] = load_fragment(a_tile + (ms + 1) * 2048, frag_off);
                    mfma_promote_step(part[i & DEPTH], bf[ns], af[ms & 1], acc[jms][jns], jscale, part[(i + 1) & DEPTH]);
                    if constexpr (SPREAD > 0) {
                        // spread the next stage's LDS-DMA pieces over the first steps, one per SPREAD MFMAs
                        if (i % SPREAD == SPREAD - 1 && i / SPREAD < A_ITERS + B_ITERS && has_next)
                            issue_piece(cur ^ 1, kb + 1, i / SPREAD);
                    }
                }
                static_assert((TOTAL - DEPTH) / NS == MS - 1, "the ring tail must lie within the last M-subtile");
                scale_tail = scale[MS - 1];
                sb_cur = sb_nxt;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (p.dbg != nullptr) t_loop1 = DG_STAMP_CLOCK();
            // drain the ring: steps TOTAL-3 .. TOTAL-1 of the last K block
            #pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                const int j = TOTAL - DEPTH + i;
                promote_only(acc[j / NS][j % NS], scale_tail, part[(TOTAL + i + 1) & DEPTH]);
            }
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
            dbg_stamp(p, NW, 3, DG_STAMP_CLOCK());
        }
    }
}

// The body lives in a __device__ function: it uses gfx950-only types (buffer resources) that the host pass of hipcc
// cannot name, and a __global__ function whose body the host pass rejects gets no launch stub.
template <int BM, int BN, int WAVES_M, int WAVES_N, int SPREAD>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_pipe_kernel(const GemmParams p) {
    pipe_kernel_body<BM, BN, WAVES_M, WAVES_N, SPREAD>(p);
}

// ---------------------------------------------------------------------------------------------------------------
// Per-column-SFB form of the pipe kernel (recipe (1, 1, 128): one scale per row of A AND per row of B for every 128-K
// block; the reference's "1D1D" kernel, impls/sm90_fp8_gemm_1d1d.cuh:279-311, used for FP32-accumulating wgrad GEMMs).
//
//   final[m][n] += (sfa[m][kb] * sfb[n][kb]) * partial_kb[m][n]
//
// Differences from the per-128-column form above:
//   * the scale product is no longer one value per (lane, M-subtile) but one per accumulator element: a step carries four
//     multiplies (sfb x sfa) and four FMAs.  (First version: two packed multiplies + two packed FMAs, the same VALU issue
//     count as the per-128 form -- but packed FP32 ops beside MFMAs are slower than two scalar ops each on this part:
//     4.8 k cycles per K block against 4.0 k, kept as pipe_pcpk_256x256.)
//   * both scale vectors of a K block travel with the block's stage as two extra 1 KiB LDS-DMA pieces (256 FP32 row
//     scales of A, 256 of B: the MN-major SF layouts make each a contiguous run), so the loop holds no register-landing
//     global loads at all; every wave picks its 8 + 16 values out of LDS at the top of the block.
// ---------------------------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// The same step with single-rate VALU: 4 products + 4 FMAs.  (v_pk_*_f32 beside MFMAs costs more than two scalar ops
// each on this part -- MI355X_MICROARCH.md, "price of one filler beside MFMAs" -- so the packed form is the slower one.)
__device__ __forceinline__ void mfma_promote_step_pc_scalar(v4f& part_new, const v8i& rows_operand, const v8i& cols_operand,
                                                            float (&c)[4], const v4f& sb, float sa, const v4f& part_old) {
    float t0, t1, t2, t3;
    asm volatile(
        "v_mfma_f32_16x16x128_f8f6f4 %0, %9, %10, 0\n\t"
        "v_mul_f32 %5, %11, %15\n\t"
        "v_mul_f32 %6, %12, %15\n\t"
        "v_mul_f32 %7, %13, %15\n\t"
        "v_mul_f32 %8, %14, %15\n\t"
        "v_fmac_f32 %1, %5, %16\n\t"
        "v_fmac_f32 %2, %6, %17\n\t"
        "v_fmac_f32 %3, %7, %18\n\t"
        "v_fmac_f32 %4, %8, %19"
        : "=&v"(part_new), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(rows_operand), "v"(cols_operand), "v"(sb[0]), "v"(sb[1]), "v"(sb[2]), "v"(sb[3]), "v"(sa), "v"(part_old[0]),
          "v"(part_old[1]), "v"(part_old[2]), "v"(part_old[3])
        : "memory");
}

__device__ __forceinline__ void promote_only_pc_scalar(float (&c)[4], const v4f& sb, float sa, const v4f& part_old) {
    float t0, t1, t2, t3;
    asm volatile(
        "s_nop 3\n\t"
        "v_mul_f32 %4, %8, %12\n\t"
        "v_mul_f32 %5, %9, %12\n\t"
        "v_mul_f32 %6, %10, %12\n\t"
        "v_mul_f32 %7, %11, %12\n\t"
        "v_fmac_f32 %0, %4, %13\n\t"
        "v_fmac_f32 %1, %5, %14\n\t"
        "v_fmac_f32 %2, %6, %15\n\t"
        "v_fmac_f32 %3, %7, %16"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(sb[0]), "v"(sb[1]), "v"(sb[2]), "v"(sb[3]), "v"(sa), "v"(part_old[0]), "v"(part_old[1]), "v"(part_old[2]),
          "v"(part_old[3])
        : "memory");
}

// gfx9 s_waitcnt immediate: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; expcnt left at "no wait".
constexpr int waitcnt_imm(int vmcnt, int lgkmcnt) {
    return (vmcnt & 0xf) | (0x7 << 4) | ((lgkmcnt & 0xf) << 8) | (((vmcnt >> 4) & 0x3) << 14);
}

// Fragment of an MN-major operand tile ([128 k][256 mn bytes] in LDS, 16-byte chunk c of row k stored at chunk
// c ^ f(k), f(k) = (k & 7) | ((k >> 4 & 1) << 3)) through the hardware transpose read: in a 16-lane group lanes 2r / 2r+1
// address the two halves of row r of an [8 k][16 mn] block and lane c receives byte c of every row (probed:
// tools/ubench/tr_b8_probe.hip).  Four reads give the lane the same 32 K slots a K-major fragment holds:
// k = 16g .. 16g+15 and 64+16g .. 64+16g+15 (g = lane >> 4).  lane_base = (16g + (i >> 1)) * 256 + (i & 1) * 8 with
// i = lane & 15; chunk_off = (c ^ ((i >> 1) | ((g & 1) << 3))) << 4 for the fragment's chunk c.
// The four 8-byte results land in separate register pairs; they are put together into the MFMA operand only AFTER the
// wait for them (assemble_fragment_tr): building the 8-register tuple right after the asm would let hipcc copy registers the
// loads have not written yet.
typedef int v2i_t __attribute__((ext_vector_type(2)));
struct FragTr { v2i_t q0, q1, q2, q3; };

__device__ __forceinline__ FragTr load_fragment_tr(const uint8_t* tile, int lane_base, int chunk_off) {
    FragTr f;
    const int addr = static_cast<int>(reinterpret_cast<uintptr_t>(tile)) + lane_base + chunk_off;
    asm volatile(
        "ds_read_b64_tr_b8 %0, %4\n\t"
        "ds_read_b64_tr_b8 %1, %4 offset:2048\n\t"
        "ds_read_b64_tr_b8 %2, %4 offset:16384\n\t"
        "ds_read_b64_tr_b8 %3, %4 offset:18432"
        : "=&v"(f.q0), "=&v"(f.q1), "=&v"(f.q2), "=&v"(f.q3)
        : "v"(addr)
        : "memory");
    return f;
}

__device__ __forceinline__ v8i assemble_fragment_tr(FragTr& f) {
    asm volatile("" : "+v"(f.q0), "+v"(f.q1), "+v"(f.q2), "+v"(f.q3));      // the values exist from here on
    return v8i{f.q0[0], f.q0[1], f.q1[0], f.q1[1], f.q2[0], f.q2[1], f.q3[0], f.q3[1]};
}

// MN = true: both FP8 operands are MN-major ([K][M] and [K][N], unit stride along m / n, row pitch a_sk / b_sk): the
// operand form of the K-grouped TN GEMM.  LDS-DMA pieces are 4 k-rows x 256 bytes, fragments come through
// load_fragment_tr, A and B rows keep their natural order (=> FP32 output only gets vector stores).
template <int BM, int BN, int WAVES_M, int WAVES_N, int SPREAD, bool MN = false>
__device__ __forceinline__ void pipe_pc_kernel_body(const GemmParams& p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16;
    constexpr int TOTAL = MS * NS, DEPTH = 3;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_UNITS = BM / 8, B_UNITS = BN / 8;
    constexpr int A_ITERS = A_UNITS / NW, B_ITERS = B_UNITS / NW;
    constexpr int SC_BASE = 2 * STAGE_BYTES, SC_STAGE = 2048;      // per stage: 256 row scales of A, 256 of B
    // (BM = 192, round 5: 576-row wgrads are three exact tile rows instead of 2.25 of three; the A scale piece still fetches 256 values --
    //  the 64 beyond the tile are the next rows' or lie outside the descriptor -- and nothing reads them)
    static_assert((BM == 256 || (BM == 192 && !MN)) && BN == 256, "one 1 KiB scale piece per operand per K block");
    static_assert(MS % 2 == 0 && NS % 2 == 0 && NW >= 2 && A_UNITS % NW == 0 && B_UNITS % NW == 0, "tile shape");
    static_assert(TOTAL >= DEPTH + 1 && (TOTAL - DEPTH) / NS == MS - 1, "the ring tail must lie within the last M-subtile");
    static_assert(SPREAD * (A_ITERS + B_ITERS) <= TOTAL, "not enough steps to spread the LDS-DMA pieces");

    constexpr int TOUCH_BASE = SC_BASE + 2 * SC_STAGE;             // 256 bytes nobody reads: where the touch-ahead loads of C land
    __shared__ __attribute__((aligned(1024))) uint8_t lds[TOUCH_BASE + 256];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    int num_kb = p.k / 128;                         // K-grouped launch: set per tile from the group's K extent
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    int lda = static_cast<int>(MN ? p.a_sk : p.a_sm), ldb = static_cast<int>(MN ? p.b_sk : p.b_sn);
    // MN: lane l of piece u carries k-row 4u + (l >> 4), source chunk (l & 15) ^ f(k); u = wave + 8q makes f lane-constant
    const int mn_swz = ((4 * (wave & 1) + (lane >> 4)) & 7) | (((wave >> 2) & 1) << 3);
    const int mn_col = (((lane & 15) ^ mn_swz) << 4);
    int a_voff = MN ? (lane >> 4) * lda + mn_col : (wave * 8 + piece_row) * lda + src_chunk * 16;
    int b_voff = MN ? (lane >> 4) * ldb + mn_col : b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
    const int tr_lane_base = (16 * (lane >> 4) + ((lane & 15) >> 1)) * 256 + (lane & 1) * 8;
    const int tr_swz = ((lane & 15) >> 1) | (((lane >> 4) & 1) << 3);
    const int sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kb_stride = static_cast<int>(p.sfb_sk) * 4;
    // where this lane's scales sit inside a stage's scale block
    const int sa_lds_off = (wm * WM + (lane & 15)) * 4;                            // + ms * 64
    const int sb_lds_off = MN ? 1024 + (wn * WN + (lane >> 4) * 4) * 4               // + ns * 64 (natural column order)
                              : 1024 + (wn * WN + (lane >> 4) * 8) * 4;              // + (ns >> 1) * 128 + (ns & 1) * 16
    const long long t_entry = p.dbg != nullptr ? DG_STAMP_CLOCK() : 0;
    long long t_loop0 = 0, t_loop1 = 0;

    MaskedWalk walk;
    const int num_launched = gridDim.x;
    for (int tile_id = blockIdx.x;; tile_id += num_launched) {
        const Tile t = get_tile<BM, BN>(p, tile_id, walk);
        if (!t.valid)
            break;
        const int64_t ad_group = (p.gemm_type == kMasked) ? t.group : 0;
        int64_t kg_a_off = 0, kg_b_off = 0, kg_sf_blocks = 0;      // K-grouped launch: where the group's operands start
        int k_extent = p.k;
        if (p.gemm_type == kKGrouped) {
            int k_begin, k_stop;
            if (p.kg_psum && p.m_alignment != 0 && p.m_alignment != 128) {
                // K alignment A != 128 (round 6; the reference's SM100 sweep: 32 / 160 / 192 / 224, tests/generators.py:192-194,
                // scheduler/gemm.cuh:74-85): a group starts at the previous end rounded up to A, its scale blocks count from ITS start (compact
                // rows: ceil(extent / 128) per non-empty group, scheduler/gemm.cuh:247), its last block is partial -- the k-rows at and beyond
                // the group's end lie outside the operand descriptors below and arrive in the LDS as zeros (MN-major operands only: host check)
                int prev_end = 0, rows = 0;
                for (int g = 0; g < t.group; ++g) {
                    const int end = p.layout[g], start = (prev_end + p.m_alignment - 1) / p.m_alignment * p.m_alignment;
                    if (end > start)
                        rows += (end - start + 127) >> 7;
                    prev_end = end;
                }
                k_begin = __builtin_amdgcn_readfirstlane((prev_end + p.m_alignment - 1) / p.m_alignment * p.m_alignment);
                k_stop = __builtin_amdgcn_readfirstlane(imin(p.layout[t.group], p.k));
                kg_sf_blocks = __builtin_amdgcn_readfirstlane(rows);
            } else if (p.kg_psum) {
                const int prev_end = t.group > 0 ? p.layout[t.group - 1] : 0;
                k_begin = __builtin_amdgcn_readfirstlane((prev_end + 127) & ~127);
                k_stop = __builtin_amdgcn_readfirstlane(imin((p.layout[t.group] + 127) & ~127, p.k));
            } else {
                k_begin = p.kg_prefix[t.group];
                k_stop = p.kg_prefix[t.group + 1];
            }
            k_extent = k_stop - k_begin;
            if (k_extent <= 0)
                continue;                                         // empty group: D[g] stays as it is
            num_kb = (k_extent + 127) / 128;                  // (a partial last block only in the psum form with a K alignment != 128)
            if (!(p.kg_psum && p.m_alignment != 0 && p.m_alignment != 128))
                kg_sf_blocks = k_begin / 128;
            if constexpr (MN) {
                kg_a_off = static_cast<int64_t>(k_begin) * lda;
                kg_b_off = static_cast<int64_t>(k_begin) * ldb;
            } else if (p.kg_blocks) {
                lda = ldb = k_extent;
                a_voff = (wave * 8 + piece_row) * lda + src_chunk * 16;
                b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
                kg_a_off = static_cast<int64_t>(k_begin) * p.m;
                kg_b_off = static_cast<int64_t>(k_begin) * p.n;
            } else {
                kg_a_off = kg_b_off = k_begin;
            }
        }
        const int64_t bs_group = (p.gemm_type == kKGrouped) ? 0 : t.group;          // group index into B / SFB

        float accs[MS][NS][4];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                #pragma unroll
                for (int r = 0; r < 4; ++r)
                    accs[ms][ns][r] = 0.f;

        if (t.m_end > t.m0) {
            // (every descriptor input through readfirstlane: tile coordinates are wave-uniform in fact, not provably -- see uniform_int)
            const uint8_t* a_base = uniform_pointer(p.a + ad_group * p.a_sg + kg_a_off + static_cast<int64_t>(t.m0) * (MN ? 1 : lda));
            const uint8_t* b_base = uniform_pointer(p.b + bs_group * p.b_sg + kg_b_off + static_cast<int64_t>(t.n0) * (MN ? 1 : ldb));
            const int a_rows = imin(t.m_end - t.m0, BM), b_rows = imin(p.n - t.n0, BN);
            // MN: the descriptor ends with the last k-row's valid bytes; a lane past M (N) inside an earlier row reads the
            // next row's head -- finite bytes that only reach rows / columns which are never stored
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(a_base), 0, uniform_int(MN ? (k_extent - 1) * lda + (p.m - t.m0) : (a_rows - 1) * lda + k_extent), 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint8_t*>(b_base), 0, uniform_int(MN ? (k_extent - 1) * ldb + (p.n - t.n0) : (b_rows - 1) * ldb + k_extent), 0x00020000);
            // scale rows: MN-major (stride 1 along m / n), one K block = one contiguous run; lanes past the end of the
            // last run fall outside the descriptor and fetch zeros (rows / columns that are never stored)
            const float* sfa_tile = uniform_pointer(p.sfa + ad_group * p.sfa_sg + kg_sf_blocks * p.sfa_sk + t.m0);
            const float* sfb_tile = uniform_pointer(p.sfb + bs_group * p.sfb_sg + kg_sf_blocks * p.sfb_sk + t.n0);
            const auto sfa_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(sfa_tile), 0, uniform_int((num_kb - 1) * sfa_kb_stride + (p.m - t.m0) * 4), 0x00020000);
            const auto sfb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(sfb_tile), 0, uniform_int((num_kb - 1) * sfb_kb_stride + (p.n - t.n0) * 4), 0x00020000);

            auto issue_piece = [&](int stage, int kb, int q) {
                uint8_t* stage_base = lds + stage * STAGE_BYTES;
                if (q < A_ITERS) {
                    const int unit = wave + NW * q;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        a_rsrc, (__attribute__((address_space(3))) void*)(stage_base + unit * 1024), 16, a_voff,
                        MN ? (kb * 128 + 4 * unit) * lda : q * (NW * 8) * lda + kb * 128, 0, 0);
                } else {
                    const int j = q - A_ITERS, unit = wave + NW * j;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        b_rsrc, (__attribute__((address_space(3))) void*)(stage_base + A_BYTES + unit * 1024), 16,
                        b_voff, MN ? (kb * 128 + 4 * unit) * ldb : b_row_perm<WN>(j * (NW * 8)) * ldb + kb * 128, 0, 0);
                }
            };
            // waves 0 and 1 carry the block's two scale pieces (lane l: 4 consecutive FP32 scales)
            auto issue_scale_piece = [&](int stage, int kb) {
                uint8_t* sc = lds + SC_BASE + stage * SC_STAGE;
                if (wave == 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(sfa_rsrc, (__attribute__((address_space(3))) void*)sc, 16,
                                                             lane * 16, kb * sfa_kb_stride, 0, 0);
                else if (wave == 1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(sfb_rsrc, (__attribute__((address_space(3))) void*)(sc + 1024),
                                                             16, lane * 16, kb * sfb_kb_stride, 0, 0);
            };

            // Touch-ahead of C (round 4).  An accumulating call (D += A B^T: every wgrad and K-grouped call) ends with a read-modify-write
            // of the whole output tile -- 256 KiB per workgroup fetched cold from HBM and written back while nothing else runs (21.7 us
            // of the 134 us wgrad call).  The old values are known to be needed from the first cycle on: one dword of every line of the
            // tile is pulled through the L2 while the K loop runs, spread evenly over the K blocks (LDS-DMA into a dummy slot: no
            // register, no data dependency; the block's closing vmcnt(0) covers it), so that the epilogue's reads are served by the
            // Infinity Cache and HBM only has the writes left.  touch_units wave-instructions of 64 lines cover the tile.
            // NEGATIVE (same-box A/B, profiles/r04_probe/c_touch_ahead_negative.log): the epilogue shrinks 41.0 k -> 36.0 k ticks but the K loop
            // grows 4171 -> 4343 ticks per block (one more VMEM instruction under the block's closing vmcnt(0)): wgrad 142.5-144.0 -> 144.2 us,
            // K-grouped 1258 -> 1321 us; 64-byte granules: worse.  Compiled in only with -DDG_C_TOUCH.
#ifndef DG_C_TOUCH_BYTES
#define DG_C_TOUCH_BYTES 128
#endif
            const int d_elem = p.d_dtype == 0 ? 2 : 4;
            const int touch_per_row = BN * d_elem / DG_C_TOUCH_BYTES;                 // touches per tile row (a power of two)
            const int touch_shift = __builtin_ctz(touch_per_row);
            const int touch_units = p.accumulate && p.head_lr == 0 ? BM * touch_per_row / 64 : 0;
            const int d_rows = imin(t.m_end - t.m0, BM);
            const int64_t d_tile_off = ((p.gemm_type == kKGrouped ? t.group : ad_group) * p.d_sg + static_cast<int64_t>(t.m0) * p.d_sm + t.n0) * d_elem;
            const auto d_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                uniform_pointer(reinterpret_cast<uint8_t*>(p.d) + d_tile_off), 0,
                uniform_int(((d_rows - 1) * static_cast<int>(p.d_sm) + imin(p.n - t.n0, BN)) * d_elem), 0x00020000);
            [[maybe_unused]] int touch_next = 0, touch_credit = 0;
            [[maybe_unused]] auto touch_c = [&](int kb_total) {
                // Bresenham: touch_units instructions over kb_total K blocks; unit u belongs to wave u % NW
                touch_credit += touch_units;
                while (touch_credit >= kb_total && touch_next < touch_units) {
                    touch_credit -= kb_total;
                    if ((touch_next % NW) == wave) {
                        const int line = touch_next * 64 + lane;
                        const int row = line >> touch_shift, in_row = line & (touch_per_row - 1);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsrc, (__attribute__((address_space(3))) void*)(lds + TOUCH_BASE), 4,
                                                                 row * static_cast<int>(p.d_sm) * d_elem + in_row * DG_C_TOUCH_BYTES, 0, 0, 0);
                    }
                    ++touch_next;
                }
            };

            v4f part[DEPTH + 1];
            #pragma unroll
            for (int i = 0; i <= DEPTH; ++i)
                part[i] = v4f{0.f, 0.f, 0.f, 0.f};
            v2f sa_pair[MS / 2], sa_tail = v2f{0.f, 0.f};
            v4f sb4[NS], sb_tail[NS];
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                sb_tail[ns] = v4f{0.f, 0.f, 0.f, 0.f};

            #pragma unroll
            for (int q = 0; q < A_ITERS + B_ITERS; ++q)
                issue_piece(0, 0, q);
            issue_scale_piece(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();

            v8i bf[NS], af[2];
            if (p.dbg != nullptr) t_loop0 = DG_STAMP_CLOCK();
            for (int kb = 0; kb < num_kb; ++kb) {
                const int cur = kb & 1;
                const bool has_next = kb + 1 < num_kb;
                const uint8_t* sc = lds + SC_BASE + cur * SC_STAGE;
                #pragma unroll
                for (int h = 0; h < MS / 2; ++h)
                    sa_pair[h] = v2f{*reinterpret_cast<const float*>(sc + sa_lds_off + (2 * h) * 64),
                                     *reinterpret_cast<const float*>(sc + sa_lds_off + (2 * h + 1) * 64)};
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    sb4[ns] = *reinterpret_cast<const v4f*>(sc + sb_lds_off + (MN ? ns * 64 : (ns >> 1) * 128 + (ns & 1) * 16));
                if constexpr (MN) {
                    // the transpose reads below are asm with hand-placed lgkmcnt waits: hipcc's own LDS loads (the scales)
                    // must be complete before the first of them is issued, or its wait counts would be off
                    #pragma unroll
                    for (int h = 0; h < MS / 2; ++h)
                        asm volatile("" : "+v"(sa_pair[h]));
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        asm volatile("" : "+v"(sb4[ns]));
                }
                if (has_next)
                    issue_scale_piece(cur ^ 1, kb + 1);
#ifdef DG_C_TOUCH                   // (tuning builds only -- negative, see above)
                touch_c(num_kb);
#endif

                const uint8_t* a_tile = MN ? lds + cur * STAGE_BYTES : lds + cur * STAGE_BYTES + (wm * WM) * 128;
                const uint8_t* b_tile = MN ? lds + cur * STAGE_BYTES + A_BYTES : lds + cur * STAGE_BYTES + A_BYTES + (wn * WN) * 128;
                // MN: transpose reads land in bfq / afq and become MFMA operands (bf / af) at their first use, after the wait
                [[maybe_unused]] FragTr bfq[NS], afq[2];
                auto frag_a = [&](int ms) {
                    if constexpr (MN) afq[ms & 1] = load_fragment_tr(a_tile, tr_lane_base, ((wm * (WM / 16) + ms) ^ tr_swz) << 4);
                    else af[ms & 1] = load_fragment(a_tile + ms * 2048, frag_off);
                };
                auto frag_b = [&](int ns) {
                    if constexpr (MN) bfq[ns] = load_fragment_tr(b_tile, tr_lane_base, ((wn * (WN / 16) + ns) ^ tr_swz) << 4);
                    else bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                };
                frag_b(0);
                frag_a(0);

                #pragma unroll
                for (int i = 0; i < TOTAL; ++i) {
                    const int ms = i / NS, ns = i % NS;
                    const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;     // step being promoted
                    const int jms = j / NS, jns = j % NS;
                    if (ms == 0 && ns + 1 < NS)
                        frag_b(ns + 1);
                    if (ns == 0 && ms + 1 < MS)
                        frag_a(ms + 1);
                    if constexpr (MN) {
                        // everything but the reads just issued (4 per fragment) has landed: the operands of this step
                        const int fresh = 4 * ((ms == 0 && ns + 1 < NS ? 1 : 0) + (ns == 0 && ms + 1 < MS ? 1 : 0));
                        if (ms == 0 || ns == 0) {       // the steps that consume a fragment for the first time
                            if (fresh == 8) __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 8));
                            else if (fresh == 4) __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 4));
                            else __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
                        }
                        if (ms == 0) bf[ns] = assemble_fragment_tr(bfq[ns]);
                        if (ns == 0) af[ms & 1] = assemble_fragment_tr(afq[ms & 1]);
                    }
                    const v4f& po = part[(i + 1) & DEPTH];
                    // i < DEPTH: the previous block's tail steps (last M-subtile) with the previous block's scales
                    const v4f& sb = (i >= DEPTH) ? sb4[jns] : sb_tail[jns];
                    mfma_promote_step_pc_scalar(part[i & DEPTH], bf[ns], af[ms & 1], accs[jms][jns], sb,
                                                (i >= DEPTH) ? sa_pair[jms >> 1][jms & 1] : sa_tail[(MS - 1) & 1], po);
                    if (i % SPREAD == SPREAD - 1 && i / SPREAD < A_ITERS + B_ITERS && has_next)
                        issue_piece(cur ^ 1, kb + 1, i / SPREAD);
                }
                sa_tail = sa_pair[(MS - 1) >> 1];
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    sb_tail[ns] = sb4[ns];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            if (p.dbg != nullptr) t_loop1 = DG_STAMP_CLOCK();
            #pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                const int j = TOTAL - DEPTH + i;
                const v4f& po = part[(TOTAL + i + 1) & DEPTH];
                promote_only_pc_scalar(accs[j / NS][j % NS], sb_tail[j % NS], sa_tail[(MS - 1) & 1], po);
            }
        }

        v4f out[MS][NS];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                out[ms][ns] = v4f{accs[ms][ns][0], accs[ms][ns][1], accs[ms][ns][2], accs[ms][ns][3]};
        store_tile<MS, NS, false, false, MN, true>(p, t, (p.gemm_type == kKGrouped ? t.group : ad_group) * p.d_sg, out, t.m0 + wm * WM,
                                             t.n0 + wn * WN);
        if (p.dbg != nullptr && tile_id == blockIdx.x) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_stamp(p, NW, 0, t_entry);
            dbg_stamp(p, NW, 1, t_loop0);
            dbg_stamp(p, NW, 2, t_loop1);
            dbg_stamp(p, NW, 3, DG_STAMP_CLOCK());
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int SPREAD, bool MN = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_pipe_pc_kernel(const GemmParams p) {
    pipe_pc_kernel_body<BM, BN, WAVES_M, WAVES_N, SPREAD, MN>(p);
}

// LDS-DMA pieces in groups of four that share ONE M0 value (round 4).  The LDS address of a `buffer_load ... lds` is M0 + the instruction's
// immediate offset + 16 * lane, and the immediate is added to the memory address as well.  A wave that owns FOUR CONSECUTIVE 1 KiB units of a
// tile issues them with immediates 0 / 1024 / 2048 / 3072 against one M0 value instead of rewriting M0 (s_mov m0 + the wait state behind
// it) in front of every piece: tools/ubench/frag_rate.hip measured 733 against 792 cycles per four pieces next to a matrix stream
// (profiles/r04_probe/frag_rate.log).  The immediate on the memory side is taken back out of the per-lane offset; so that this never
// goes negative the descriptor's base is moved DOWN by M0_SHARE_BIAS bytes and every offset up by the same amount (the range check
// is relative to the base: num_records grows by the bias; nothing below the real base is ever addressed).
#ifndef DG_M0_SHARE
#define DG_M0_SHARE 1
#endif
constexpr int M0_SHARE_BIAS = 4096;
// (a macro, not a function: hipcc's host pass rejects the gfx950-only 16-byte form of the builtin wherever it is not inside a device template)
#define DG_LDS_DMA_PIECE_SUB(rsrc, lds_group_base, voff, soff, sub, aux)                                                                  \
    do {                                                                                                                                  \
        auto* dg_dst_ = (__attribute__((address_space(3))) void*)(lds_group_base);                                                        \
        switch ((sub) & 3) { /* the immediate must be a literal; `sub` is a constant after unrolling */                                   \
        case 0: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dg_dst_, 16, voff, soff, 0, aux); break;                                   \
        case 1: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dg_dst_, 16, voff, soff, 1024, aux); break;                                \
        case 2: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dg_dst_, 16, voff, soff, 2048, aux); break;                                \
        default: __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, dg_dst_, 16, voff, soff, 3072, aux); break;                               \
        }                                                                                                                                 \
    } while (0)

__device__ __forceinline__ void raw_barrier() {
    // A bare s_barrier: __syncthreads() would add a vmcnt(0) fence and drain the LDS-DMA queue.
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// Duo kernel: the ring kernel's data movement (3-slot A ring, 2-slot B ring, counted vmcnt, asm scale loads) with the
// K block cut into four role-split segments per wave:
//     L_a : scales of the block, LDS-DMA of A(kb+2), fragment reads  B(kb) x NS  and  A(kb) subtiles 0 .. MS/2-1
//     M_a : MS/2 * NS  MFMA + promotion steps, nothing else in the instruction stream
//     L_b : LDS-DMA of B(kb+2), fragment reads A(kb) subtiles MS/2 .. MS-1, wait "my pieces of block kb+1 landed"
//     M_b : the other MS/2 * NS steps
// with a workgroup barrier in front of every segment and the upper half of the waves (the second wave of every SIMD)
// running ONE segment behind the lower half.  At any time one wave of a SIMD is in a pure matrix segment -- and
// alone it sustains the matrix pipe's 32-cycle issue rate, which a wave that also issues LDS / DMA / scalar work
// between its MFMAs does not (an in-order wave cannot slip an MFMA into a gap shorter than 32 cycles, so two mixed
// streams on one SIMD leave the pipe idle ~25 % of the time) -- while its partner does all the memory work in the
// shadow.  Barrier t (counting the lower half's segments) certifies: t = 4kb: block kb landed everywhere and A(kb-1)
// is dead; t = 4kb+2: B(kb) is in everybody's registers.  Prefetch distance: A 1.5 K blocks, B 1.
// ---------------------------------------------------------------------------------------------------------------
// Scale landing registers of the duo kernel: with the A rows of a wave interleaved (LDS row position ms * 16 + i holds
// tile row i * MS + ms) a lane's MS row scales are MS consecutive floats of the MN-major SFA: MS / 4 dwordx4 loads.
template <int MS>
struct ScaleLandingV { v4f q[MS / 4]; float sb; };

template <int MS>
__device__ __forceinline__ void issue_scale_loads_v(ScaleLandingV<MS>& l, const v4i& sfa_rsrc, int sfa_voff,
                                                    const v4i& sfb_rsrc, int sfb_voff) {
    static_assert(MS == 8 || MS == 4, "unrolled by hand");
    if constexpr (MS == 8)
        asm volatile(
            "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
            "buffer_load_dwordx4 %0, %3, %4, 0 offen\n\t"
            "buffer_load_dwordx4 %1, %3, %4, 0 offen offset:16\n\t"
            "buffer_load_dword %2, %5, %6, 0 offen"
            : "=&v"(l.q[0]), "=&v"(l.q[1]), "=&v"(l.sb)
            : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff), "s"(sfb_rsrc)
            : "memory");
    else
        asm volatile(
            "s_nop 4\n\t"
            "buffer_load_dwordx4 %0, %2, %3, 0 offen\n\t"
            "buffer_load_dword %1, %4, %5, 0 offen"
            : "=&v"(l.q[0]), "=&v"(l.sb)
            : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff), "s"(sfb_rsrc)
            : "memory");
}

// 256 x 224 tile (round 6): a wave tile of 112 columns may straddle ONE 128-column boundary of the SFB grid (reference: the two-value SFB of
// sm90_fp8_gemm_1d2d.cuh:232-237, 290-291, 342-346): both candidate values land, `delta` = byte distance from the first block's scale to the
// second's (0 when the wave tile lies in one block or the second would lie past N: the same value twice).
template <int MS>
struct ScaleLandingV2 { v4f q[MS / 4]; float sb, sb1; int delta; };

template <int MS>
__device__ __forceinline__ void issue_scale_loads_v2(ScaleLandingV2<MS>& l, const v4i& sfa_rsrc, int sfa_voff,
                                                     const v4i& sfb_rsrc, int sfb_voff) {
    static_assert(MS == 4, "unrolled by hand");
    const int sfb_voff1 = sfb_voff + l.delta;
    asm volatile(
        "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
        "buffer_load_dwordx4 %0, %3, %4, 0 offen\n\t"
        "buffer_load_dword %1, %5, %6, 0 offen\n\t"
        "buffer_load_dword %2, %7, %6, 0 offen"
        : "=&v"(l.q[0]), "=&v"(l.sb), "=&v"(l.sb1)
        : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff), "s"(sfb_rsrc), "v"(sfb_voff1)
        : "memory");
}

template <int ALLOWED, int MS>
__device__ __forceinline__ void wait_landing_v2(ScaleLandingV2<MS>& l) {
    static_assert(ALLOWED >= 0 && ALLOWED < 64, "vmcnt is a 6-bit counter");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(ALLOWED, 0));
    asm volatile("" : "+v"(l.q[0]), "+v"(l.sb), "+v"(l.sb1) :: "memory");
}

// The wait goes through the builtin so that hipcc's own waitcnt pass sees the LDS counter drained and does not re-wait
// for the fragment reads inside the following matrix segment; the empty asm ties the landing registers to this point.
template <int ALLOWED, int MS>
__device__ __forceinline__ void wait_landing_v(ScaleLandingV<MS>& l) {
    static_assert(ALLOWED >= 0 && ALLOWED < 64, "vmcnt is a 6-bit counter");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(ALLOWED, 0));
    if constexpr (MS == 8)
        asm volatile("" : "+v"(l.q[0]), "+v"(l.q[1]), "+v"(l.sb) :: "memory");
    else
        asm volatile("" : "+v"(l.q[0]), "+v"(l.sb) :: "memory");
}

// Scale landing registers for an MN-major A tile: the hardware transpose read fixes lane i of a fragment to row 16 ms + i of the
// wave's rows (natural order), so a lane's MS row scales lie 16 floats apart in the MN-major SFA: MS dword loads (immediate
// offsets) instead of MS / 4 dwordx4.
template <int MS>
struct ScaleLandingN { float s[MS]; float sb; };

template <int MS>
__device__ __forceinline__ void issue_scale_loads_n(ScaleLandingN<MS>& l, const v4i& sfa_rsrc, int sfa_voff,
                                                    const v4i& sfb_rsrc, int sfb_voff) {
    static_assert(MS == 8, "unrolled by hand");
    asm volatile(
        "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
        "buffer_load_dword %0, %9, %10, 0 offen\n\t"
        "buffer_load_dword %1, %9, %10, 0 offen offset:64\n\t"
        "buffer_load_dword %2, %9, %10, 0 offen offset:128\n\t"
        "buffer_load_dword %3, %9, %10, 0 offen offset:192\n\t"
        "buffer_load_dword %4, %9, %10, 0 offen offset:256\n\t"
        "buffer_load_dword %5, %9, %10, 0 offen offset:320\n\t"
        "buffer_load_dword %6, %9, %10, 0 offen offset:384\n\t"
        "buffer_load_dword %7, %9, %10, 0 offen offset:448\n\t"
        "buffer_load_dword %8, %11, %12, 0 offen"
        : "=&v"(l.s[0]), "=&v"(l.s[1]), "=&v"(l.s[2]), "=&v"(l.s[3]), "=&v"(l.s[4]), "=&v"(l.s[5]), "=&v"(l.s[6]),
          "=&v"(l.s[7]), "=&v"(l.sb)
        : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff), "s"(sfb_rsrc)
        : "memory");
}

// The same landing registers for a ROW-major SFA ([M][K / 128] floats, the layout the reference's callers hold before its layout step:
// tests/test_fp8_fp4.py:45-55, csrc/jit_kernels/impls/smxx_layout.hpp:120-153 is the transpose this saves): a lane's MS rows lie
// `row_stride` bytes apart, so the MS dword loads take their row offsets from the scalar offset operand (k * row_stride, formed by SALU
// inside the block: two SGPRs instead of MS, nothing for hipcc to spill).  voff = first row * row_stride + 4 * K block.
template <int MS>
__device__ __forceinline__ void issue_scale_loads_rm(ScaleLandingN<MS>& l, const v4i& sfa_rsrc, int sfa_voff, int row_stride,
                                                     const v4i& sfb_rsrc, int sfb_voff) {
    static_assert(MS == 8, "unrolled by hand");
    int tmp;
    asm volatile(
        "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
        "buffer_load_dword %0, %10, %11, 0 offen\n\t"
        "buffer_load_dword %1, %10, %11, %12 offen\n\t"
        "s_mul_i32 %9, %12, 2\n\t"
        "buffer_load_dword %2, %10, %11, %9 offen\n\t"
        "s_mul_i32 %9, %12, 3\n\t"
        "buffer_load_dword %3, %10, %11, %9 offen\n\t"
        "s_mul_i32 %9, %12, 4\n\t"
        "buffer_load_dword %4, %10, %11, %9 offen\n\t"
        "s_mul_i32 %9, %12, 5\n\t"
        "buffer_load_dword %5, %10, %11, %9 offen\n\t"
        "s_mul_i32 %9, %12, 6\n\t"
        "buffer_load_dword %6, %10, %11, %9 offen\n\t"
        "s_mul_i32 %9, %12, 7\n\t"
        "buffer_load_dword %7, %10, %11, %9 offen\n\t"
        "buffer_load_dword %8, %13, %14, 0 offen"
        : "=&v"(l.s[0]), "=&v"(l.s[1]), "=&v"(l.s[2]), "=&v"(l.s[3]), "=&v"(l.s[4]), "=&v"(l.s[5]), "=&v"(l.s[6]),
          "=&v"(l.s[7]), "=&v"(l.sb), "=&s"(tmp)
        : "v"(sfa_voff), "s"(sfa_rsrc), "s"(row_stride), "v"(sfb_voff), "s"(sfb_rsrc)
        : "memory");
}

template <int ALLOWED, int MS>
__device__ __forceinline__ void wait_landing_n(ScaleLandingN<MS>& l) {
    static_assert(ALLOWED >= 0 && ALLOWED < 64, "vmcnt is a 6-bit counter");
    static_assert(MS == 8, "unrolled by hand");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(ALLOWED, 0));
    asm volatile("" : "+v"(l.s[0]), "+v"(l.s[1]), "+v"(l.s[2]), "+v"(l.s[3]), "+v"(l.s[4]), "+v"(l.s[5]), "+v"(l.s[6]),
                      "+v"(l.s[7]), "+v"(l.sb) :: "memory");
}

// One name for both landing forms (interleaved rows: dwordx4 loads; natural rows: dword loads).
template <int MS> __device__ __forceinline__ void issue_scale_loads_any(ScaleLandingV<MS>& l, const v4i& ra, int va, const v4i& rb, int vb) { issue_scale_loads_v<MS>(l, ra, va, rb, vb); }
template <int MS> __device__ __forceinline__ void issue_scale_loads_any(ScaleLandingN<MS>& l, const v4i& ra, int va, const v4i& rb, int vb) { issue_scale_loads_n<MS>(l, ra, va, rb, vb); }
template <int ALLOWED, int MS> __device__ __forceinline__ void wait_landing_any(ScaleLandingV<MS>& l) { wait_landing_v<ALLOWED, MS>(l); }
template <int ALLOWED, int MS> __device__ __forceinline__ void wait_landing_any(ScaleLandingN<MS>& l) { wait_landing_n<ALLOWED, MS>(l); }
template <int MS> __device__ __forceinline__ void issue_scale_loads_any(ScaleLandingV2<MS>& l, const v4i& ra, int va, const v4i& rb, int vb) { issue_scale_loads_v2<MS>(l, ra, va, rb, vb); }
template <int ALLOWED, int MS> __device__ __forceinline__ void wait_landing_any(ScaleLandingV2<MS>& l) { wait_landing_v2<ALLOWED, MS>(l); }
template <int MS> __device__ __forceinline__ float landed_sfa(const ScaleLandingV2<MS>& l, int ms) { return l.q[ms / 4][ms % 4]; }
template <int MS> __device__ __forceinline__ float landed_sfa(const ScaleLandingV<MS>& l, int ms) { return l.q[ms / 4][ms % 4]; }
template <int MS> __device__ __forceinline__ float landed_sfa(const ScaleLandingN<MS>& l, int ms) { return l.s[ms]; }
template <int MS, bool NATURAL, bool TWO_SFB = false> struct ScaleLandingSel { typedef ScaleLandingV<MS> type; };
template <int MS> struct ScaleLandingSel<MS, true, false> { typedef ScaleLandingN<MS> type; };
template <int MS> struct ScaleLandingSel<MS, false, true> { typedef ScaleLandingV2<MS> type; };

// Scale landing registers of the per-column form (PC: recipe (1, 1, 128), one SFB value per ROW of B): the lane's MS = 4 row scales (one
// dwordx4 of the MN-major SFA, interleaved rows) and its 16 column scales -- N-subtile ns, accumulator register r sits on column
// wave_n0 + (ns >> 1) * 32 + (lane >> 4) * 8 + (ns & 1) * 4 + r (b_row_perm), i.e. four dwordx4 of the MN-major SFB at byte offsets
// 0 / 16 / 128 / 144 from the lane's first column.
struct ScaleLandingPC { v4f sa; v4f sb[4]; };

__device__ __forceinline__ void issue_scale_loads_pc(ScaleLandingPC& l, const v4i& sfa_rsrc, int sfa_voff, const v4i& sfb_rsrc, int sfb_voff) {
    asm volatile(
        "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
        "buffer_load_dwordx4 %0, %5, %6, 0 offen\n\t"
        "buffer_load_dwordx4 %1, %7, %8, 0 offen\n\t"
        "buffer_load_dwordx4 %2, %7, %8, 0 offen offset:16\n\t"
        "buffer_load_dwordx4 %3, %7, %8, 0 offen offset:128\n\t"
        "buffer_load_dwordx4 %4, %7, %8, 0 offen offset:144"
        : "=&v"(l.sa), "=&v"(l.sb[0]), "=&v"(l.sb[1]), "=&v"(l.sb[2]), "=&v"(l.sb[3])
        : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff), "s"(sfb_rsrc)
        : "memory");
}

template <int ALLOWED>
__device__ __forceinline__ void wait_landing_pc(ScaleLandingPC& l) {
    static_assert(ALLOWED >= 0 && ALLOWED < 64, "vmcnt is a 6-bit counter");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(ALLOWED, 0));
    asm volatile("" : "+v"(l.sa), "+v"(l.sb[0]), "+v"(l.sb[1]), "+v"(l.sb[2]), "+v"(l.sb[3]) :: "memory");
}

// MFMA + promotion of an older step with one scale PER accumulator register (the per-column form's ring tail: products formed a K block ago)
__device__ __forceinline__ void mfma_promote_step_v(v4f& part_new, const v8i& rows_operand, const v8i& cols_operand, float (&c)[4],
                                                    const float (&s)[4], const v4f& part_old) {
    asm volatile(
        "v_mfma_f32_16x16x128_f8f6f4 %0, %5, %6, 0\n\t"
        "v_fmac_f32 %1, %7, %11\n\t"
        "v_fmac_f32 %2, %8, %12\n\t"
        "v_fmac_f32 %3, %9, %13\n\t"
        "v_fmac_f32 %4, %10, %14"
        : "=&v"(part_new), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
        : "v"(rows_operand), "v"(cols_operand), "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(part_old[0]), "v"(part_old[1]),
          "v"(part_old[2]), "v"(part_old[3])
        : "memory");
}

__device__ __forceinline__ void promote_only_v(float (&c)[4], const float (&s)[4], const v4f& part_old) {
    asm volatile(
        "s_nop 3\n\t"
        "v_fmac_f32 %0, %4, %8\n\t"
        "v_fmac_f32 %1, %5, %9\n\t"
        "v_fmac_f32 %2, %6, %10\n\t"
        "v_fmac_f32 %3, %7, %11"
        : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
        : "v"(s[0]), "v"(s[1]), "v"(s[2]), "v"(s[3]), "v"(part_old[0]), "v"(part_old[1]), "v"(part_old[2]), "v"(part_old[3])
        : "memory");
}

// PERSIST: persistent launch (one workgroup per CU walks the tile list) with cross-tile prologue prefetch.
// B_MN: operand B is MN-major ([K][N], unit stride along n, row pitch b_sk): the nn / tn layouts without the re-majoring
// pass.  LDS-DMA pieces are 4 k-rows x 256 bytes, B fragments come through the hardware transpose read, B rows keep their
// natural order (=> 8-byte instead of 16-byte BF16 stores).  See load_fragment_tr.
// SPLITK (persistent launch only): tail balancing for tile counts just above a multiple of the CU count.  The first
// p.sk_first_tile tiles are walked as usual; the remaining p.sk_tiles tiles (the partial last round) are cut along K into
// p.sk_factor pieces, one per otherwise idle CU.  Every piece writes its FP32 partial tile to the workspace; the sum (in piece order:
// bit-repeatable) and the output stores are done by dg_split_k_reduce_kernel, launched right behind on the same stream.
// A_MN: operand A is MN-major ([K][M], unit stride along m, row pitch a_sk): the tn / tt layouts without the re-majoring pass.
// Same piece / transpose-read geometry as B_MN; A rows keep their natural order (subtile ms of a wave = rows 16 ms .. 16 ms + 15),
// so a lane's row scales come as MS dword loads (ScaleLandingN) and the epilogue runs with INTERLEAVED_ROWS = false.
// (The timing ablations this kernel was tuned with -- no stagger, priorities, early barriers, pieces between MFMAs, per-step
// traces ... -- live in tools/experiments/fp8_gemm_experiments.hpp, DG_EXPERIMENTS builds only.)
// K_TAIL: K need not be a multiple of 128 (whole 16-byte chunks, K > 128): the partial last K block is computed once per tile after
// the loop (separate instantiations: the stage costs registers that the tuned whole-block kernels do not have to spare).
// MERGED (128-row tiles): TWO segments per K block instead of four -- L: scales, every piece of block kb+2, every fragment of block
// kb; M: all MS * NS steps -- with a 3-slot B ring (B(kb+2) lands in B(kb-1)'s slot, so no barrier has to separate "B(kb) is in
// everybody's registers" from the refill).  A 64 x 64 wave tile has the registers for all of its fragments, and with 8-step
// matrix segments the four barrier round trips and the load segments (longer than the matrix segments they hide behind) cost
// the 128-row tile 1.43 k cycles per K block against 1.02 k of matrix work; 16-step segments halve the barriers.  Waits: at the
// end of L "everything but this segment's own loads has landed" (my pieces of block kb+1, certified to the others by the next
// barrier), at the end of M the scales of block kb+1 (straight-line from their loads, in front of the loop's back edge).
// STREAM_A (256-row tiles, K-major operands): TWO segments per K block like MERGED, for a wave tile whose fragments do not all fit in
// registers: L reads the B fragments and the first two A fragments, M runs ALL MS * NS steps and streams the remaining A
// fragments through a 4-deep register ring (one fragment = two ds_read_b128 per row of NS steps).  Half the barrier round trips of
// the four-segment schedule (2 per K block and wave instead of 4: each costs ~70 cycles of matrix pipe).  The LDS-DMA work is split
// by role so that the 3-slot A ring / 2-slot B ring still suffice with the halves one segment apart: the LOWER half issues all pieces
// of B(kb+1) in its L(kb) (B(kb-1)'s slot: everybody's L(kb-1) reads are done), the UPPER half all pieces of A(kb+2) in its L(kb)
// (A(kb-1)'s slot: the upper half itself finished M(kb-1) last).  Prologue: A(0) B(0) A(1).
// LDS bytes of a duo body: three A slots, two B slots (MERGED: three)
// Step i of a matrix segment of the duo forms -> its N-subtile: boustrophedon order (round 5), so that the B-side operand of a step equals its
// predecessor's at every change of the A fragment -- one operand of EVERY MFMA is then unchanged from the step before.  The part is power-limited:
// same box, four alternating pairs, C2 93.76 -> 93.16 us, C3 nt 30.65 -> 30.42 (profiles/r05_probe/duo_serpentine_order_ab.log); every
// accumulator still sees its own K blocks in order: same bits.  -DDG_DUO_ROWMAJOR restores the old order.
#ifndef DG_DUO_ROWMAJOR
#define DUO_NS(i) ((((i) / NS) & 1) ? NS - 1 - (i) % NS : (i) % NS)
#else
#define DUO_NS(i) ((i) % NS)
#endif
constexpr int duo_lds_bytes(int bm, int bn, bool merged) { return 3 * bm * 128 + (merged ? 3 : 2) * (bn == 224 ? 256 : bn) * 128; }   // (224: a 256-row slot)

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PERSIST = false, bool B_MN = false, bool SPLITK = false, bool A_MN = false,
          bool K_TAIL = false, bool MERGED = false, bool STREAM_A = false, bool PC = false, bool SFA_RM = false, int CALLER = 0>
__device__ __forceinline__ void duo_kernel_body(const GemmParams& p, uint8_t* const lds) {
    static_assert(!SFA_RM || (BM == 256 && PERSIST && !A_MN && !B_MN && !K_TAIL && !MERGED && !SPLITK && !STREAM_A && !PC),
                  "SFA_RM: the dense persistent 256-row form reading a row-major SFA in place");
    static_assert(!PC || (MERGED && !A_MN && !B_MN && !K_TAIL && !SPLITK && !STREAM_A), "PC: the two-segment 128-row tile, K-major operands");
    constexpr int NW = WAVES_M * WAVES_N;
    // N224 (round 6): 256 x 224 tiles, 4 x 2 waves, wave tile 64 x 112 (MS = 4, NS = 7) -- N = 7168 is 32 x 224: C3 (2048 x 7168 x 2048) becomes
    // exactly 256 tiles instead of 224 of 256 x 256 on 256 CUs, 4096 x 7168 two full rounds instead of 1.75.  The B slot stays 256 rows (the 32 rows
    // behind the tile are the next tile's, or out of range: never read); the seventh N-subtile keeps its natural column order (b_row_perm); a wave
    // tile may straddle one 128-column boundary of the SFB grid: both values land (ScaleLandingV2), the scale product of a lane is formed per
    // 32-column GROUP of its columns (the boundary is a multiple of 16: a group that straddles it splits between lane groups 0-1 and 2-3).
    constexpr bool N224 = BN == 224;
    static_assert(!N224 || (BM == 256 && WAVES_M == 4 && WAVES_N == 2 && PERSIST && !B_MN && !SPLITK && !A_MN && !K_TAIL && !MERGED && !STREAM_A && !PC && !SFA_RM),
                  "N224: the dense persistent form with K-major operands");
    constexpr int BN_LDS = N224 ? 256 : BN;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16, HS = MERGED ? MS : MS / 2;
    static_assert(!STREAM_A || (BM == 256 && !MERGED && !A_MN && !B_MN && !K_TAIL && !SPLITK), "STREAM_A: dense 256-row tiles, K-major operands");
    constexpr int TOTAL = MS * NS, SEG = HS * NS, DEPTH = 3;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN_LDS * 128, A_SLOTS = 3, B_SLOTS = MERGED ? 3 : 2;
    constexpr int SCALE_LOADS = PC ? 5 : (A_MN || SFA_RM ? MS + 1 : MS / 4 + 1 + (N224 ? 1 : 0));    // vector-memory operations of one issue_scale_loads_any / _pc
    constexpr int B_BASE = A_SLOTS * A_BYTES, LDS_BYTES = B_BASE + B_SLOTS * B_BYTES;
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN_LDS / 8 / NW;
#ifdef DG_A_EARLY                               // (tuning builds: DG_VARIANT_FLAGS=-DDG_A_EARLY=n)
    constexpr int A_EARLY = DG_A_EARLY < A_ITERS ? DG_A_EARLY : A_ITERS;
#else
    // A pieces issued in L_a (next to the scale loads); the rest go with B in L_b.  Any split is legal (A(kb-1)'s slot is dead from barrier 4 kb
    // on, the counted waits only know the total); round 4 same-box A/B on C2 (profiles/r04_probe/a_early_ab.log): 0 / 1 of 4 early 93.8-93.9 us,
    // 2 (the round-1 choice) 94.0-94.8, 3 / 4 95.0-95.7 -- the scale loads want L_a, the pieces want the segment without them
    constexpr int A_EARLY = A_ITERS / 4;
#endif
    // groups of four pieces that share one M0 value (DG_LDS_DMA_PIECE_SUB): K-major operands whose wave share is a multiple of four units
    constexpr bool M0S_A = DG_M0_SHARE && !A_MN && !STREAM_A && A_ITERS % 4 == 0, M0S_B = DG_M0_SHARE && !B_MN && !STREAM_A && B_ITERS % 4 == 0;
    static_assert(!B_MN || (BN == 256 && NW == 8), "MN-major B tile: 128 k-rows x 256 bytes, 32 pieces over 8 waves");
    static_assert(!A_MN || (BM == 256 && NW == 8), "MN-major A tile: 128 k-rows x 256 bytes, 32 pieces over 8 waves");
    static_assert(!SPLITK || (PERSIST && !A_MN), "the K-split tail belongs to the persistent forms");
    static_assert(!K_TAIL || !SPLITK, "K tail and K split are not combined");
    static_assert(!MERGED || (BM == 128 && !A_MN), "the two-segment form needs all fragments of the wave tile in registers");
    static_assert(NW % 2 == 0 && BM % (8 * NW) == 0 && BN_LDS % (8 * NW) == 0, "every wave issues the same number of pieces");
    static_assert(WM % 32 == 0 && (WN % 32 == 0 || N224), "wave tile shape");
    static_assert(N224 || (WN <= 128 && 128 % WN == 0), "one SFB value per wave");
    static_assert(SEG > DEPTH, "the promotion ring must fit in a segment");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert((NW * 8) % 16 == 0 && ((NW * 8) % WN == 0 || WN % (NW * 8) == 0 || (N224 && M0S_B)),
                  "the row permutation of a B piece must be lane-independent (N224: every piece has its own per-lane offset)");

    static_assert(LDS_BYTES == duo_lds_bytes(BM, BN, MERGED), "the kernel wrappers size the LDS array with duo_lds_bytes");
    // (`lds`: the calling kernel's __shared__ array -- a kernel that runs two bodies one after the other gives both the same bytes)

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool upper_half = wave >= NW / 2;
    const int num_kb = p.k / 128;
    int kb0 = 0, nkb = num_kb;                  // K blocks [kb0, kb0 + nkb) of the current work item (SPLITK: a piece of the K range)
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);
    // A rows are interleaved inside a wave's WM rows: LDS row position P holds tile row a_row_of(P).  For the rows of one
    // piece (P = 8 u + j) that is a_unit_row(u) + j * MS: the lane part goes into a_voff, the unit part is wave-uniform.
    auto a_unit_row = [](int u) { return (u / (WM / 8)) * WM + (u & 1) * 8 * MS + ((u % (WM / 8)) >> 1); };
    const int a_voff = piece_row * MS * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;
    const long long t_entry = p.dbg != nullptr ? DG_STAMP_CLOCK() : 0;
    long long t_loop0 = 0, t_loop1 = 0;

    MaskedWalk walk;
#ifndef DG_NO_TABLE_MASK
    if constexpr (PERSIST && !A_MN && !B_MN && !K_TAIL)         // (the launches of launch_contiguous_tabled)
        if (p.table_mode != 0) {
            walk.table_mask = contiguous_tile_mask(p.layout, p.m, p.table_mode, &walk.block_group);
            walk.have_block_groups = true;
        }
#endif
    const int num_launched = gridDim.x;
    const int sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kb_stride = static_cast<int>(p.sfb_sk) * 4;
    const int k_tail = K_TAIL ? (p.k & 127) : 0;    // a partial last K block (multiple of 16 bytes): handled after the loop, see below
    const int num_sf_kb = num_kb + (k_tail != 0);
    [[maybe_unused]] const int sfa_row_stride = static_cast<int>(p.sfa_sm) * 4;          // SFA_RM: bytes between the scale rows of m and m + 1
    const int sfa_extent = SFA_RM ? (p.m - 1) * sfa_row_stride + (num_sf_kb - 1) * sfa_kb_stride + 4
                                  : (p.m - 1) * 4 + (num_sf_kb - 1) * sfa_kb_stride + 4;
    // (PC: the lane's 16 column scales of a K block; columns past N read the next block's head or fall out of range -- never stored)
    const int sfb_extent = (num_sf_kb - 1) * sfb_kb_stride + (PC ? p.n * 4 : 4) + (N224 ? static_cast<int>(p.sfb_sn) * 4 : 0);      // (N224: the next block's value too)
    [[maybe_unused]] const int sfb_lane_off = (lane >> 4) * 32;

    // Per-piece source offsets (rows + chunk: the bounds-checked part of the address) are kernel invariants held in
    // VGPRs; the K block goes in the soffset.  Blocks past the end re-read the last K block into a dead slot -- no
    // out-of-range arithmetic in the loop, the vmcnt counts stay exact, the bytes come from L2.
    int a_piece_voff[A_ITERS], b_piece_voff[B_ITERS];
    #pragma unroll
    for (int q = 0; q < A_ITERS; ++q)
        a_piece_voff[q] = M0S_A ? a_voff + a_unit_row(wave * A_ITERS + q) * lda + M0_SHARE_BIAS - (q & 3) * 1024
                                : a_voff + a_unit_row(wave + NW * q) * lda;
    #pragma unroll
    for (int q = 0; q < B_ITERS; ++q)
        b_piece_voff[q] = M0S_B ? b_row_perm<WN>((wave * B_ITERS + q) * 8 + piece_row) * ldb + src_chunk * 16 + M0_SHARE_BIAS - (q & 3) * 1024
                                : b_voff + b_row_perm<WN>(q * (NW * 8)) * ldb;
    // STREAM_A: the 2 * A_ITERS pieces this wave issues per K block in its role (upper half: A units, lower half: B units)
    [[maybe_unused]] int role_piece_voff[STREAM_A ? 2 * A_ITERS : 1];
    if constexpr (STREAM_A) {
        #pragma unroll
        for (int q = 0; q < 2 * A_ITERS; ++q) {
            const int unit = (upper_half ? wave - NW / 2 : wave) + (NW / 2) * q;
            role_piece_voff[q] = upper_half ? a_voff + a_unit_row(unit) * lda
                                            : b_row_perm<WN>(unit * 8 + piece_row) * ldb + src_chunk * 16;
        }
    }
    // MN-major B: lane l of piece u carries k-row 4u + (l >> 4), source chunk (l & 15) ^ f(k); u = wave + 8q => f lane-constant
    const int ldb_mn = static_cast<int>(p.b_sk), lda_mn = static_cast<int>(p.a_sk);
    const int mn_chunk = ((lane & 15) ^ (((4 * (wave & 1) + (lane >> 4)) & 7) | (((wave >> 2) & 1) << 3))) << 4;
    const int bmn_voff = (lane >> 4) * ldb_mn + mn_chunk;
    const int amn_voff = (lane >> 4) * lda_mn + mn_chunk;
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
        tm.a_base = uniform_ptr(p.a + adg * p.a_sg + static_cast<int64_t>(tt.m0) * (A_MN ? 1 : p.a_sm));
        tm.b_base = uniform_ptr(p.b + static_cast<int64_t>(tt.group) * p.b_sg + static_cast<int64_t>(tt.n0) * (B_MN ? 1 : p.b_sn));
        tm.a_bytes = __builtin_amdgcn_readfirstlane(A_MN ? (p.k - 1) * lda_mn + (p.m - tt.m0) : (imin(tt.m_end - tt.m0, BM) - 1) * lda + p.k);
        tm.b_bytes = __builtin_amdgcn_readfirstlane(B_MN ? (p.k - 1) * ldb_mn + (p.n - tt.n0) : (imin(p.n - tt.n0, BN_LDS) - 1) * ldb + p.k);
        tm.sfa_addr = reinterpret_cast<uint64_t>(p.sfa + adg * p.sfa_sg);
        tm.sfb_addr = reinterpret_cast<uint64_t>(p.sfb + static_cast<int64_t>(tt.group) * p.sfb_sg +
                                                 (PC ? static_cast<int64_t>(tt.n0 + wn * WN)           // MN-major SFB: one value per column
                                                     : static_cast<int64_t>((tt.n0 + wn * WN) / 128) * p.sfb_sn));
        tm.sfa_voff = (tt.m0 + wm * WM + (lane & 15) * (A_MN ? 1 : MS)) * (SFA_RM ? sfa_row_stride : 4);
        return tm;
    };
    auto scale_rsrc = [&](uint64_t addr, int extent) {
        return v4i{__builtin_amdgcn_readfirstlane(static_cast<int>(addr)),
                   __builtin_amdgcn_readfirstlane(static_cast<int>(addr >> 32) & 0xffff),
                   __builtin_amdgcn_readfirstlane(extent), 0x00020000};
    };
    auto issue_a_piece_r = [&](const uint8_t* base, int bytes, int slot_off, int j, int q) {
        const int unit = wave + NW * q;
        if constexpr (M0S_A) {
            DG_LDS_DMA_PIECE_SUB(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base) - M0_SHARE_BIAS, 0, bytes + M0_SHARE_BIAS, 0x00020000),
                                 lds + slot_off + (wave * A_ITERS + (q & ~3)) * 1024, a_piece_voff[q], (kb0 + imin(j, nkb - 1)) * 128, q, 0);
            return;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, bytes, 0x00020000), (__attribute__((address_space(3))) void*)(lds + slot_off + unit * 1024), 16,
            A_MN ? amn_voff : a_piece_voff[q],
            A_MN ? ((kb0 + imin(j, nkb - 1)) * 128 + 4 * unit) * lda_mn : (kb0 + imin(j, nkb - 1)) * 128, 0, 0);
    };
    auto issue_b_piece_r = [&](const uint8_t* base, int bytes, int slot_off, int j, int q) {
        const int unit = wave + NW * q;
        if constexpr (M0S_B) {
            DG_LDS_DMA_PIECE_SUB(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base) - M0_SHARE_BIAS, 0, bytes + M0_SHARE_BIAS, 0x00020000),
                                 lds + B_BASE + slot_off + (wave * B_ITERS + (q & ~3)) * 1024, b_piece_voff[q], (kb0 + imin(j, nkb - 1)) * 128, q, 0);
            return;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(base), 0, bytes, 0x00020000), (__attribute__((address_space(3))) void*)(lds + B_BASE + slot_off + unit * 1024), 16,
            B_MN ? bmn_voff : b_piece_voff[q],
            B_MN ? ((kb0 + imin(j, nkb - 1)) * 128 + 4 * unit) * ldb_mn : (kb0 + imin(j, nkb - 1)) * 128, 0, 0);
    };
    // Prologue pieces of a tile: A(0) B(0) A(1) B(1) into ring slots 0 / 1.  Issued at kernel entry for the first tile
    // and, in the persistent launch, for tile i+1 as soon as tile i's K loop has released the LDS -- i.e. BEFORE tile i's
    // output stores, so that the cold-start latency of a tile and its predecessor's store tail overlap.  Only LDS-DMA
    // travels ahead: a VGPR-destination load (the scales) must reach its wait in straight-line code, because hipcc is
    // free to copy the destination registers at any control-flow join in between -- before the data has arrived.
    typename ScaleLandingSel<MS, A_MN || SFA_RM, N224>::type land;
    // N224: byte distance to the second SFB value of a tile's wave tile (0: one block, or the next block lies past N) and its boundary column
    [[maybe_unused]] auto sfb_delta_of = [&](const Tile& tt) {
        const int col0 = tt.n0 + wn * WN;
        return __builtin_amdgcn_readfirstlane(((col0 >> 7) + 1) * 128 < p.n ? static_cast<int>(p.sfb_sn) * 4 : 0);
    };
    [[maybe_unused]] ScaleLandingPC land_pc0, land_pc1;        // PC: block kb's scales in one, block kb+1's landing in the other
    // halves: 0 = block 0 only, 1 = block 1 only, 2 = both.  (Round 4: the FIRST tile of a workgroup issues block 0 the moment its
    // coordinates are known -- 1.6 k cycles of accumulator zeroing and descriptor set-up used to run in front of the first load:
    // tools/prologue_stamps.py, profiles/r04_probe/prologue_stamps.log.)
    auto issue_prologue = [&](const Tile& tt, int halves = 2) {
        const TileMem tm = tile_mem(tt);
        if (halves != 1) {
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece_r(tm.a_base, tm.a_bytes, 0, 0, q);
            #pragma unroll
            for (int q = 0; q < B_ITERS; ++q) issue_b_piece_r(tm.b_base, tm.b_bytes, 0, 0, q);
        }
        if (halves != 0) {
            #pragma unroll
            for (int q = 0; q < A_ITERS; ++q) issue_a_piece_r(tm.a_base, tm.a_bytes, A_BYTES, 1, q);
            if constexpr (!STREAM_A) {          // (STREAM_A: B(1) belongs to the lower half's first load segment)
                #pragma unroll
                for (int q = 0; q < B_ITERS; ++q) issue_b_piece_r(tm.b_base, tm.b_bytes, B_BYTES, 1, q);
            }
        }
    };

    // Tile iteration state: (tile_id, pass); contiguous layout with BM = 2 x alignment: a tile whose halves belong to two
    // groups is walked twice.
    int tile_id = blockIdx.x, pass = 0;
    bool prefetched = false, first_tile = true;
    // SPLITK: virtual tile ids >= sk_first_tile are pieces w = id - sk_first_tile of the tail: tile sk_first_tile + w % sk_tiles,
    // K piece w / sk_tiles of sk_factor (sk_first_tile is a multiple of the grid size, so a workgroup meets at most one piece)
    struct Piece { int kb0, nkb, tail, index, factor; bool split; };
    auto work_of = [&](int id, int ps, Piece& pc) {
        pc = Piece{0, num_kb, 0, 0, 1, false};
        if constexpr (SPLITK) {
            // table launch: every tile is a split tile, count and pieces come from the device-side table
            const int sk_first = table_launch(p) ? 0 : p.sk_first_tile;
            const int sk_tiles = table_launch(p) ? table_count(p, walk) * p.num_n_tiles : p.sk_tiles;
            const int sk_factor = table_launch(p) ? table_pieces(p, sk_tiles) : p.sk_factor;
            if (id >= sk_first && sk_factor >= 2) {
                const int w = id - sk_first;
                if (w >= sk_tiles * sk_factor) {
                    Tile none;
                    none.valid = false;
                    return none;
                }
                pc.tail = w % sk_tiles;
                pc.index = w / sk_tiles;
                pc.factor = sk_factor;
                pc.kb0 = pc.index * num_kb / sk_factor;
                pc.nkb = (pc.index + 1) * num_kb / sk_factor - pc.kb0;
                pc.split = true;
                id = sk_first + pc.tail;
            }
        }
        return get_tile<BM, BN>(p, id, walk, ps);
    };
    Piece piece, piece_next;
#if defined(DG_STAMP_ISSUE) && DG_STAMP_ISSUE == 3
    if (p.dbg != nullptr) t_loop1 = DG_STAMP_CLOCK();
#endif
    Tile t = work_of(tile_id, pass, piece);
#if defined(DG_STAMP_ISSUE) && DG_STAMP_ISSUE == 2
    if (p.dbg != nullptr) t_loop1 = DG_STAMP_CLOCK();
#endif
    while (t.valid) {
        kb0 = piece.kb0;
        nkb = piece.nkb;
        const bool early_block0 = !prefetched && t.m_end > t.m0;
        if (early_block0)
            issue_prologue(t, 0);               // block 0 flies while the accumulators are zeroed and the scale descriptors are built
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
            tn = work_of(tile_id, pass, piece_next);
            if (PERSIST && tn.valid && tn.m_end > tn.m0) {
                kb0 = piece_next.kb0;                   // the prologue pieces and scales below belong to the NEXT work item
                nkb = piece_next.nkb;
                // The next tile's first two K blocks and the scales of its block 0, drained HERE -- in front of this tile's
                // output stores: once stores are pending they count towards vmcnt, and waiting for "block 0 has landed" at
                // the top of the next tile would wait for (nearly) all of them.  Drained now, the next tile starts without
                // any wait and the stores overlap its first K block instead of standing between the two tiles.
                const TileMem tmn = tile_mem(tn);
                issue_prologue(tn);
                if constexpr (PC) {
                    issue_scale_loads_pc(land_pc0, scale_rsrc(tmn.sfa_addr, sfa_extent), tmn.sfa_voff + kb0 * sfa_kb_stride,
                                         scale_rsrc(tmn.sfb_addr, sfb_extent), sfb_lane_off + kb0 * sfb_kb_stride);
                    wait_landing_pc<0>(land_pc0);
                } else if constexpr (SFA_RM) {
                    issue_scale_loads_rm<MS>(land, scale_rsrc(tmn.sfa_addr, sfa_extent), tmn.sfa_voff + kb0 * sfa_kb_stride, sfa_row_stride,
                                             scale_rsrc(tmn.sfb_addr, sfb_extent), kb0 * sfb_kb_stride);
                    wait_landing_any<0, MS>(land);
                } else {
                if constexpr (N224)
                    land.delta = sfb_delta_of(tn);
                issue_scale_loads_any<MS>(land, scale_rsrc(tmn.sfa_addr, sfa_extent), tmn.sfa_voff + kb0 * sfa_kb_stride,
                                          scale_rsrc(tmn.sfb_addr, sfb_extent), kb0 * sfb_kb_stride);
                wait_landing_any<0, MS>(land);      // the landed values stay in `land` until the next tile's L_a(0) consumes them
                }
                next_prefetched = true;
            }
        };

        float acc[MS][NS][4];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                #pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[ms][ns][r] = 0.f;
#ifndef DG_NO_ZERO_TIE
                // materialised HERE, under the flight of block 0: hipcc sinks a plain zero-initialisation to the first MFMA -- behind the
                // landing wait and the barrier, ~125 v_mov per wave on the critical path of every tile
                // (the non-persistent 256-row form -- selectable by name only -- answers the tie with 27 prologue spills: left alone)
                if constexpr (PERSIST || MERGED)
                    asm volatile("" : "+v"(acc[ms][ns][0]), "+v"(acc[ms][ns][1]), "+v"(acc[ms][ns][2]), "+v"(acc[ms][ns][3]));
#endif
            }

        if (t.m_end > t.m0) {
            const TileMem tm = tile_mem(t);
            const v4i sfa_rsrc = scale_rsrc(tm.sfa_addr, sfa_extent), sfb_rsrc = scale_rsrc(tm.sfb_addr, sfb_extent);
            const int sfa_voff = tm.sfa_voff;
            auto issue_a_piece = [&](int slot_off, int j, int q) { issue_a_piece_r(tm.a_base, tm.a_bytes, slot_off, j, q); };
            auto issue_b_piece = [&](int slot_off, int j, int q) { issue_b_piece_r(tm.b_base, tm.b_bytes, slot_off, j, q); };
            auto issue_scales = [&](typename ScaleLandingSel<MS, A_MN || SFA_RM, N224>::type& l, int j) {
                const int jj = kb0 + imin(j, nkb - 1);   // past the end: the last block's scales again (never consumed)
                if constexpr (SFA_RM)
                    issue_scale_loads_rm<MS>(l, sfa_rsrc, sfa_voff + jj * sfa_kb_stride, sfa_row_stride, sfb_rsrc, jj * sfb_kb_stride);
                else
                    issue_scale_loads_any<MS>(l, sfa_rsrc, sfa_voff + jj * sfa_kb_stride, sfb_rsrc, jj * sfb_kb_stride);
            };

            float scale[MS], scale_tail = 0.f;
            // N224: scale product per M-subtile and 32-column GROUP of the lane's columns (groups 0 .. 2 = the subtile pairs, 3 = the seventh
            // subtile); `second[g]`: this lane's columns of group g lie at or behind the wave tile's 128-column boundary (the second SFB value)
            [[maybe_unused]] float scaleg[N224 ? MS : 1][4], tailg[4] = {0.f, 0.f, 0.f, 0.f};
            [[maybe_unused]] bool second[4] = {false, false, false, false};
            if constexpr (N224) {
                const int col0 = t.n0 + wn * WN;
                const int to_boundary = (128 - (col0 & 127)) & 127;                     // 0: col0 is itself a boundary, none inside before + 128
                const int boundary = to_boundary == 0 ? 128 : to_boundary;              // first boundary strictly behind col0, relative to col0
                #pragma unroll
                for (int g = 0; g < 3; ++g)
                    second[g] = 32 * g + (lane >> 4) * 8 >= boundary;
                second[3] = 96 >= boundary;
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    #pragma unroll
                    for (int g = 0; g < 4; ++g)
                        scaleg[ms][g] = 0.f;
            }
            v4f part[DEPTH + 1];
            #pragma unroll
            for (int i = 0; i <= DEPTH; ++i)
                part[i] = v4f{0.f, 0.f, 0.f, 0.f};
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                scale[ms] = 0.f;

            // ---- block 0 and its scales must land before the first segment ----
#if defined(DG_STAMP_ISSUE) && DG_STAMP_ISSUE == 1     // (tuning builds: how long does a workgroup run before its first load goes out?  slot 2 = this stamp instead of the loop end)
            if (p.dbg != nullptr && first_tile) t_loop1 = DG_STAMP_CLOCK();
#endif
            if (!prefetched) {
                // SF(0) first, then the pieces of blocks 0 and 1: the wait leaves block 1's pieces in flight (the first K block's
                // own counted wait covers them), so the first segment starts as soon as block 0 is in.  (Straight-line from the
                // scale loads to their wait: hipcc may copy the landing registers at any control-flow join in between.)
                // (block 0's pieces went out at the top of the tile: older than the scale loads, so the counted wait still covers them)
                if constexpr (PC) {
                    issue_scale_loads_pc(land_pc0, sfa_rsrc, sfa_voff + kb0 * sfa_kb_stride, sfb_rsrc, sfb_lane_off + kb0 * sfb_kb_stride);
                    issue_prologue(t, 1);
                    wait_landing_pc<A_ITERS + B_ITERS>(land_pc0);
                } else {
                if constexpr (N224)
                    land.delta = sfb_delta_of(t);
                issue_scales(land, 0);
                issue_prologue(t, 1);
                wait_landing_any<STREAM_A ? A_ITERS : A_ITERS + B_ITERS, MS>(land);
                }
            }
            raw_barrier();
            if (upper_half)
                raw_barrier();                      // the upper half runs one segment behind from here on

            int a_cur = 0, a_fill = 2 * A_BYTES, b_cur = 0;     // slots of A(kb), A(kb+2) [= A(kb-1)'s], B(kb)
            [[maybe_unused]] int b_fill = 2 * B_BYTES;          // MERGED: slot of B(kb+2) [= B(kb-1)'s]
            v8i bf[N224 ? 4 : NS], af[N224 ? 1 : HS];
            if (p.dbg != nullptr) t_loop0 = DG_STAMP_CLOCK();

            [[maybe_unused]] float tailp[DEPTH][4];            // PC: scale products of the last DEPTH steps of the previous K block
            #pragma unroll
            for (int i = 0; i < DEPTH; ++i)
                #pragma unroll
                for (int r = 0; r < 4; ++r)
                    tailp[i][r] = 0.f;
            if constexpr (N224) {
                // The four-segment schedule of the 256-row tile (see below) on the 64 x 112 wave tile, cut along N instead of M so that the
                // fragments fit the register file: M_a = all four A fragments x N-subtiles 0 .. 3 (16 steps), M_b = the same A fragments x
                // subtiles 4 .. 6 (12 steps; their B fragments take the registers of subtiles 0 .. 2).  The scale of a step is the product of
                // its M-subtile and its column GROUP.  Every accumulator still sees its K blocks in order: same bits as the 256 x 256 tile.
                auto grp = [](int ns) { return ns >> 1; };                          // (ns = 6 -> group 3)
                auto s_ms = [](int i) { return i < 16 ? i / 4 : (i - 16) / 3; };
                auto s_ns = [](int i) { return i < 16 ? (((i / 4) & 1) ? 3 - i % 4 : i % 4) : 4 + ((((i - 16) / 3) & 1) ? 2 - (i - 16) % 3 : (i - 16) % 3); };
                v8i af4[MS];
                for (int kb = 0; kb < nkb; ++kb) {
                    const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                    const uint8_t* b_tile = lds + B_BASE + b_cur + (wn * WN) * 128;
                    // ---------------- L_a ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int ns = 0; ns < 4; ++ns)
                        bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                    #pragma unroll
                    for (int h = 0; h < MS; ++h)
                        af4[h] = load_fragment(a_tile + h * 2048, frag_off);
                    tailg[2] = scaleg[MS - 1][2];
                    tailg[3] = scaleg[MS - 1][3];
                    {
                        float sbg[4];
                        #pragma unroll
                        for (int g = 0; g < 4; ++g)
                            sbg[g] = second[g] ? land.sb1 : land.sb;
                        #pragma unroll
                        for (int ms = 0; ms < MS; ++ms)
                            #pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                scaleg[ms][g] = landed_sfa<MS>(land, ms) * sbg[g];
                                pin_vgpr(scaleg[ms][g]);
                            }
                    }
                    issue_scales(land, kb + 1);
                    #pragma unroll
                    for (int q = 0; q < A_EARLY; ++q)
                        issue_a_piece(a_fill, kb + 2, q);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    // ---------------- M_a ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;
                        const float jscale = (i >= DEPTH) ? scaleg[s_ms(j)][grp(s_ns(j))] : tailg[grp(s_ns(j))];
                        mfma_promote_step(part[i & DEPTH], bf[s_ns(i)], af4[s_ms(i)], acc[s_ms(j)][s_ns(j)], jscale, part[(i + 1) & DEPTH]);
                    }
                    // ---------------- L_b ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int ns = 0; ns < 3; ++ns)
                        bf[ns] = load_fragment(b_tile + (4 + ns) * 2048, frag_off);
                    #pragma unroll
                    for (int q = A_EARLY; q < A_ITERS; ++q)
                        issue_a_piece(a_fill, kb + 2, q);
                    #pragma unroll
                    for (int q = 0; q < B_ITERS; ++q)
                        issue_b_piece(b_cur, kb + 2, q);
                    wait_landing_any<A_ITERS + B_ITERS, MS>(land);
                    #pragma unroll
                    for (int ns = 0; ns < 3; ++ns)
                        asm volatile("" : "+v"(bf[ns]) :: "memory");
                    // ---------------- M_b ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int i = 16; i < TOTAL; ++i) {
                        const int j = i - DEPTH;
                        mfma_promote_step(part[i & DEPTH], bf[s_ns(i) - 4], af4[s_ms(i)], acc[s_ms(j)][s_ns(j)], scaleg[s_ms(j)][grp(s_ns(j))], part[(i + 1) & DEPTH]);
                    }
                    b_cur ^= B_BYTES;
                    const int a_next = (a_cur == (A_SLOTS - 1) * A_BYTES) ? 0 : a_cur + A_BYTES;
                    a_fill = a_cur;
                    a_cur = a_next;
                }
            } else
            if constexpr (PC) {
                // The MERGED schedule with one scale per row of A AND per row of B (reference: impls/sm90_fp8_gemm_1d1d.cuh:279-311): a step
                // is MFMA + 4 v_mul (sfa x sfb) + 4 v_fmac.  The landing registers alternate between two sets (no copies): the loop body
                // exists twice, for "block kb in set 0 / block kb+1 landing in set 1" and the reverse.
                auto k_block = [&](ScaleLandingPC& cur, ScaleLandingPC& nxt, int kb) {
                    const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                    const uint8_t* b_tile = lds + B_BASE + b_cur + (wn * WN) * 128;
                    // ---------------- L ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                    #pragma unroll
                    for (int h = 0; h < MS; ++h)
                        af[h] = load_fragment(a_tile + h * 2048, frag_off);
                    {
                        const int jj = kb0 + imin(kb + 1, nkb - 1);   // past the end: the last block's scales again (never consumed)
                        issue_scale_loads_pc(nxt, sfa_rsrc, sfa_voff + jj * sfa_kb_stride, sfb_rsrc, sfb_lane_off + jj * sfb_kb_stride);
                    }
                    #pragma unroll
                    for (int q = 0; q < A_ITERS; ++q)
                        issue_a_piece(a_fill, kb + 2, q);
                    #pragma unroll
                    for (int q = 0; q < B_ITERS; ++q)
                        issue_b_piece(b_fill, kb + 2, q);
                    // my pieces of block kb+1 (issued one K block ago) have landed; this segment's own loads stay in flight
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(A_ITERS + B_ITERS + SCALE_LOADS, 0));
                    asm volatile("" ::: "memory");
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        asm volatile("" : "+v"(bf[ns]) :: "memory");
                    #pragma unroll
                    for (int h = 0; h < MS; ++h)
                        asm volatile("" : "+v"(af[h]) :: "memory");
                    // ---------------- M ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int i = 0; i < TOTAL; ++i) {
                        const int ns = i % NS, h = i / NS;
                        const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;
                        if (i < DEPTH)
                            mfma_promote_step_v(part[i & DEPTH], bf[ns], af[h], acc[j / NS][j % NS], tailp[i], part[(i + 1) & DEPTH]);
                        else
                            mfma_promote_step_pc_scalar(part[i & DEPTH], bf[ns], af[h], acc[j / NS][j % NS], cur.sb[j % NS], cur.sa[j / NS],
                                                        part[(i + 1) & DEPTH]);
                    }
                    // the products the next block's first DEPTH steps (or the drain after the loop) promote this block's last steps with
                    #pragma unroll
                    for (int i = 0; i < DEPTH; ++i) {
                        const int j = TOTAL - DEPTH + i;
                        #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            tailp[i][r] = cur.sa[j / NS] * cur.sb[j % NS][r];
                            pin_vgpr(tailp[i][r]);
                        }
                    }
                    wait_landing_pc<A_ITERS + B_ITERS>(nxt);          // the scales of block kb+1, in front of the back edge
                    const int b_next = (b_cur == (B_SLOTS - 1) * B_BYTES) ? 0 : b_cur + B_BYTES;
                    b_fill = b_cur;
                    b_cur = b_next;
                    const int a_next = (a_cur == (A_SLOTS - 1) * A_BYTES) ? 0 : a_cur + A_BYTES;
                    a_fill = a_cur;
                    a_cur = a_next;
                };
                int kb = 0;
                for (; kb + 2 <= nkb; kb += 2) {
                    k_block(land_pc0, land_pc1, kb);
                    k_block(land_pc1, land_pc0, kb + 1);
                }
                if (kb < nkb)
                    k_block(land_pc0, land_pc1, kb);
            } else
            if constexpr (STREAM_A) {
                // One copy of the K loop per wave half, chosen once: the halves differ in which pieces they issue and in two wait
                // counts (immediates), and no branch may sit between an asm scale load and its wait (the landing-register rule).
                auto k_loop = [&](auto upper_tag) {
                    constexpr bool UPPER = decltype(upper_tag)::value;
                    for (int kb = 0; kb < nkb; ++kb) {
                        const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                        const uint8_t* b_tile = lds + B_BASE + b_cur + (wn * WN) * 128;
                        // ---------------- L ----------------
                        raw_barrier();
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                        #pragma unroll
                        for (int h = 0; h < 2; ++h)
                            af[h] = load_fragment(a_tile + h * 2048, frag_off);
                        scale_tail = scale[MS - 1];
                        #pragma unroll
                        for (int ms = 0; ms < MS; ++ms) {
                            scale[ms] = landed_sfa<MS>(land, ms) * land.sb;
                            pin_vgpr(scale[ms]);
                        }
                        issue_scales(land, kb + 1);
                        // role split of the refills, 2 * A_ITERS = 2 * B_ITERS pieces per wave either way (see the STREAM_A note above)
                        if constexpr (UPPER) {
                            #pragma unroll
                            for (int q = 0; q < 2 * A_ITERS; ++q) {
                                const int unit = (wave - NW / 2) + (NW / 2) * q;          // 32 units over the 4 upper waves
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                    __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tm.a_base), 0, tm.a_bytes, 0x00020000),
                                    (__attribute__((address_space(3))) void*)(lds + a_fill + unit * 1024), 16,
                                    role_piece_voff[q], (kb0 + imin(kb + 2, nkb - 1)) * 128, 0, 0);
                            }
                        } else {
                            #pragma unroll
                            for (int q = 0; q < 2 * B_ITERS; ++q) {
                                const int unit = wave + (NW / 2) * q;                     // 32 units over the 4 lower waves
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                    __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tm.b_base), 0, tm.b_bytes, 0x00020000),
                                    (__attribute__((address_space(3))) void*)(lds + B_BASE + (b_cur ^ B_BYTES) + unit * 1024), 16,
                                    role_piece_voff[q], (kb0 + imin(kb + 1, nkb - 1)) * 128, 0, 0);
                            }
                        }
                        // upper half: my pieces of A(kb+1) -- issued one K block ago, read by the lower half from its next L on -- have
                        // landed (this segment's scale loads and pieces stay in flight); lower half: nothing to certify here
                        asm volatile("" ::: "memory");
                        __builtin_amdgcn_s_waitcnt(waitcnt_imm(UPPER ? 2 * A_ITERS + SCALE_LOADS : 63, 0));
                        asm volatile("" ::: "memory");
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            asm volatile("" : "+v"(bf[ns]) :: "memory");
                        #pragma unroll
                        for (int h = 0; h < 2; ++h)
                            asm volatile("" : "+v"(af[h]) :: "memory");

                        // ---------------- M ----------------
                        raw_barrier();
                        #pragma unroll
                        for (int i = 0; i < TOTAL; ++i) {
                            const int ns = i % NS, h = i / NS;
                            // the fragment of row h + 2 replaces the one row h - 2 used (its MFMAs were issued >= NS steps ago)
                            if (ns == 0 && h + 2 < MS)
                                af[(h + 2) & 3] = load_fragment(a_tile + (h + 2) * 2048, frag_off);
                            const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;
                            const float jscale = (i >= DEPTH) ? scale[j / NS] : scale_tail;
                            mfma_promote_step(part[i & DEPTH], bf[ns], af[h & 3], acc[j / NS][j % NS], jscale, part[(i + 1) & DEPTH]);
                        }
                        // in front of the back edge: the scales of block kb+1 (both halves), and for the lower half its pieces of B(kb+1),
                        // which everybody reads from the next L on (issued a whole matrix segment ago)
                        wait_landing_any<UPPER ? 2 * A_ITERS : 0, MS>(land);
                        b_cur ^= B_BYTES;
                        const int a_next = (a_cur == (A_SLOTS - 1) * A_BYTES) ? 0 : a_cur + A_BYTES;
                        a_fill = a_cur;
                        a_cur = a_next;
                    }
                };
                if (upper_half)
                    k_loop(std::true_type{});
                else
                    k_loop(std::false_type{});
            } else
            for (int kb = 0; kb < nkb; ++kb) {
                const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                const uint8_t* b_tile = lds + B_BASE + b_cur + (wn * WN) * 128;

                if constexpr (STREAM_A) {
                    // (own loop below: one copy per wave half)
                } else if constexpr (MERGED) {
                    // ---------------- L ----------------
                    raw_barrier();
                    [[maybe_unused]] FragTr bfq[NS];
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        if constexpr (B_MN)
                            bfq[ns] = load_fragment_tr(lds + B_BASE + b_cur, tr_lane_base, ((wn * (WN / 16) + ns) ^ tr_swz) << 4);
                        else
                            bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                    }
                    #pragma unroll
                    for (int h = 0; h < MS; ++h)
                        af[h] = load_fragment(a_tile + h * 2048, frag_off);
                    scale_tail = scale[MS - 1];
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        scale[ms] = landed_sfa<MS>(land, ms) * land.sb;
                        pin_vgpr(scale[ms]);
                    }
                    issue_scales(land, kb + 1);
                    #pragma unroll
                    for (int q = 0; q < A_ITERS; ++q)
                        issue_a_piece(a_fill, kb + 2, q);
                    #pragma unroll
                    for (int q = 0; q < B_ITERS; ++q)
                        issue_b_piece(b_fill, kb + 2, q);
                    // my pieces of block kb+1 (issued one K block ago) have landed; this segment's own loads stay in flight
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(waitcnt_imm(A_ITERS + B_ITERS + SCALE_LOADS, 0));
                    asm volatile("" ::: "memory");
                    if constexpr (B_MN) {
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            bf[ns] = assemble_fragment_tr(bfq[ns]);
                    } else {
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            asm volatile("" : "+v"(bf[ns]) :: "memory");
                    }
                    #pragma unroll
                    for (int h = 0; h < MS; ++h)
                        asm volatile("" : "+v"(af[h]) :: "memory");

                    // ---------------- M ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int i = 0; i < TOTAL; ++i) {
                        const int ns = DUO_NS(i), h = i / NS;
                        const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;
                        const float jscale = (i >= DEPTH) ? scale[j / NS] : scale_tail;
                        mfma_promote_step(part[i & DEPTH], bf[ns], af[h], acc[j / NS][DUO_NS(j)], jscale, part[(i + 1) & DEPTH]);
                    }
                    wait_landing_any<A_ITERS + B_ITERS, MS>(land);      // the scales of block kb+1, in front of the back edge
                    const int b_next = (b_cur == (B_SLOTS - 1) * B_BYTES) ? 0 : b_cur + B_BYTES;
                    b_fill = b_cur;
                    b_cur = b_next;
                } else {
                    // ---------------- L_a ----------------
                    raw_barrier();
                    // fragment reads first: they complete in the shadow of the slow vector-memory issue that follows
                    [[maybe_unused]] FragTr bfq[NS];
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        if constexpr (B_MN)
                            bfq[ns] = load_fragment_tr(lds + B_BASE + b_cur, tr_lane_base, ((wn * (WN / 16) + ns) ^ tr_swz) << 4);
                        else
                            bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                    }
                    [[maybe_unused]] FragTr afq[HS];
                    #pragma unroll
                    for (int h = 0; h < HS; ++h) {
                        if constexpr (A_MN)
                            afq[h] = load_fragment_tr(lds + a_cur, tr_lane_base, ((wm * MS + h) ^ tr_swz) << 4);
                        else
                            af[h] = load_fragment(a_tile + h * 2048, frag_off);
                    }
                    scale_tail = scale[MS - 1];
                    // block kb's scales landed before the previous L_b's wait (block 0: before the prologue's / the prefetch's)
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        scale[ms] = landed_sfa<MS>(land, ms) * land.sb;
                        pin_vgpr(scale[ms]);
                    }
                    issue_scales(land, kb + 1);
                    #pragma unroll
                    for (int q = 0; q < A_EARLY; ++q)
                        issue_a_piece(a_fill, kb + 2, q);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if constexpr (B_MN) {
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            bf[ns] = assemble_fragment_tr(bfq[ns]);
                    }
                    if constexpr (A_MN) {
                        #pragma unroll
                        for (int h = 0; h < HS; ++h)
                            af[h] = assemble_fragment_tr(afq[h]);
                    }

                    // ---------------- M_a ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int i = 0; i < SEG; ++i) {
                        const int ns = DUO_NS(i), h = i / NS;
                        const int j = (i >= DEPTH) ? i - DEPTH : TOTAL - DEPTH + i;
                        const float jscale = (i >= DEPTH) ? scale[j / NS] : scale_tail;
                        mfma_promote_step(part[i & DEPTH], bf[ns], af[h], acc[j / NS][DUO_NS(j)], jscale, part[(i + 1) & DEPTH]);
                    }

                    // ---------------- L_b ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int h = 0; h < HS; ++h) {
                        if constexpr (A_MN)
                            afq[h] = load_fragment_tr(lds + a_cur, tr_lane_base, ((wm * MS + HS + h) ^ tr_swz) << 4);
                        else
                            af[h] = load_fragment(a_tile + (HS + h) * 2048, frag_off);
                    }
                    #pragma unroll
                    for (int q = A_EARLY; q < A_ITERS; ++q)
                        issue_a_piece(a_fill, kb + 2, q);
                    #pragma unroll
                    for (int q = 0; q < B_ITERS; ++q)
                        issue_b_piece(b_cur, kb + 2, q);
                    // Block kb+1 and its scales: my pieces have landed.  (Persistent launch: a predecessor tile's output stores may
                    // still be pending in the first K block.  They count towards vmcnt too, which can only make this wait
                    // stricter -- loads retire in order among themselves, so "at most 8 operations outstanding" still implies
                    // "every load but the newest 8 has landed".)
                    wait_landing_any<A_ITERS + B_ITERS, MS>(land);
                    #pragma unroll
                    for (int h = 0; h < HS; ++h) {
                        if constexpr (A_MN)
                            af[h] = assemble_fragment_tr(afq[h]);
                        else
                            asm volatile("" : "+v"(af[h]) :: "memory");
                    }

                    // ---------------- M_b ----------------
                    raw_barrier();
                    #pragma unroll
                    for (int i2 = 0; i2 < SEG; ++i2) {
                        const int i = SEG + i2;
                        const int ns = DUO_NS(i), h = i2 / NS;
                        const int j = i - DEPTH;
                        mfma_promote_step(part[i & DEPTH], bf[ns], af[h], acc[j / NS][DUO_NS(j)], scale[j / NS], part[(i + 1) & DEPTH]);
                    }

                    b_cur ^= B_BYTES;
                }
                const int a_next = (a_cur == (A_SLOTS - 1) * A_BYTES) ? 0 : a_cur + A_BYTES;
                a_fill = a_cur;             // A(kb+3) will take the slot block kb just finished with
                a_cur = a_next;
            }
            if (!upper_half)
                raw_barrier();              // pairs with the barrier in front of the upper half's last segment
#ifndef DG_STAMP_ISSUE
            if (p.dbg != nullptr) t_loop1 = DG_STAMP_CLOCK();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the tail's re-read pieces: the ring is about to be reused
            __syncthreads();                                    // every wave is done with the LDS
            auto promote_pending = [&]() {
                #pragma unroll
                for (int i = 0; i < DEPTH; ++i) {
                    const int j = TOTAL - DEPTH + i;
                    if constexpr (PC)
                        promote_only_v(acc[j / NS][j % NS], tailp[i], part[(TOTAL + i + 1) & DEPTH]);
                    else if constexpr (N224)       // steps 25 .. 27 of the N-cut order: M-subtile 3, N-subtiles 6, 5, 4
                        promote_only(acc[MS - 1][6 - i], scaleg[MS - 1][(6 - i) >> 1], part[(TOTAL + i + 1) & DEPTH]);
                    else
                        promote_only(acc[j / NS][STREAM_A ? j % NS : DUO_NS(j)], scale[MS - 1], part[(TOTAL + i + 1) & DEPTH]);
                }
            };
            if constexpr (K_TAIL) {
                promote_pending();                              // block order of the promotion: the loop's last steps first
                if (k_tail != 0) {
                    // K not a multiple of 128 (dgrad shapes such as K = 2112, 576): the partial last block, once per tile, outside the
                    // tuned loop and without its role split.  K-major operands: the 16-byte chunks at and beyond k_tail get an offset
                    // that fails the buffer range check -- an LDS-DMA lane that is out of range writes ZEROS into the LDS (probed:
                    // tools/ubench/lds_dma_oob_probe.hip), so neither the next row's bytes nor row padding reach the matrix core.
                    // MN-major operands: k-rows >= K lie beyond the descriptor's extent by themselves.
                    const int tail_bias = (src_chunk * 16 >= k_tail) ? 0x40000000 : 0;
                    #pragma unroll
                    for (int q = 0; q < A_ITERS; ++q) {
                        const int unit = wave + NW * q;
                        if constexpr (M0S_A)
                            DG_LDS_DMA_PIECE_SUB(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tm.a_base) - M0_SHARE_BIAS, 0, tm.a_bytes + M0_SHARE_BIAS, 0x00020000),
                                 lds + (wave * A_ITERS + (q & ~3)) * 1024, a_piece_voff[q] + tail_bias, num_kb * 128, q, 0);
                        else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tm.a_base), 0, tm.a_bytes, 0x00020000),
                            (__attribute__((address_space(3))) void*)(lds + unit * 1024), 16,
                            A_MN ? amn_voff : a_piece_voff[q] + tail_bias,
                            A_MN ? (num_kb * 128 + 4 * unit) * lda_mn : num_kb * 128, 0, 0);
                    }
                    #pragma unroll
                    for (int q = 0; q < B_ITERS; ++q) {
                        const int unit = wave + NW * q;
                        if constexpr (M0S_B)
                            DG_LDS_DMA_PIECE_SUB(__builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tm.b_base) - M0_SHARE_BIAS, 0, tm.b_bytes + M0_SHARE_BIAS, 0x00020000),
                                 lds + B_BASE + (wave * B_ITERS + (q & ~3)) * 1024, b_piece_voff[q] + tail_bias, num_kb * 128, q, 0);
                        else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tm.b_base), 0, tm.b_bytes, 0x00020000),
                            (__attribute__((address_space(3))) void*)(lds + B_BASE + unit * 1024), 16,
                            B_MN ? bmn_voff : b_piece_voff[q] + tail_bias,
                            B_MN ? (num_kb * 128 + 4 * unit) * ldb_mn : num_kb * 128, 0, 0);
                    }
                    issue_scale_loads_any<MS>(land, sfa_rsrc, sfa_voff + num_kb * sfa_kb_stride, sfb_rsrc, num_kb * sfb_kb_stride);
                    wait_landing_any<0, MS>(land);              // straight-line from the loads; also lands the pieces
                    __syncthreads();
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        scale[ms] = landed_sfa<MS>(land, ms) * land.sb;
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        if constexpr (B_MN) {
                            FragTr fq = load_fragment_tr(lds + B_BASE, tr_lane_base, ((wn * (WN / 16) + ns) ^ tr_swz) << 4);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            bf[ns] = assemble_fragment_tr(fq);
                        } else {
                            bf[ns] = load_fragment(lds + B_BASE + (wn * WN) * 128 + ns * 2048, frag_off);
                        }
                    }
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        __builtin_amdgcn_sched_barrier(0);      // one subtile row at a time: hoisted fragment loads would cost 64 registers
                        v8i a_frag;
                        if constexpr (A_MN) {
                            FragTr fq = load_fragment_tr(lds, tr_lane_base, ((wm * MS + ms) ^ tr_swz) << 4);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            a_frag = assemble_fragment_tr(fq);
                        } else {
                            a_frag = load_fragment(lds + (wm * WM) * 128 + ms * 2048, frag_off);
                        }
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns) {
                            const v4f pr = mfma_fp8_k128(bf[ns], a_frag);
                            #pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[ms][ns][r] = __builtin_fmaf(scale[ms], pr[r], acc[ms][ns][r]);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();                            // the LDS is free again (the next tile's prologue may fly)
                }
                fetch_next();
            } else {
                fetch_next();                                   // persistent launch: the next tile's prologue flies from here
                promote_pending();
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
        bool store = true;
        if constexpr (SPLITK) {
            if (piece.split && !(t.m_end > t.m0)) {
                store = piece.index == 0;               // an all-padding tile: its zero rows are written once
            } else if (piece.split && p.sk_exchange != 0 && piece.factor == 2) {
                // two pieces, exchanged inside the kernel (GemmParams::sk_exchange)
                unsigned* flag = static_cast<unsigned*>(p.sk_workspace) + piece.tail;
                uint8_t* slab_bytes = static_cast<uint8_t*>(p.sk_workspace) + 4096 + static_cast<int64_t>(piece.tail) * (BM * BN * 4);
                const auto slab = __builtin_amdgcn_make_buffer_rsrc(slab_bytes, 0, BM * BN * 4, 0x00020000);
                const int lane_off = (wave * 64 + lane) * 16;
                if (piece.index == 0) {
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, out[ms][ns]), slab, lane_off, (ms * NS + ns) * (NW * 1024),
                                                                   DG_SK_XCHG_AUX);       // sc0 sc1: written through, past the (per-XCD) L2
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();                    // every wave's partial stores are acknowledged ...
                    if (threadIdx.x == 0)               // ... before the flag goes out (no cache-wide writeback: nothing of this sits in a cache)
                        __hip_atomic_store(flag, p.sk_exchange, DG_SK_XCHG_ORDER_REL, __HIP_MEMORY_SCOPE_AGENT);
                    store = false;
                } else {
                    if (threadIdx.x == 0) {
                        // the producer was dispatched before this workgroup (its work item precedes this one by the tile count): it is
                        // resident or done.  Bounded all the same: a lost flag must end in a wrong tile, not in a hung device.
                        int spins = 0;
                        while (__hip_atomic_load(flag, DG_SK_XCHG_ORDER_ACQ, __HIP_MEMORY_SCOPE_AGENT) != p.sk_exchange && ++spins < (1 << 22))
                            __builtin_amdgcn_s_sleep(8);
                    }
                    __syncthreads();
                    if (DG_SK_XCHG_AUX == 0)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns) {
                            const v4f part0 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(slab, lane_off, (ms * NS + ns) * (NW * 1024), DG_SK_XCHG_AUX));
                            out[ms][ns] = part0 + out[ms][ns];          // piece order, as dg_split_k_reduce_kernel sums
                        }
                    __syncthreads();                    // every wave has read the partial before the flag is cleared for the next launch
                    if (threadIdx.x == 0)
                        __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else if (piece.split) {
                // FP32 partial tile of this K piece: lane-linear 16-byte stores, [subtile][wave][lane], ordinary (write-back) stores.
                // The sum over the pieces and the output stores belong to dg_split_k_reduce_kernel, launched behind this kernel on
                // the same stream: the kernel boundary is the only synchronisation (no counters, no spinning, nothing written
                // through), and the reduction runs on every CU instead of on the last arriver of each tile alone.
                uint8_t* slab_bytes = static_cast<uint8_t*>(p.sk_workspace) + 4096 +
                                      static_cast<int64_t>(piece.tail) * piece.factor * (BM * BN * 4);
                const auto slab = __builtin_amdgcn_make_buffer_rsrc(slab_bytes, 0, piece.factor * (BM * BN * 4), 0x00020000);
                const int lane_off = (wave * 64 + lane) * 16;
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, out[ms][ns]), slab,
                                                               lane_off, piece.index * (BM * BN * 4) + (ms * NS + ns) * (NW * 1024), 0);
                store = false;
            }
        }
        if constexpr (N224) {
            // subtiles 0 .. 3: the 64-column full-line stores; 4, 5: one permuted pair (32 columns: 16 bytes per lane); 6: natural order (16
            // columns: 8 bytes per lane).  The common case -- BF16, no accumulation, aligned rows, the wave tile inside N -- inline and lean;
            // everything else through the shared epilogue.
            const int m_base = t.m0 + wm * WM, n_base = t.n0 + wn * WN;
            if (p.d_dtype == 0 && !p.accumulate && p.d_vec_ok && n_base + WN <= p.n && p.head_lr == 0) {
                const int lg = lane >> 4;
                uint16_t* dbase = reinterpret_cast<uint16_t*>(p.d) + ad_group * p.d_sg;
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    const v4f quad4[4] = {out[ms][0], out[ms][1], out[ms][2], out[ms][3]};
                    store_rows_full_line<MS, true>(p, t, ad_group * p.d_sg, quad4, ms, m_base, n_base);
                    const int row = m_base + (lane & 15) * MS + ms;
                    const bool compute_row = row >= t.m_begin && row < t.m_end, zero_row = row >= t.zero_from && row < t.zero_to;
                    if (compute_row || zero_row) {
                        uint16_t* drow = dbase + static_cast<int64_t>(row) * p.d_sm + n_base;
                        const uint4 pair = zero_row ? make_uint4(0u, 0u, 0u, 0u)
                                                    : make_uint4(pack_bf16(out[ms][4][0], out[ms][4][1]), pack_bf16(out[ms][4][2], out[ms][4][3]),
                                                                 pack_bf16(out[ms][5][0], out[ms][5][1]), pack_bf16(out[ms][5][2], out[ms][5][3]));
                        *reinterpret_cast<uint4*>(drow + 64 + lg * 8) = pair;
                        *reinterpret_cast<uint2*>(drow + 96 + lg * 4) = zero_row ? make_uint2(0u, 0u)
                                                                                 : make_uint2(pack_bf16(out[ms][6][0], out[ms][6][1]), pack_bf16(out[ms][6][2], out[ms][6][3]));
                    }
                }
            } else {
                v4f out4[MS][4], out2[MS][2], out1[MS][1];
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    #pragma unroll
                    for (int ns = 0; ns < 4; ++ns)
                        out4[ms][ns] = out[ms][ns];
                    out2[ms][0] = out[ms][4];
                    out2[ms][1] = out[ms][5];
                    out1[ms][0] = out[ms][6];
                }
                store_tile<MS, 4, true>(p, t, ad_group * p.d_sg, out4, m_base, n_base);
                store_tile<MS, 2, true>(p, t, ad_group * p.d_sg, out2, m_base, n_base + 64);
                store_tile<MS, 1, true, false, true>(p, t, ad_group * p.d_sg, out1, m_base, n_base + 96);
            }
        } else
        if (store)
            store_tile<MS, NS, !A_MN, false, B_MN, PC>(p, t, ad_group * p.d_sg, out, t.m0 + wm * WM, t.n0 + wn * WN);
        if (p.dbg != nullptr && first_tile && !next_prefetched) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_stamp(p, NW, 0, t_entry);
            dbg_stamp(p, NW, 1, t_loop0);
            dbg_stamp(p, NW, 2, t_loop1);
            dbg_stamp(p, NW, 3, DG_STAMP_CLOCK());
        }
        t = tn;
        piece = piece_next;
        prefetched = next_prefetched;
        first_tile = false;
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool PERSIST = false, bool B_MN = false, bool SPLITK = false, bool A_MN = false,
          bool K_TAIL = false, bool MERGED = false, bool STREAM_A = false, bool PC = false, bool SFA_RM = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_duo_kernel(const GemmParams p) {
    __shared__ __attribute__((aligned(1024))) uint8_t lds[duo_lds_bytes(BM, BN, MERGED)];
    duo_kernel_body<BM, BN, WAVES_M, WAVES_N, PERSIST, B_MN, SPLITK, A_MN, K_TAIL, MERGED, STREAM_A, PC, SFA_RM>(p, lds);
}

// The two GEMM launches of launch_contiguous_tabled as ONE: every workgroup walks the 256-row tiles of the in-kernel tile list (q) and then
// the K-split pieces of the 128-row remainders (r) -- no kernel boundary between the two (drain, launch latency, a second evaluation of the
// tile list against a cold layout read), and a workgroup whose 256-row walk ends early starts on its piece at once.  Both bodies use the
// same LDS bytes; the barrier between them is the only coupling.  The reduction over the pieces stays a separate launch.
// (CALLER = 1: the host pass of hipcc accepts ONE reference per body specialization in a translation unit -- the body's gfx950-only
// builtins make its host-side instantiation invalid, the first reference defers that, a second one from another kernel is answered with an
// unexplained substitution failure or a silently missing launch stub.  The tag makes these two their own specializations.)
template <int BIG_BM, int REM_BM, int BN, int WAVES_M, int WAVES_N>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_duo_tab_fused_kernel(const GemmParams q, const GemmParams r) {
    static_assert(duo_lds_bytes(REM_BM, BN, true) <= duo_lds_bytes(BIG_BM, BN, false), "the remainder body fits in the 256-row body's LDS");
    __shared__ __attribute__((aligned(1024))) uint8_t lds[duo_lds_bytes(BIG_BM, BN, false)];
    // (round 6, negative: letting every second workgroup of an XCD -- or all of them -- walk its remainder pieces FIRST, so that the HBM-bound remainder
    //  phase and the compute-bound 256-row phase overlap across the chip: C4 155.6-157.1 / 153.4-154.9 us against 151.2-152.5 in this order,
    //  profiles/r06_probe/c4_remainder_first_interleave_negative.log -- and the loop over the two bodies cost the kernel 40 spilled registers)
    duo_kernel_body<BIG_BM, BN, WAVES_M, WAVES_N, true, false, false, false, false, false, false, false, false, 1>(q, lds);
    __syncthreads();
    duo_kernel_body<REM_BM, BN, WAVES_M, WAVES_N, true, false, true, false, false, true, false, false, false, 1>(r, lds);
}


// Second phase of the K split: one workgroup per (split tile, M-subtile row) -- sk_tiles x MS workgroups, the thread geometry of the
// duo kernel that wrote the partials -- sums the tile's sk_factor partial subtiles in piece order and stores them through the
// shared epilogue (padding rows, accumulation, FP32 / BF16, column maps as for an unsplit tile).  A piece-count-sized read per
// workgroup instead of sk_factor whole tiles on the last arriver of each tile.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool B_MN>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_split_k_reduce_kernel(const GemmParams p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int tail = blockIdx.x / MS, ms_mine = blockIdx.x % MS;
    MaskedWalk walk;
    int sk_first = p.sk_first_tile, sk_factor = p.sk_factor;
    if (p.table_mode != 0) {
        walk.table_mask = contiguous_tile_mask(p.layout, p.m, p.table_mode, &walk.block_group);
        walk.have_block_groups = true;
    }
    if (p.sk_exchange != 0)
        return;                                 // (the two pieces were summed inside the first kernel)
    if (table_launch(p)) {                      // table launch: the grid is an upper bound, tile count and pieces live on the device
        const int sk_tiles = table_count(p, walk) * p.num_n_tiles;
        sk_first = 0;
        sk_factor = table_pieces(p, sk_tiles);
        if (tail >= sk_tiles || sk_factor < 2)
            return;                             // (no such tile / the tiles were computed whole and stored by the first phase)
    }
    const Tile t = get_tile<BM, BN>(p, sk_first + tail, walk, 0);
    if (!t.valid || !(t.m_end > t.m0))
        return;                                 // (an all-padding tile: its zero rows were written by the first phase)
    uint8_t* slab_bytes = static_cast<uint8_t*>(p.sk_workspace) + 4096 + static_cast<int64_t>(tail) * sk_factor * (BM * BN * 4);
    const auto slab = __builtin_amdgcn_make_buffer_rsrc(slab_bytes, 0, sk_factor * (BM * BN * 4), 0x00020000);
    const int lane_off = (wave * 64 + lane) * 16;
    v4f sum[NS], nxt[NS];
    auto load_piece = [&](v4f (&dst)[NS], int s) {
        #pragma unroll
        for (int ns = 0; ns < NS; ++ns)
            dst[ns] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(
                slab, lane_off, s * (BM * BN * 4) + (ms_mine * NS + ns) * (NW * 1024), 0));
    };
    load_piece(sum, 0);
    for (int s = 1; s < sk_factor; ++s) {       // piece order: the sum does not depend on which piece finished when
        load_piece(nxt, s);
        #pragma unroll
        for (int ns = 0; ns < NS; ++ns)
            sum[ns] += nxt[ns];
    }
    auto store_row = [&](auto ms_c) {
        constexpr int ROW = decltype(ms_c)::value;
        v4f out[MS][NS];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                out[ms][ns] = ms == ROW ? sum[ns] : v4f{0.f, 0.f, 0.f, 0.f};
        store_tile<MS, NS, true, false, B_MN>(p, t, 0, out, t.m0 + wm * WM, t.n0 + wn * WN, ROW);
    };
    static_assert(MS == 4, "one case per M-subtile row of the 64-row wave tile");
    switch (ms_mine) {
        case 0: store_row(std::integral_constant<int, 0>{}); break;
        case 1: store_row(std::integral_constant<int, 1>{}); break;
        case 2: store_row(std::integral_constant<int, 2>{}); break;
        default: store_row(std::integral_constant<int, 3>{}); break;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Stream kernel: the HBM-bound end of the path (masked / decode-sized M, small dense M).  A 64 x 128 tile per
// 4-wave workgroup and a STAGES-deep LDS ring (6 x 24.5 KiB): each stage holds the A and B tiles AND the scales of one
// K block (SFA rows as one 256-byte LDS-DMA dword piece, SFB as a broadcast dword piece), so the K loop contains no
// VGPR-destination loads at all -- five K blocks (120 KiB per CU, 30 MB across the chip) stay in flight, one barrier
// per K block both certifies "block kb has landed" (after a counted vmcnt) and frees the slot of block kb-1.
// The matrix work (8 MFMAs per wave per K block) is far below the pipe's rate here: weights stream once from HBM.
// ---------------------------------------------------------------------------------------------------------------
// B_AUX: cache-policy bits of the weight stream's LDS-DMA (2 = nt: a weight byte is read by exactly one CU, once).
// KBS: K blocks per ring stage -- a wave requests KBS x 128 contiguous bytes of each row back to back, which is what
// gives the HBM controller row-buffer hits on K-major weights whose rows lie K bytes apart.
// E8: packed UE8M0 scales (one int32 word of four exponent bytes per row of A and per ROW of B per four K blocks; p.sfa / p.sfb hold the
// words, strides per K quad): every stage carries the words of its K block's quad (64 for A, BN for B), the block's byte is
// shifted down by VALU and the hardware-scaled MFMA accumulates in place -- the decode-sized form of the packed-UE8M0 path.
// LW: extra LOADER waves (round 4).  A stream tile is bound by the LDS-DMA issue rate of its workgroup (profiles/r03_fill/NOTES.md: 4 waves
// fill a CU at ~50 GB/s, 8 at ~80, 16 at ~96 whatever is in flight), not by its matrix work: the NW compute waves keep the tile's MFMA /
// epilogue geometry, LW more waves do nothing but issue their share of every stage's pieces and join the barriers.  The pieces of a stage
// (KBS blocks x (BM / 8 + BN / 8) units) are dealt round-robin over all NW + LW waves.
// KSPLIT (round 6): dense mid-M problems (64 < m <= 256) whose 64 x 128 tiles fill a quarter of the chip or less.  What bounds them is the
// L2 -> LDS rate of a CU, i.e. the bytes ONE workgroup pulls (m = 128, 4096 x 7168 on 64 x 32 tiles: 688 KB per CU); here a work item is
// (tile, K piece): p.sk_factor pieces per tile, piece q = K blocks [q kb / f, (q + 1) kb / f) -- 64 tiles x 4 pieces = 256 work items of
// 344 KB each.  Pieces 0 .. f - 2 write their FP32 partial tile (lane-linear, written through) into the caller's workspace and raise a flag
// (the launch's own epoch value, GemmParams::sk_exchange; the last piece takes the flags back so that a hipGraph replay -- same epoch -- waits again); the LAST piece of a tile -- dispatched after all the
// others, so they are resident or done: no deadlock whatever the residency -- waits for them (bounded) and adds the partials IN PIECE
// ORDER (((p0 + p1) + ...) + own: bit-repeatable), then stores the tile through the shared epilogue.
// Workspace: 4 KiB header | 32 KiB of flags ([tile][8]) | slabs [tile][8][BM * BN] FP32.
// G32 (with E8, round 6): scale granularity 32 along K (the MX recipe) at decode-sized M -- one packed word per row and 128-K block (byte g = MX
// block g of the block), every stage carries ITS block's words, and lane group g of the scaled MFMA shifts its own byte down (what
// quad_e8_kernel_body's G32 form does; probed in profiles/r06_probe/g32_scale_byte_mapping.log).  Before it these problems ran the 128-row
// four-wave tile: 68-70 us for m <= 256 at 4096 x 7168 against 25-38 for granularity 128, the masked C5 shape 80 against 43.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int B_AUX = 0, int KBS = 1, bool E8 = false, int LW = 0, bool KSPLIT = false, bool G32 = false>
__device__ __forceinline__ void stream_kernel_body(const GemmParams& p) {
    static_assert(!KSPLIT || (LW == 0 && B_AUX != 64), "KSPLIT: the plain stream tile (FP32 scales or packed words), no loader waves");
    static_assert(!G32 || E8, "G32: a form of the packed-scale stream tile");
    constexpr int NW = WAVES_M * WAVES_N, TW = NW + LW;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16;
    constexpr bool GSF = !E8, GSE = E8 && !G32, GSG = E8 && G32;           // FP32 scales / packed words per K quad / packed words per K block
    constexpr int SFB_PIECES = GSG ? (BN * 16 + 1023) / 1024 : E8 ? (BN + 63) / 64 : 1;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SFA_BYTES = GSE ? 256 : 1024;
    // FP32 scales (GSF): the scales do NOT ride in the stages.  An LDS-DMA instruction costs its wave ~57 (4 bytes per lane) to ~83 ns
    // (16 bytes per lane) whatever it moves (profiles/r03_fill/NOTES.md), and the two scale pieces per K block that every wave used to
    // issue were 114 of the 612 ns a wave spends per K block of a 64 x 128 tile (363 for 64 x 32).  Now ONE 16-byte-per-lane piece
    // carries the 64 row scales of FOUR K blocks (lane l: rows 4 (l & 15) .. + 3 of block 4 g + (l >> 4): the MN-major layout makes a
    // lane's four rows contiguous) and one 4-byte piece the four SFB values: 35 ns per K block.  They live in a ring of four group
    // slots behind the stages and are issued in front of the data pieces of the group's first block.
    // GSE (end of round 6): packed words of granularity 128 likewise -- a word covers a K QUAD, and every wave used to issue the quad's two or
    // three word pieces again with each of its four K blocks (8 of the 20 pieces a wave issues per stage of the 64 x 32 tile).  Now the words of
    // a quad land once, in the group ring (slot: 64 A words, then BN weight-row words), in front of the data pieces of the quad's first block
    // (same box: dense packed m = 128, 4096 x 7168 18.0 -> 16.5 us, profiles/r06_probe/e8_stream_group_words_ab.log).
    // GSG: granularity 32 -- a word per row and K BLOCK, laid out like the FP32 scales ([K block][row], rows contiguous): the A words of four K
    // blocks are the FP32 form's one 16-byte-per-lane piece, the weight-row words of four K blocks ([4 blocks][BN rows] in the slot) take
    // BN / 64 such pieces (lane l of piece r: bytes 1024 r + 16 l of that image; BN = 32: the upper half of the lanes is out of range).
    constexpr int BLOCK_BYTES = A_BYTES + B_BYTES;
    constexpr int STAGE_BYTES = KBS * BLOCK_BYTES;
    // (a piece writes all 64 lanes' bytes -- zeros for lanes that are out of range: a slot holds WHOLE pieces, 1 KiB each at 16 bytes per lane)
    constexpr int SFG_SLOT = GSG ? 1024 + 1024 * SFB_PIECES : 1024 + 256;
    // (two slots where the ring holds at most four K blocks -- the two-per-CU 64 x 128 tile: 72 + 6 KiB -- four otherwise)
    constexpr int SFG_SLOTS = (GSG && STAGES * KBS <= 4) ? 2 : 4;
    constexpr int SFG_OFF = STAGES * STAGE_BYTES;
    constexpr int LDS_BYTES = SFG_OFF + SFG_SLOTS * SFG_SLOT;
    static_assert(!GSE || SFA_BYTES + 256 * SFB_PIECES <= SFG_SLOT, "a group slot holds the quad's words of both operands");
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN / 8 / NW;
    constexpr bool NO_A = (B_AUX == 64);                       // timing experiment: the A tile is never loaded
    // per wave per stage; the group pieces (two or three per four K blocks) are NOT counted -- the counted waits then ask for up to three
    // more of the younger pieces than needed (stricter, never looser; the ring has STAGES - 2 stages of slack)
    constexpr int A_PER = KBS * (BM / 8) / TW, B_PER = KBS * (BN / 8) / TW;     // LW > 0: pieces per wave and STAGE
    static_assert(LW == 0 || (!NO_A && (KBS * (BM / 8)) % TW == 0 && (KBS * (BN / 8)) % TW == 0),
                  "loader waves: every wave issues the same number of A and of B pieces per stage");
    constexpr int PIECES = LW > 0 ? A_PER + B_PER : ((NO_A ? 0 : A_ITERS) + B_ITERS) * KBS;
    static_assert(STAGES * KBS <= 4 * (SFG_SLOTS - 1), "a group slot is refilled only after its last reader");
    static_assert(!E8 || MS == 4 || MS == 1, "packed-scale form: a lane reads its MS row words with one LDS read");
    constexpr unsigned OOB = 0x80000000u;
    static_assert(BM == 64 && (BN == 128 || BN == 64 || BN == 32), "one 256-byte SFA piece and one SFB value per tile");
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave issues the same number of pieces");
    static_assert((MS == 4 || MS == 1) && NS % 2 == 0, "a lane reads its MS row scales with one LDS read");
    static_assert((STAGES - 1) * PIECES < 64, "vmcnt is a 6-bit counter");
    static_assert(LDS_BYTES <= 160 * 1024 && STAGES >= 3, "LDS budget");
    static_assert((NW * 8) % 16 == 0 && ((NW * 8) % WN == 0 || WN % (NW * 8) == 0),
                  "the row permutation of a B piece must be lane-independent");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int num_kb_total = p.k / 128;
    int num_kb = num_kb_total, kb0 = 0;             // KSPLIT: the K blocks [kb0, kb0 + num_kb) of the work item's piece
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);
    // interleaved A rows (LDS row position ms * 16 + i holds tile row i * MS + ms), as in the duo kernel
    auto a_unit_row = [](int u) { return (u / (WM / 8)) * WM + (u & 1) * 8 * MS + ((u % (WM / 8)) >> 1); };
    const int a_voff = piece_row * MS * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;

    MaskedWalk walk;
    const int num_launched = gridDim.x;
    for (int tile_id = blockIdx.x;; tile_id += num_launched) {
        int tile = tile_id;
        [[maybe_unused]] int ks_piece = 0, ks_pieces = 1;
        if constexpr (KSPLIT) {
            const int tiles = p.num_m_tiles * p.num_n_tiles;
            ks_pieces = p.sk_factor;
            if (tile_id >= tiles * ks_pieces)
                break;
            tile = tile_id % tiles;
            ks_piece = tile_id / tiles;
            if constexpr (E8) {                     // packed words: pieces of whole K quads (a word of granularity 128 covers one; the group ring is filled per quad)
                const int quads = (num_kb_total + 3) / 4;
                kb0 = (ks_piece * quads / ks_pieces) * 4;
                num_kb = imin(num_kb_total, ((ks_piece + 1) * quads / ks_pieces) * 4) - kb0;
            } else {
                kb0 = ks_piece * num_kb_total / ks_pieces;
                num_kb = (ks_piece + 1) * num_kb_total / ks_pieces - kb0;
            }
        }
        const Tile t = get_tile<BM, BN>(p, tile, walk);
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
            // (every descriptor input through readfirstlane: round 4 found waterfall loops around the A and scale pieces of every stream kernel)
            const uint8_t* a_base = uniform_pointer(p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm);
            const uint8_t* b_base = uniform_pointer(p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn);
            const int a_rows = uniform_int(imin(t.m_end - t.m0, BM)), b_rows = uniform_int(imin(p.n - t.n0, BN));
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base), 0,
                                                                  (a_rows - 1) * lda + p.k, 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base), 0,
                                                                  (b_rows - 1) * ldb + p.k, 0x00020000);
            // SFA of the tile's rows: MN-major, rows m0 .. m0+63 are 256 contiguous bytes per K block
            const int sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kb_stride = static_cast<int>(p.sfb_sk) * 4;
            // (E8: the strides are per K quad and the "K block" index of a scale row is j >> 2)
            const int num_sf_k = GSE ? (num_kb_total + 3) / 4 : num_kb_total;               // (KSPLIT: a piece's blocks are addressed from the operands' first block)
            float* sfa_tile = uniform_pointer(const_cast<float*>(p.sfa) + ad_group * p.sfa_sg + t.m0);
            const int sfa_rows = uniform_int(imin(p.m - t.m0, BM));
            // (16-byte requests: the MN-major layouts -- FP32 and packed -- pad the rows to a multiple of four, so a request that starts below the row count is whole)
            const auto sfa_rsrc = __builtin_amdgcn_make_buffer_rsrc(sfa_tile, 0, (num_sf_k - 1) * sfa_kb_stride + (GSE ? sfa_rows : (sfa_rows + 3) / 4 * 4) * 4, 0x00020000);
            float* sfb_tile = uniform_pointer(const_cast<float*>(p.sfb) + static_cast<int64_t>(t.group) * p.sfb_sg +
                                              (E8 ? static_cast<int64_t>(t.n0) : static_cast<int64_t>(t.n0 / 128) * p.sfb_sn));
            const int sfb_rows = uniform_int(imin(p.n - t.n0, BN));
            const auto sfb_rsrc = __builtin_amdgcn_make_buffer_rsrc(sfb_tile, 0, (num_sf_k - 1) * sfb_kb_stride + (GSG ? (sfb_rows + 3) / 4 * 4 * 4 : GSE ? sfb_rows * 4 : 4),
                                                                  0x00020000);
            const int sfg_a_voff = (lane >> 4) * sfa_kb_stride + (lane & 15) * 16, sfg_b_voff = (lane & 3) * sfb_kb_stride;

            // All pieces of K block j into ring slot j % STAGES (slot_off in bytes).  Blocks past the end are issued as
            // out-of-range no-ops so that the vmcnt arithmetic stays exact.
            // the scales of K blocks j .. j + 3, j a multiple of four (blocks past the end: out of range, zeros), into the group ring
            // ONE wave per piece (end of round 6; every wave used to issue all of them -- identical destinations, identical data -- and they were
            // 2 of the 8 pieces a wave of the 64 x 32 tile with loader waves issues per stage): the last wave issues the A piece, the waves in front
            // of it the B pieces.  The issuing wave's counted wait covers its group piece (it is older than the stage's data pieces; the waits only
            // get stricter), the stage's barrier publishes it to the others.  -DDG_GROUP_SCALES_EVERY_WAVE: the form before (tuning build).
            auto issue_group_scales = [&](int j) {
                const unsigned oob = j < num_kb ? 0u : OOB;
                uint8_t* slot = lds + SFG_OFF + ((j >> 2) & (SFG_SLOTS - 1)) * SFG_SLOT;
#ifdef DG_GROUP_SCALES_EVERY_WAVE
                auto mine = [&](int) { return true; };
#else
                static_assert(TW >= 1 + SFB_PIECES, "a wave per group piece");
                auto mine = [&](int piece) { return wave == TW - 1 - piece; };
#endif
                if constexpr (GSG) {                           // a word per row and K block: the FP32 form's A piece, [4 blocks][BN rows] of weight-row words
                    if (mine(0))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            sfa_rsrc, (__attribute__((address_space(3))) void*)slot, 16,
                            static_cast<int>(static_cast<unsigned>(sfg_a_voff) | oob), (kb0 + j) * sfa_kb_stride, 0, 0);
                    #pragma unroll
                    for (int r = 0; r < SFB_PIECES; ++r)
                        if (mine(1 + r)) {
                            const int off = r * 1024 + lane * 16, blk = off / (BN * 4), within = off % (BN * 4);
                            const unsigned lane_oob = blk < 4 ? 0u : OOB;
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                sfb_rsrc, (__attribute__((address_space(3))) void*)(slot + 1024 + r * 1024), 16,
                                static_cast<int>(static_cast<unsigned>(blk * sfb_kb_stride + within) | oob | lane_oob), (kb0 + j) * sfb_kb_stride, 0, 0);
                        }
                } else if constexpr (GSF) {
                    if (mine(0))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            sfa_rsrc, (__attribute__((address_space(3))) void*)slot, 16,
                            static_cast<int>(static_cast<unsigned>(sfg_a_voff) | oob), (kb0 + j) * sfa_kb_stride, 0, 0);
                    if (mine(1))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            sfb_rsrc, (__attribute__((address_space(3))) void*)(slot + 1024), 4,
                            static_cast<int>(static_cast<unsigned>(sfg_b_voff) | oob), (kb0 + j) * sfb_kb_stride, 0, 0);
                } else {                                        // GSE: the quad's packed words (strides per K quad)
                    if (mine(0))
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            sfa_rsrc, (__attribute__((address_space(3))) void*)slot, 4,
                            static_cast<int>(static_cast<unsigned>(lane * 4 + ((kb0 + j) >> 2) * sfa_kb_stride) | oob), 0, 0, 0);
                    #pragma unroll
                    for (int r = 0; r < SFB_PIECES; ++r)        // the words of the tile's BN weight rows: 64 per piece
                        if (mine(1 + r))
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                sfb_rsrc, (__attribute__((address_space(3))) void*)(slot + SFA_BYTES + r * 256), 4,
                                static_cast<int>(static_cast<unsigned>((r * 64 + lane) * 4 + ((kb0 + j) >> 2) * sfb_kb_stride) | oob), 0, 0, 0);
                }
            };
            auto issue_block = [&](int slot_off, int j) {
                const unsigned oob = j < num_kb ? 0u : OOB;
                uint8_t* stage = lds + slot_off;
                if ((j & 3) == 0)
                    issue_group_scales(j);
                #pragma unroll
                for (int q = 0; q < (NO_A ? 0 : A_ITERS); ++q) {
                    const int unit = wave + NW * q;
                    const int voff = static_cast<int>(static_cast<unsigned>(a_voff) + (static_cast<unsigned>(a_unit_row(unit) * lda) | oob));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        a_rsrc, (__attribute__((address_space(3))) void*)(stage + unit * 1024), 16, voff, (kb0 + j) * 128, 0, 0);
                }
                #pragma unroll
                for (int q = 0; q < B_ITERS; ++q) {
                    const int unit = wave + NW * q;
                    const int voff = static_cast<int>(static_cast<unsigned>(b_voff) +
                                                      (static_cast<unsigned>(b_row_perm<WN>(q * (NW * 8)) * ldb) | oob));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        b_rsrc, (__attribute__((address_space(3))) void*)(stage + A_BYTES + unit * 1024), 16, voff, (kb0 + j) * 128, 0, B_AUX & 3);
                }
            };

            auto issue_stage = [&](int slot_off, int sb) {            // stage sb = K blocks sb * KBS .. + KBS - 1
                if constexpr (LW > 0) {
                    // the stage's pieces dealt over all TW waves: piece index wave + TW q -> (block, unit); the group scales as issue_block
                    #pragma unroll
                    for (int u = 0; u < KBS; ++u) {
                        const int j = sb * KBS + u;
                        if ((j & 3) == 0)
                            issue_group_scales(j);
                    }
                    #pragma unroll
                    for (int q = 0; q < A_PER; ++q) {
                        const int idx = wave + TW * q, u = idx / (BM / 8), unit = idx % (BM / 8), j = sb * KBS + u;
                        const unsigned oob = j < num_kb ? 0u : OOB;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            a_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + u * BLOCK_BYTES + unit * 1024), 16,
                            static_cast<int>(static_cast<unsigned>(a_voff) + (static_cast<unsigned>(a_unit_row(unit) * lda) | oob)), (kb0 + j) * 128, 0, 0);
                    }
                    #pragma unroll
                    for (int q = 0; q < B_PER; ++q) {
                        const int idx = wave + TW * q, u = idx / (BN / 8), unit = idx % (BN / 8), j = sb * KBS + u;
                        const unsigned oob = j < num_kb ? 0u : OOB;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            b_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + u * BLOCK_BYTES + A_BYTES + unit * 1024), 16,
                            static_cast<int>(static_cast<unsigned>(b_row_perm<WN>(unit * 8 + piece_row) * ldb + src_chunk * 16) | oob), (kb0 + j) * 128, 0,
                            B_AUX & 3);
                    }
                    return;
                }
                #pragma unroll
                for (int u = 0; u < KBS; ++u)
                    issue_block(slot_off + u * BLOCK_BYTES, sb * KBS + u);
            };
            #pragma unroll
            for (int j = 0; j < STAGES - 1; ++j)
                issue_stage(j * STAGE_BYTES, j);

            int cur = 0, fill = (STAGES - 1) * STAGE_BYTES;
            const int num_sb = (num_kb + KBS - 1) / KBS;
            for (int sb = 0; sb < num_sb; ++sb) {
                // stage sb: my pieces have landed (the STAGES-2 younger stages may still fly); then everybody's have,
                // and everybody is done reading stage sb-1, whose slot takes stage sb+STAGES-1
                asm volatile("s_waitcnt vmcnt(%c0)" :: "i"((STAGES - 2) * PIECES) : "memory");
                raw_barrier();
                issue_stage(fill, sb + STAGES - 1);
              bool stage_done = false;
              if constexpr (GSF && KBS > 1 && KBS * MS * NS <= 16) {
                  // A whole stage (every one of its KBS blocks exists: the common case) as ONE software pipeline -- all fragment and scale reads,
                  // then the KBS * MS * NS MFMAs back to back, then the promotions in K-block order (the arithmetic of the loop below, same
                  // bits).  The loop below runs a block at a time with a branch between blocks: read, wait, MFMA, wait, FMA -- four dependent
                  // latency chains per stage where this form has one (round 4: the 64 x 32 tile spends a third of its time in them).
                  if ((LW == 0 || wave < NW) && (sb + 1) * KBS <= num_kb) {
                      v8i bfq[KBS][NS], afq[KBS][MS];
                      float sc[KBS][MS];
                      #pragma unroll
                      for (int u = 0; u < KBS; ++u) {
                          const int jb = sb * KBS + u;
                          const uint8_t* stage = lds + cur + u * BLOCK_BYTES;
                          const uint8_t* sfg = lds + SFG_OFF + ((jb >> 2) & (SFG_SLOTS - 1)) * SFG_SLOT;
                          const float sb_val = *reinterpret_cast<const float*>(sfg + 1024 + (jb & 3) * 4);
                          if constexpr (MS == 4) {
                              const v4f q = *reinterpret_cast<const v4f*>(sfg + (jb & 3) * 256 + (wm * WM + (lane & 15) * MS) * 4);
                              sc[u][0] = q[0] * sb_val; sc[u][1] = q[1] * sb_val; sc[u][2] = q[2] * sb_val; sc[u][3] = q[3] * sb_val;
                          } else {
                              sc[u][0] = *reinterpret_cast<const float*>(sfg + (jb & 3) * 256 + (wm * WM + (lane & 15)) * 4) * sb_val;
                          }
                          #pragma unroll
                          for (int ns = 0; ns < NS; ++ns)
                              bfq[u][ns] = load_fragment(stage + A_BYTES + (wn * WN) * 128 + ns * 2048, frag_off);
                          #pragma unroll
                          for (int ms = 0; ms < MS; ++ms)
                              afq[u][ms] = load_fragment(stage + (wm * WM) * 128 + ms * 2048, frag_off);
                      }
                      v4f part[KBS][MS][NS];
                      #pragma unroll
                      for (int u = 0; u < KBS; ++u)
                          #pragma unroll
                          for (int ms = 0; ms < MS; ++ms)
                              #pragma unroll
                              for (int ns = 0; ns < NS; ++ns)
                                  part[u][ms][ns] = mfma_fp8_k128(bfq[u][ns], afq[u][ms]);
                      #pragma unroll
                      for (int u = 0; u < KBS; ++u)
                          #pragma unroll
                          for (int ms = 0; ms < MS; ++ms)
                              #pragma unroll
                              for (int ns = 0; ns < NS; ++ns)
                                  acc[ms][ns] += sc[u][ms] * part[u][ms][ns];
                      stage_done = true;
                  }
              }
              if (!stage_done && (LW == 0 || wave < NW))               // (loader waves: pieces and barriers only)
              #pragma unroll
              for (int u = 0; u < KBS; ++u) {
                if (sb * KBS + u >= num_kb)
                    break;
                const uint8_t* stage = lds + cur + u * BLOCK_BYTES;
                if constexpr (E8) {
                    const int shift = G32 ? (lane >> 4) * 8 : ((sb * KBS + u) & 3) * 8;     // this block's byte of the quad's words (G32: the lane group's byte of the block's words)
                    int ea[MS], eb[NS];
                    // the block's words in its group's slot of the ring: the quad's (granularity 128) or its own of the slot's four K blocks
                    const int jb = sb * KBS + u;
                    const uint8_t* sfg = lds + SFG_OFF + ((jb >> 2) & (SFG_SLOTS - 1)) * SFG_SLOT;
                    const uint8_t* words_a = sfg + (GSG ? (jb & 3) * 256 : 0);
                    const uint8_t* words_b = sfg + SFA_BYTES + (GSG ? (jb & 3) * (BN * 4) : 0);
                    if constexpr (MS == 4) {
                        const v4i qa = *reinterpret_cast<const v4i*>(words_a + (wm * WM + (lane & 15) * MS) * 4);
                        #pragma unroll
                        for (int ms = 0; ms < MS; ++ms)
                            ea[ms] = static_cast<int>(static_cast<unsigned>(qa[ms]) >> shift);
                    } else {
                        ea[0] = static_cast<int>(*reinterpret_cast<const unsigned*>(words_a + (wm * WM + (lane & 15)) * 4) >> shift);
                    }
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        // the weight row in MFMA row slot i = lane & 15 of N-subtile ns (see b_row_perm)
                        const int col = b_row_perm<WN>(wn * WN + ns * 16 + (lane & 15));
                        eb[ns] = static_cast<int>(*reinterpret_cast<const unsigned*>(words_b + col * 4) >> shift);
                    }
                    const uint8_t* a_tile = stage + (wm * WM) * 128;
                    const uint8_t* b_tile = stage + A_BYTES + (wn * WN) * 128;
                    v8i bf[NS];
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        const v8i af = load_fragment(a_tile + ms * 2048, frag_off);
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            acc[ms][ns] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[ns], af, acc[ms][ns], 0, 0, 0, eb[ns], 0, ea[ms]);
                    }
                    continue;
                }
                float sa[MS];
                const int jb = sb * KBS + u;                                                       // this K block
                const uint8_t* sfg = lds + SFG_OFF + ((jb >> 2) & (SFG_SLOTS - 1)) * SFG_SLOT;     // its group's slot: [4 blocks][64 rows], then 4 SFB values
                if constexpr (MS == 4) {
                    const v4f q = *reinterpret_cast<const v4f*>(sfg + (jb & 3) * 256 + (wm * WM + (lane & 15) * MS) * 4);
                    sa[0] = q[0]; sa[1] = q[1]; sa[2] = q[2]; sa[3] = q[3];
                } else {
                    sa[0] = *reinterpret_cast<const float*>(sfg + (jb & 3) * 256 + (wm * WM + (lane & 15)) * 4);
                }
                const float sb_val = *reinterpret_cast<const float*>(sfg + 1024 + (jb & 3) * 4);
                const uint8_t* a_tile = stage + (wm * WM) * 128;
                const uint8_t* b_tile = stage + A_BYTES + (wn * WN) * 128;
                v8i bf[NS];
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    const v8i af = load_fragment(a_tile + ms * 2048, frag_off);
                    const float scale = sa[ms] * sb_val;
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        const v4f part = mfma_fp8_k128(bf[ns], af);
                        acc[ms][ns] += scale * part;
                    }
                }
              }
                fill = cur;
                cur = (cur == (STAGES - 1) * STAGE_BYTES) ? 0 : cur + STAGE_BYTES;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if constexpr (KSPLIT) {
            if (ks_pieces > 1) {
                uint8_t* ws = static_cast<uint8_t*>(p.sk_workspace);
                unsigned* flags = reinterpret_cast<unsigned*>(ws + 4096) + tile * 8;
                uint8_t* slabs = ws + 4096 + 32768 + static_cast<int64_t>(tile) * 8 * (BM * BN * 4);
                const int lane_off = (wave * 64 + lane) * 16;
                if (ks_piece != ks_pieces - 1) {
                    const auto slab = __builtin_amdgcn_make_buffer_rsrc(slabs + static_cast<int64_t>(ks_piece) * (BM * BN * 4), 0, BM * BN * 4, 0x00020000);
                    if (LW == 0 || wave < NW) {
                        #pragma unroll
                        for (int ms = 0; ms < MS; ++ms)
                            #pragma unroll
                            for (int ns = 0; ns < NS; ++ns)
                                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, acc[ms][ns]), slab, lane_off, (ms * NS + ns) * (NW * 1024), 17);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();                    // every wave's partial is acknowledged (written through) before the flag goes out
                    if (threadIdx.x == 0)
                        __hip_atomic_store(flags + ks_piece, p.sk_exchange, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    continue;
                }
                if (threadIdx.x < ks_pieces - 1) {
                    // the pieces in front of this one were dispatched earlier: resident or done.  Bounded all the same: a lost flag must end in a
                    // wrong tile, not in a hung device
                    int spins = 0;
                    while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_exchange && ++spins < (1 << 22))
                        __builtin_amdgcn_s_sleep(4);
                }
                __syncthreads();
                // the partials of up to four pieces are in flight together (a written-through slab is a ~2 us round trip: one piece at a time
                // made the exchange of a 4-piece tile ~8 us); the SUM keeps the piece order
                v4f total[MS][NS];
                if (LW == 0 || wave < NW)
                for (int qb = 0; qb < ks_pieces - 1; qb += 4) {
                    v4f part[4][MS][NS];
                    #pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (qb + u < ks_pieces - 1) {
                            const auto slab = __builtin_amdgcn_make_buffer_rsrc(slabs + static_cast<int64_t>(qb + u) * (BM * BN * 4), 0, BM * BN * 4, 0x00020000);
                            #pragma unroll
                            for (int ms = 0; ms < MS; ++ms)
                                #pragma unroll
                                for (int ns = 0; ns < NS; ++ns)
                                    part[u][ms][ns] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(slab, lane_off, (ms * NS + ns) * (NW * 1024), 17));
                        }
                    #pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (qb + u < ks_pieces - 1) {
                            #pragma unroll
                            for (int ms = 0; ms < MS; ++ms)
                                #pragma unroll
                                for (int ns = 0; ns < NS; ++ns)
                                    total[ms][ns] = qb + u == 0 ? part[u][ms][ns] : total[ms][ns] + part[u][ms][ns];
                        }
                }
                if (LW == 0 || wave < NW) {
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms)
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            acc[ms][ns] = total[ms][ns] + acc[ms][ns];      // piece order: ((p0 + p1) + ...) + the last piece
                }
                // the flags are taken back: a launch replayed from a hipGraph carries the SAME epoch value again, and flags left standing would
                // let the next replay's last piece run ahead of its partials (every wave's loads have returned: their values were just used)
                __syncthreads();
                if (threadIdx.x < ks_pieces - 1)
                    __hip_atomic_store(flags + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (LW == 0 || wave < NW)
            store_tile<MS, NS, true>(p, t, ad_group * p.d_sg, acc, t.m0 + wm * WM, t.n0 + wn * WN);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES, int B_AUX = 0, int KBS = 1, bool E8 = false, int LW = 0, bool KSPLIT = false, bool G32 = false>
__global__ __launch_bounds__((WAVES_M * WAVES_N + LW) * 64)
void dg_fp8_gemm_stream_kernel(const GemmParams p) {
    stream_kernel_body<BM, BN, WAVES_M, WAVES_N, STAGES, B_AUX, KBS, E8, LW, KSPLIT, G32>(p);
}

// ---------------------------------------------------------------------------------------------------------------
// Split-ring stream kernel (round 4): the 64-row stream tile with the two operands on SEPARATE rings fed by SEPARATE waves.
//
// What bounds the single-ring stream tile at mid M (m = 64 .. 256, dense): a stage completes when its slowest piece has landed, and
// the weight pieces come from HBM (~2 us loaded) while the activation pieces come from the L2 (~0.5 us) -- yet both wait in the same
// ring: of the 96 KiB a CU keeps in flight on the 64 x 32 tile only a third is weight bytes, 32 KiB / 2 us = 16 GB/s of weight stream
// per CU (measured: 38-40 GB/s per CU in total whether 4, 8 or 16 waves issue the pieces and whether 128 or 256 CUs are busy:
// profiles/r04_probe/sweep_midm_loader_waves.jsonl).  The two streams cannot simply get different prefetch distances inside one wave:
// vector-memory operations retire IN ORDER per wave, so a wait for a young activation piece is a wait for every older weight piece.
// Per WAVE, though: here waves [0, AW) issue nothing but A pieces (a short ring: SA stages) and waves [AW, AW + BW) nothing but B pieces
// (a deep ring: SB stages) -- each group waits on its own counter for "my pieces of stage sb" and the stage barrier joins them.
// Compute waves = the first WAVES_M * WAVES_N of the A loaders (8 MFMAs per K block: nothing).  FP32 scales ride in the group ring
// of the single-ring kernel (one 16-byte-per-lane piece per four K blocks), issued by the A loaders.
// ---------------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int KBS, int SA, int SB, int AW, int BW, int B_AUX = 0>
__device__ __forceinline__ void stream2_kernel_body(const GemmParams& p) {
    constexpr int NW = WAVES_M * WAVES_N, TW = AW + BW;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    constexpr int A_STAGE = KBS * A_BYTES, B_STAGE = KBS * B_BYTES;
    constexpr int B_RING = SA * A_STAGE, SFG_OFF = B_RING + SB * B_STAGE, SFG_SLOT = 1024 + 256, SFG_SLOTS = 4;
    constexpr int LDS_BYTES = SFG_OFF + SFG_SLOTS * SFG_SLOT;
    constexpr int A_PER = KBS * (BM / 8) / AW, B_PER = KBS * (BN / 8) / BW;       // pieces per loader wave and stage
    static_assert(BM == 64 && (BN == 128 || BN == 64 || BN == 32), "one 256-byte SFA row per K block and one SFB value per tile");
    static_assert((KBS * (BM / 8)) % AW == 0 && (KBS * (BN / 8)) % BW == 0 && A_PER >= 1 && B_PER >= 1, "every loader wave of a group issues the same number of pieces");
    static_assert(NW <= AW && TW <= 16, "the compute waves are A loaders; 1024 threads at most");
    static_assert((MS == 4 || MS == 1) && NS % 2 == 0, "a lane reads its MS row scales with one LDS read");
    static_assert(SA >= 3 && SB >= 3 && SA * KBS <= 4 * (SFG_SLOTS - 1), "ring depths; a scale group slot is refilled only after its last reader");
    static_assert((SA - 1) * (A_PER + 2) < 64 && (SB - 1) * B_PER < 64, "vmcnt is a 6-bit counter");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    constexpr unsigned OOB = 0x80000000u;

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool a_loader = wave < AW, computes = wave < NW;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int num_kb = p.k / 128;
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);
    auto a_unit_row = [](int u) { return (u / (WM / 8)) * WM + (u & 1) * 8 * MS + ((u % (WM / 8)) >> 1); };
    const int a_voff = piece_row * MS * lda + src_chunk * 16;

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
            const uint8_t* a_base = uniform_pointer(p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm);
            const uint8_t* b_base = uniform_pointer(p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn);
            const int a_rows = uniform_int(imin(t.m_end - t.m0, BM)), b_rows = uniform_int(imin(p.n - t.n0, BN));
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base), 0, (a_rows - 1) * lda + p.k, 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base), 0, (b_rows - 1) * ldb + p.k, 0x00020000);
            const int sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kb_stride = static_cast<int>(p.sfb_sk) * 4;
            float* sfa_tile = uniform_pointer(const_cast<float*>(p.sfa) + ad_group * p.sfa_sg + t.m0);
            const int sfa_rows = uniform_int(imin(p.m - t.m0, BM));
            const auto sfa_rsrc = __builtin_amdgcn_make_buffer_rsrc(sfa_tile, 0, (num_kb - 1) * sfa_kb_stride + (sfa_rows + 3) / 4 * 4 * 4, 0x00020000);
            float* sfb_tile = uniform_pointer(const_cast<float*>(p.sfb) + static_cast<int64_t>(t.group) * p.sfb_sg + static_cast<int64_t>(t.n0 / 128) * p.sfb_sn);
            const auto sfb_rsrc = __builtin_amdgcn_make_buffer_rsrc(sfb_tile, 0, (num_kb - 1) * sfb_kb_stride + 4, 0x00020000);
            const int sfg_a_voff = (lane >> 4) * sfa_kb_stride + (lane & 15) * 16, sfg_b_voff = (lane & 3) * sfb_kb_stride;

            // stage sb (K blocks sb * KBS ..) of this wave's operand into its ring slot; blocks past the end: out-of-range no-ops (exact counts)
            auto issue_a_stage = [&](int slot_off, int sb) {
                #pragma unroll
                for (int u = 0; u < KBS; ++u) {
                    const int j = sb * KBS + u;
                    if ((j & 3) == 0) {                         // the scales of K blocks j .. j + 3 (every A loader: identical destinations)
                        const unsigned oob = j < num_kb ? 0u : OOB;
                        uint8_t* slot = lds + SFG_OFF + ((j >> 2) & (SFG_SLOTS - 1)) * SFG_SLOT;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(sfa_rsrc, (__attribute__((address_space(3))) void*)slot, 16,
                                                                 static_cast<int>(static_cast<unsigned>(sfg_a_voff) | oob), j * sfa_kb_stride, 0, 0);
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(sfb_rsrc, (__attribute__((address_space(3))) void*)(slot + 1024), 4,
                                                                 static_cast<int>(static_cast<unsigned>(sfg_b_voff) | oob), j * sfb_kb_stride, 0, 0);
                    }
                }
                #pragma unroll
                for (int q = 0; q < A_PER; ++q) {
                    const int idx = wave + AW * q, u = idx / (BM / 8), unit = idx % (BM / 8), j = sb * KBS + u;
                    const unsigned oob = j < num_kb ? 0u : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        a_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + u * A_BYTES + unit * 1024), 16,
                        static_cast<int>(static_cast<unsigned>(a_voff) + (static_cast<unsigned>(a_unit_row(unit) * lda) | oob)), j * 128, 0, 0);
                }
            };
            auto issue_b_stage = [&](int slot_off, int sb) {
                #pragma unroll
                for (int q = 0; q < B_PER; ++q) {
                    const int idx = (wave - AW) + BW * q, u = idx / (BN / 8), unit = idx % (BN / 8), j = sb * KBS + u;
                    const unsigned oob = j < num_kb ? 0u : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        b_rsrc, (__attribute__((address_space(3))) void*)(lds + B_RING + slot_off + u * B_BYTES + unit * 1024), 16,
                        static_cast<int>(static_cast<unsigned>(b_row_perm<WN>(unit * 8 + piece_row) * ldb + src_chunk * 16) | oob), j * 128, 0,
                        B_AUX & 3);
                }
            };
            if (a_loader) {
                #pragma unroll
                for (int j = 0; j < SA - 1; ++j)
                    issue_a_stage(j * A_STAGE, j);
            } else {
                #pragma unroll
                for (int j = 0; j < SB - 1; ++j)
                    issue_b_stage(j * B_STAGE, j);
            }

            int a_cur = 0, a_fill = (SA - 1) * A_STAGE, b_cur = 0, b_fill = (SB - 1) * B_STAGE;
            const int num_sb = (num_kb + KBS - 1) / KBS;
            for (int sb = 0; sb < num_sb; ++sb) {
                // my pieces of stage sb have landed (the younger stages of MY ring may still fly; the A loaders' uncounted scale pieces make
                // their wait stricter, never looser); the barrier joins the two groups and frees the slots of stage sb - 1
                if (a_loader)
                    asm volatile("s_waitcnt vmcnt(%c0)" :: "i"((SA - 2) * A_PER) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(%c0)" :: "i"((SB - 2) * B_PER) : "memory");
                raw_barrier();
                if (a_loader)
                    issue_a_stage(a_fill, sb + SA - 1);
                else
                    issue_b_stage(b_fill, sb + SB - 1);
                if (computes) {
                    #pragma unroll
                    for (int u = 0; u < KBS; ++u) {
                        if (sb * KBS + u >= num_kb)
                            break;
                        float sa[MS];
                        const int jb = sb * KBS + u;
                        const uint8_t* sfg = lds + SFG_OFF + ((jb >> 2) & (SFG_SLOTS - 1)) * SFG_SLOT;
                        if constexpr (MS == 4) {
                            const v4f q = *reinterpret_cast<const v4f*>(sfg + (jb & 3) * 256 + (wm * WM + (lane & 15) * MS) * 4);
                            sa[0] = q[0]; sa[1] = q[1]; sa[2] = q[2]; sa[3] = q[3];
                        } else {
                            sa[0] = *reinterpret_cast<const float*>(sfg + (jb & 3) * 256 + (wm * WM + (lane & 15)) * 4);
                        }
                        const float sb_val = *reinterpret_cast<const float*>(sfg + 1024 + (jb & 3) * 4);
                        const uint8_t* a_tile = lds + a_cur + u * A_BYTES + (wm * WM) * 128;
                        const uint8_t* b_tile = lds + B_RING + b_cur + u * B_BYTES + (wn * WN) * 128;
                        v8i bf[NS];
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                        #pragma unroll
                        for (int ms = 0; ms < MS; ++ms) {
                            const v8i af = load_fragment(a_tile + ms * 2048, frag_off);
                            const float scale = sa[ms] * sb_val;
                            #pragma unroll
                            for (int ns = 0; ns < NS; ++ns) {
                                const v4f part = mfma_fp8_k128(bf[ns], af);
                                acc[ms][ns] += scale * part;
                            }
                        }
                    }
                }
                a_fill = a_cur;
                a_cur = (a_cur == (SA - 1) * A_STAGE) ? 0 : a_cur + A_STAGE;
                b_fill = b_cur;
                b_cur = (b_cur == (SB - 1) * B_STAGE) ? 0 : b_cur + B_STAGE;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (computes)
            store_tile<MS, NS, true>(p, t, ad_group * p.d_sg, acc, t.m0 + wm * WM, t.n0 + wn * WN);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int KBS, int SA, int SB, int AW, int BW, int B_AUX = 0>
__global__ __launch_bounds__((AW + BW) * 64)
void dg_fp8_gemm_stream2_kernel(const GemmParams p) {
    stream2_kernel_body<BM, BN, WAVES_M, WAVES_N, KBS, SA, SB, AW, BW, B_AUX>(p);
}

// ---------------------------------------------------------------------------------------------------------------
// UE8M0 kernel: scales that are powers of two, handed over as packed exponent bytes (the reference's SM100 input format:
// int32 = four consecutive 128-K blocks of one row, MN-major; recipe (1, 1, 128): one scale per A row and per B row).
// The scaled MFMA applies 2^(ea + eb - 254) in hardware and accumulates across K blocks in its own FP32 accumulator:
// no promotion FMAs at all (128 of the FP32-scale path's ~255 instructions per wave per K block).  Verified on hardware
// (tools/ubench/mfma_scale_probe.hip): a lane's scale byte applies to its own row (lane & 15) and its own 32-K group
// (lane >> 4); 127 encodes 1.0; opsel picks the byte of the scale VGPR.
// Structure: the ring kernel's (3-slot A / 2-slot B rings, barriers P and Q, counted vmcnt); with ~3 filler instructions
// per MFMA a wave sustains the matrix pipe on its own, so no role split is needed.  Scale words for block kb+1 are
// loaded (inline asm, straight-line to their wait) at the top of block kb: two dwordx4 for the lane's 8 interleaved A rows
// and four dwords for its B rows.
// ---------------------------------------------------------------------------------------------------------------
struct E8Landing { v4i sa[2]; int sb[4]; };

__device__ __forceinline__ void issue_e8_scale_loads(E8Landing& l, const v4i& sfa_rsrc, int sfa_voff, const v4i& sfb_rsrc,
                                                     int sfb_voff0, int sfb_voff1, int sfb_voff2, int sfb_voff3) {
    asm volatile(
        "s_nop 4\n\t"      // SGPR operands written by VALU (v_readlane / v_readfirstlane) just before: 5 wait states, nothing pads an asm
        "buffer_load_dwordx4 %0, %6, %7, 0 offen\n\t"
        "buffer_load_dwordx4 %1, %6, %7, 0 offen offset:16\n\t"
        "buffer_load_dword %2, %8, %12, 0 offen\n\t"
        "buffer_load_dword %3, %9, %12, 0 offen\n\t"
        "buffer_load_dword %4, %10, %12, 0 offen\n\t"
        "buffer_load_dword %5, %11, %12, 0 offen"
        : "=&v"(l.sa[0]), "=&v"(l.sa[1]), "=&v"(l.sb[0]), "=&v"(l.sb[1]), "=&v"(l.sb[2]), "=&v"(l.sb[3])
        : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff0), "v"(sfb_voff1), "v"(sfb_voff2), "v"(sfb_voff3), "s"(sfb_rsrc)
        : "memory");
}

// A_MN: the rows of a wave tile keep their natural order (subtile ms = rows 16 ms .. + 15), so a lane's MS scale words lie 16 rows apart:
// MS dword loads instead of two dwordx4 (the ScaleLandingN of the FP32-scale kernels)
struct E8LandingN { int sa[8]; int sb[4]; };

__device__ __forceinline__ void issue_e8_scale_loads(E8LandingN& l, const v4i& sfa_rsrc, int sfa_voff, const v4i& sfb_rsrc,
                                                     int sfb_voff0, int sfb_voff1, int sfb_voff2, int sfb_voff3) {
    asm volatile(
        "s_nop 4\n\t"
        "buffer_load_dword %0, %12, %13, 0 offen\n\t"
        "buffer_load_dword %1, %12, %13, 0 offen offset:64\n\t"
        "buffer_load_dword %2, %12, %13, 0 offen offset:128\n\t"
        "buffer_load_dword %3, %12, %13, 0 offen offset:192\n\t"
        "buffer_load_dword %4, %12, %13, 0 offen offset:256\n\t"
        "buffer_load_dword %5, %12, %13, 0 offen offset:320\n\t"
        "buffer_load_dword %6, %12, %13, 0 offen offset:384\n\t"
        "buffer_load_dword %7, %12, %13, 0 offen offset:448\n\t"
        "buffer_load_dword %8, %14, %18, 0 offen\n\t"
        "buffer_load_dword %9, %15, %18, 0 offen\n\t"
        "buffer_load_dword %10, %16, %18, 0 offen\n\t"
        "buffer_load_dword %11, %17, %18, 0 offen"
        : "=&v"(l.sa[0]), "=&v"(l.sa[1]), "=&v"(l.sa[2]), "=&v"(l.sa[3]), "=&v"(l.sa[4]), "=&v"(l.sa[5]), "=&v"(l.sa[6]), "=&v"(l.sa[7]),
          "=&v"(l.sb[0]), "=&v"(l.sb[1]), "=&v"(l.sb[2]), "=&v"(l.sb[3])
        : "v"(sfa_voff), "s"(sfa_rsrc), "v"(sfb_voff0), "v"(sfb_voff1), "v"(sfb_voff2), "v"(sfb_voff3), "s"(sfb_rsrc)
        : "memory");
}

template <int ALLOWED>
__device__ __forceinline__ void wait_e8_landing(E8LandingN& l) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(ALLOWED, 0));
    asm volatile("" : "+v"(l.sa[0]), "+v"(l.sa[1]), "+v"(l.sa[2]), "+v"(l.sa[3]), "+v"(l.sa[4]), "+v"(l.sa[5]), "+v"(l.sa[6]), "+v"(l.sa[7]),
                      "+v"(l.sb[0]), "+v"(l.sb[1]), "+v"(l.sb[2]), "+v"(l.sb[3]) :: "memory");
}
__device__ __forceinline__ int e8_landed_sfa(const E8Landing& l, int ms) { return l.sa[ms / 4][ms % 4]; }
__device__ __forceinline__ int e8_landed_sfa(const E8LandingN& l, int ms) { return l.sa[ms]; }
template <bool NATURAL> struct E8LandingSel { typedef E8Landing type; };
template <> struct E8LandingSel<true> { typedef E8LandingN type; };

template <int ALLOWED>
__device__ __forceinline__ void wait_e8_landing(E8Landing& l) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(waitcnt_imm(ALLOWED, 0));
    asm volatile("" : "+v"(l.sa[0]), "+v"(l.sa[1]), "+v"(l.sb[0]), "+v"(l.sb[1]), "+v"(l.sb[2]), "+v"(l.sb[3]) :: "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// UE8M0 kernel in the duo schedule: the same role-split segments and wave-half stagger as dg_fp8_gemm_duo_kernel, with
// the hardware-scaled MFMA of dg_fp8_gemm_e8_kernel in the matrix segments (16 MFMAs, nothing else) and the packed scale
// words of block kb+1 loaded in L_a(kb), consumed (shifted to byte 0) in L_a(kb+1).
// ---------------------------------------------------------------------------------------------------------------
// B_MN (round 4): operand B MN-major ([K][N], unit stride along n) read in place -- the nn layout of a packed-scale caller without the
// re-majoring pass: the LDS-DMA pieces (4 k-rows x 256 bytes), the hardware transpose reads and the natural column order of the B_MN form
// of dg_fp8_gemm_duo_kernel; a lane's row slot i of N-subtile ns is weight row ns * 16 + i, and so is its scale word.
// A_MN: operand A MN-major ([K][M]: the tt / tn layouts) likewise; A rows in natural order (scale words as MS dword loads, epilogue without
// the row interleave).
// K_TAIL (round 5): K need not be a multiple of 128 (whole 16-byte chunks, K > 128): the partial last block is computed once per tile after
// the loop, as in duo_kernel_body -- K-major chunks at and beyond K pushed out of the descriptor's range (an out-of-range LDS-DMA lane writes
// zeros), MN-major k-rows >= K beyond the extent by themselves; its scale byte is byte (K / 128) & 3 of the last packed word.  The packed-scale
// dgrad shape fp8_gemm_nn 4096 x 7168 x 2112 then reads its MN-major B in place instead of re-majoring it in front of the 128-row quad kernel.
template <int BM, int BN, int WAVES_M, int WAVES_N, bool B_MN = false, bool A_MN = false, bool K_TAIL = false>
__device__ __forceinline__ void duo_e8_kernel_body(const GemmParams& p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N, MS = WM / 16, NS = WN / 16, HS = MS / 2;
    static_assert(!B_MN || (BN == 256 && NW == 8), "MN-major B tile: 128 k-rows x 256 bytes, 32 pieces over 8 waves");
    static_assert(!A_MN || (BM == 256 && NW == 8), "MN-major A tile: 128 k-rows x 256 bytes, 32 pieces over 8 waves");
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, A_SLOTS = 3, B_SLOTS = 2;
    constexpr int B_BASE = A_SLOTS * A_BYTES, LDS_BYTES = B_BASE + B_SLOTS * B_BYTES;
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN / 8 / NW, A_EARLY = A_ITERS / 2;
    static_assert(MS == 8 && NS == 4, "scale landing registers are written out for a 128 x 64 wave tile");
    static_assert(NW % 2 == 0 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "every wave issues the same number of pieces");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];
    // The MFMA here is a builtin, free to move: pin the instruction scheduler at every segment boundary so a matrix
    // segment cannot drift across its barrier into the neighbouring load segment.
    auto seg_barrier = [] {
        __builtin_amdgcn_sched_barrier(0);
        raw_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const bool upper_half = wave >= NW / 2;
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
    const int k_tail = K_TAIL ? (p.k & 127) : 0;
    const int num_kq = (num_kb + (k_tail != 0) + 3) / 4;
    const int sfa_kq_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kq_stride = static_cast<int>(p.sfb_sk) * 4;
    // MN-major B (see duo_kernel_body): lane l of piece u carries k-row 4u + (l >> 4), source chunk (l & 15) ^ f(k)
    [[maybe_unused]] const int ldb_mn = static_cast<int>(p.b_sk), lda_mn = static_cast<int>(p.a_sk);
    [[maybe_unused]] const int mn_chunk = ((lane & 15) ^ (((4 * (wave & 1) + (lane >> 4)) & 7) | (((wave >> 2) & 1) << 3))) << 4;
    [[maybe_unused]] const int bmn_voff = (lane >> 4) * ldb_mn + mn_chunk, amn_voff = (lane >> 4) * lda_mn + mn_chunk;
    [[maybe_unused]] const int tr_lane_base = (16 * (lane >> 4) + ((lane & 15) >> 1)) * 256 + (lane & 1) * 8;
    [[maybe_unused]] const int tr_swz = ((lane & 15) >> 1) | (((lane >> 4) & 1) << 3);
    const long long t_entry = p.dbg != nullptr ? DG_STAMP_CLOCK() : 0;
    long long t_loop0 = 0, t_loop1 = 0;

    MaskedWalk walk;
    const int num_launched = gridDim.x;
    // (tile_id, pass): contiguous layout with BM = 2 x alignment -- a tile whose halves belong to two groups is walked twice, once per
    // half with that group's B (round 5: the grouped nn form with packed scales reads its MN-major weights in place)
    int tile_id = blockIdx.x, pass = 0;
    for (;;) {
        const Tile t = get_tile<BM, BN>(p, tile_id, walk, pass);
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
            const uint8_t* a_base = uniform_pointer(p.a + ad_group * p.a_sg + static_cast<int64_t>(t.m0) * (A_MN ? 1 : p.a_sm));
            const uint8_t* b_base = uniform_pointer(p.b + static_cast<int64_t>(t.group) * p.b_sg + static_cast<int64_t>(t.n0) * (B_MN ? 1 : p.b_sn));
            const int a_rows = uniform_int(imin(t.m_end - t.m0, BM)), b_rows = uniform_int(imin(p.n - t.n0, BN));
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base), 0,
                                                                  A_MN ? uniform_int((p.k - 1) * lda_mn + (p.m - t.m0)) : (a_rows - 1) * lda + p.k,
                                                                  0x00020000);
            // (B_MN: the descriptor ends with the last k-row's valid bytes; a lane past N inside an earlier row reads the next row's head --
            //  finite bytes that only reach columns which are never stored)
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base), 0,
                                                                  B_MN ? uniform_int((p.k - 1) * ldb_mn + (p.n - t.n0)) : (b_rows - 1) * ldb + p.k,
                                                                  0x00020000);
            const uint64_t sfa_addr = reinterpret_cast<uint64_t>(p.sfa + ad_group * p.sfa_sg);
            const uint64_t sfb_addr = reinterpret_cast<uint64_t>(p.sfb + static_cast<int64_t>(t.group) * p.sfb_sg);
            const v4i sfa_rsrc = {__builtin_amdgcn_readfirstlane(static_cast<int>(sfa_addr)),
                                  __builtin_amdgcn_readfirstlane(static_cast<int>(sfa_addr >> 32) & 0xffff),
                                  __builtin_amdgcn_readfirstlane((num_kq - 1) * sfa_kq_stride + p.m * 4), 0x00020000};
            const v4i sfb_rsrc = {__builtin_amdgcn_readfirstlane(static_cast<int>(sfb_addr)),
                                  __builtin_amdgcn_readfirstlane(static_cast<int>(sfb_addr >> 32) & 0xffff),
                                  __builtin_amdgcn_readfirstlane((num_kq - 1) * sfb_kq_stride + p.n * 4), 0x00020000};
            const int sfa_voff = (t.m0 + wm * WM + (lane & 15) * (A_MN ? 1 : MS)) * 4;        // A_MN: + ms * 64 bytes in the loads
            int sfb_voff[NS];
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns) {
                const int i = lane & 15;
                sfb_voff[ns] = B_MN ? (t.n0 + wn * WN + ns * 16 + i) * 4          // natural row order
                                    : (t.n0 + wn * WN + (ns >> 1) * 32 + (i >> 2) * 8 + (ns & 1) * 4 + (i & 3)) * 4;
            }
            auto issue_a_piece = [&](int slot_off, int j, int q) {
                const int unit = wave + NW * q;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    a_rsrc, (__attribute__((address_space(3))) void*)(lds + slot_off + unit * 1024), 16,
                    A_MN ? amn_voff : a_piece_voff[q],
                    A_MN ? (imin(j, num_kb - 1) * 128 + 4 * unit) * lda_mn : imin(j, num_kb - 1) * 128, 0, 0);
            };
            auto issue_b_piece = [&](int slot_off, int j, int q) {
                const int unit = wave + NW * q;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    b_rsrc, (__attribute__((address_space(3))) void*)(lds + B_BASE + slot_off + unit * 1024), 16,
                    B_MN ? bmn_voff : b_piece_voff[q],
                    B_MN ? (imin(j, num_kb - 1) * 128 + 4 * unit) * ldb_mn : imin(j, num_kb - 1) * 128, 0, 0);
            };
            typename E8LandingSel<A_MN>::type land;
            auto issue_scales = [&](int j) {
                const int kq = imin(j, num_kb - 1) >> 2;
                issue_e8_scale_loads(land, sfa_rsrc, sfa_voff + kq * sfa_kq_stride, sfb_rsrc, sfb_voff[0] + kq * sfb_kq_stride,
                                     sfb_voff[1] + kq * sfb_kq_stride, sfb_voff[2] + kq * sfb_kq_stride,
                                     sfb_voff[3] + kq * sfb_kq_stride);
            };
            int sa_cur[MS], sb_cur[NS];
            auto take_scales = [&](int j) {
                const int shift = (imin(j, num_kb - 1) & 3) * 8;
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    sa_cur[ms] = static_cast<int>(static_cast<unsigned>(e8_landed_sfa(land, ms)) >> shift);
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    sb_cur[ns] = static_cast<int>(static_cast<unsigned>(land.sb[ns]) >> shift);
            };

            // ---- prologue: A(0) B(0) A(1) B(1) | scale words of block 0, full drain (straight-line load -> wait) ----
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
            seg_barrier();
            if (upper_half)
                seg_barrier();                      // the upper half runs one segment behind from here on

            int a_cur = 0, a_fill = 2 * A_BYTES, b_cur = 0;
            v8i bf[NS], af[HS];
            if (p.dbg != nullptr) t_loop0 = DG_STAMP_CLOCK();
            for (int kb = 0; kb < num_kb; ++kb) {
                const uint8_t* a_tile = lds + a_cur + (wm * WM) * 128;
                const uint8_t* b_tile = lds + B_BASE + b_cur + (wn * WN) * 128;

                // ---------------- L_a ----------------
                seg_barrier();
                [[maybe_unused]] FragTr bfq[NS];
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns) {
                    if constexpr (B_MN) bfq[ns] = load_fragment_tr(lds + B_BASE + b_cur, tr_lane_base, ((wn * (WN / 16) + ns) ^ tr_swz) << 4);
                    else bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                }
                [[maybe_unused]] FragTr afq[HS];
                #pragma unroll
                for (int h = 0; h < HS; ++h) {
                    if constexpr (A_MN) afq[h] = load_fragment_tr(lds + a_cur, tr_lane_base, ((wm * MS + h) ^ tr_swz) << 4);
                    else af[h] = load_fragment(a_tile + h * 2048, frag_off);
                }
                take_scales(kb);                    // the words landed before the previous L_b's wait (or the prologue's)
                // (A packed word covers four K blocks, so three of four of these fetches are redundant -- but issuing them
                // conditionally puts a control-flow join between the asm loads and their wait, where hipcc copies the landing
                // registers before the data is there: tried, wrong results.  It would take a 4x unrolled block body.)
                issue_scales(kb + 1);
                #pragma unroll
                for (int q = 0; q < A_EARLY; ++q)
                    issue_a_piece(a_fill, kb + 2, q);
                __builtin_amdgcn_s_waitcnt(waitcnt_imm(63, 0));
                if constexpr (B_MN) {               // the transpose reads have landed: now they may become MFMA operands
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        bf[ns] = assemble_fragment_tr(bfq[ns]);
                }
                if constexpr (A_MN) {
                    #pragma unroll
                    for (int h = 0; h < HS; ++h)
                        af[h] = assemble_fragment_tr(afq[h]);
                }
                asm volatile("" : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]), "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3])
                             :: "memory");

                // ---------------- M_a ----------------
                seg_barrier();
                #pragma unroll
                for (int h = 0; h < HS; ++h)
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        acc[h][ns] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[ns], af[h], acc[h][ns], 0, 0, 0, sb_cur[ns],
                                                                                      0, sa_cur[h]);
                asm volatile("" ::: "memory");

                // ---------------- L_b ----------------
                seg_barrier();
                #pragma unroll
                for (int h = 0; h < HS; ++h) {
                    if constexpr (A_MN) afq[h] = load_fragment_tr(lds + a_cur, tr_lane_base, ((wm * MS + HS + h) ^ tr_swz) << 4);
                    else af[h] = load_fragment(a_tile + (HS + h) * 2048, frag_off);
                }
                #pragma unroll
                for (int q = A_EARLY; q < A_ITERS; ++q)
                    issue_a_piece(a_fill, kb + 2, q);
                #pragma unroll
                for (int q = 0; q < B_ITERS; ++q)
                    issue_b_piece(b_cur, kb + 2, q);
                wait_e8_landing<A_ITERS + B_ITERS>(land);       // block kb+1 and its scale words: my pieces have landed (lgkmcnt(0) too)
                if constexpr (A_MN) {
                    #pragma unroll
                    for (int h = 0; h < HS; ++h)
                        af[h] = assemble_fragment_tr(afq[h]);
                }
                asm volatile("" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]) :: "memory");

                // ---------------- M_b ----------------
                seg_barrier();
                #pragma unroll
                for (int h = 0; h < HS; ++h)
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        acc[HS + h][ns] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[ns], af[h], acc[HS + h][ns], 0, 0, 0,
                                                                                           sb_cur[ns], 0, sa_cur[HS + h]);
                asm volatile("" ::: "memory");

                const int a_next = (a_cur == (A_SLOTS - 1) * A_BYTES) ? 0 : a_cur + A_BYTES;
                a_fill = a_cur;
                a_cur = a_next;
                b_cur ^= B_BYTES;
            }
            if (!upper_half)
                seg_barrier();              // pairs with the barrier in front of the upper half's last segment
            if (p.dbg != nullptr) t_loop1 = DG_STAMP_CLOCK();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if constexpr (K_TAIL) {
                if (k_tail != 0) {
                    const int tail_bias = (src_chunk * 16 >= k_tail) ? 0x40000000 : 0;
                    #pragma unroll
                    for (int q = 0; q < A_ITERS; ++q) {
                        const int unit = wave + NW * q;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            a_rsrc, (__attribute__((address_space(3))) void*)(lds + unit * 1024), 16,
                            A_MN ? amn_voff : a_piece_voff[q] + tail_bias,
                            A_MN ? (num_kb * 128 + 4 * unit) * lda_mn : num_kb * 128, 0, 0);
                    }
                    #pragma unroll
                    for (int q = 0; q < B_ITERS; ++q) {
                        const int unit = wave + NW * q;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            b_rsrc, (__attribute__((address_space(3))) void*)(lds + B_BASE + unit * 1024), 16,
                            B_MN ? bmn_voff : b_piece_voff[q] + tail_bias,
                            B_MN ? (num_kb * 128 + 4 * unit) * ldb_mn : num_kb * 128, 0, 0);
                    }
                    {
                        const int kq = num_kb >> 2;
                        issue_e8_scale_loads(land, sfa_rsrc, sfa_voff + kq * sfa_kq_stride, sfb_rsrc, sfb_voff[0] + kq * sfb_kq_stride,
                                             sfb_voff[1] + kq * sfb_kq_stride, sfb_voff[2] + kq * sfb_kq_stride,
                                             sfb_voff[3] + kq * sfb_kq_stride);
                    }
                    wait_e8_landing<0>(land);                   // straight-line from the loads; also lands the pieces
                    __syncthreads();
                    {
                        const int shift = (num_kb & 3) * 8;
                        #pragma unroll
                        for (int ms = 0; ms < MS; ++ms)
                            sa_cur[ms] = static_cast<int>(static_cast<unsigned>(e8_landed_sfa(land, ms)) >> shift);
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            sb_cur[ns] = static_cast<int>(static_cast<unsigned>(land.sb[ns]) >> shift);
                    }
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        if constexpr (B_MN) {
                            FragTr fq = load_fragment_tr(lds + B_BASE, tr_lane_base, ((wn * (WN / 16) + ns) ^ tr_swz) << 4);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            bf[ns] = assemble_fragment_tr(fq);
                        } else {
                            bf[ns] = load_fragment(lds + B_BASE + (wn * WN) * 128 + ns * 2048, frag_off);
                        }
                    }
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        __builtin_amdgcn_sched_barrier(0);      // one subtile row at a time
                        v8i a_frag;
                        if constexpr (A_MN) {
                            FragTr fq = load_fragment_tr(lds, tr_lane_base, ((wm * MS + ms) ^ tr_swz) << 4);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            a_frag = assemble_fragment_tr(fq);
                        } else {
                            a_frag = load_fragment(lds + (wm * WM) * 128 + ms * 2048, frag_off);
                        }
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            acc[ms][ns] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[ns], a_frag, acc[ms][ns], 0, 0, 0, sb_cur[ns],
                                                                                           0, sa_cur[ms]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                }
            }
        }
        store_tile<MS, NS, !A_MN, false, B_MN>(p, t, ad_group * p.d_sg, acc, t.m0 + wm * WM, t.n0 + wn * WN);
        if (p.dbg != nullptr && tile_id == static_cast<int>(blockIdx.x)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            dbg_stamp(p, NW, 0, t_entry);
            dbg_stamp(p, NW, 1, t_loop0);
            dbg_stamp(p, NW, 2, t_loop1);
            dbg_stamp(p, NW, 3, DG_STAMP_CLOCK());
        }
        if (t.second_pass) {
            pass = 1;
        } else {
            pass = 0;
            tile_id += num_launched;
        }
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool B_MN = false, bool A_MN = false, bool K_TAIL = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64)
void dg_fp8_gemm_duo_e8_kernel(const GemmParams p) {
    duo_e8_kernel_body<BM, BN, WAVES_M, WAVES_N, B_MN, A_MN, K_TAIL>(p);
}

// ---------------------------------------------------------------------------------------------------------------
// Second phase of the K split of an under-filled recipe-(1, 1, 128) launch (dg_api.hip: launch_per_col_split): D (+)= the sum of
// `pieces` FP32 partial matrices [m][n] (dense, piece_stride floats apart), added in piece order (bit-repeatable), then the
// operator's own output step -- FP32 or BF16, plain or reduce-add in D's dtype.  Four columns per thread, grid-stride.
// ---------------------------------------------------------------------------------------------------------------
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_sum_partials_kernel(const float* __restrict__ parts, int pieces, int64_t piece_stride, void* d, int m, int n, int64_t d_sm,
                            int d_dtype, int accumulate, int vec_ok) {
    const int64_t quads = (n + 3) / 4, total = static_cast<int64_t>(m) * quads;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t row = i / quads;
        const int col = static_cast<int>(i - row * quads) * 4;
        const int cnt = n - col < 4 ? n - col : 4;
        const float* src = parts + row * n + col;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec_ok && cnt == 4) {
            for (int q = 0; q < pieces; ++q) {
                const v4f t = *reinterpret_cast<const v4f*>(src + q * piece_stride);
                v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
            }
        } else {
            for (int q = 0; q < pieces; ++q)
                for (int e = 0; e < cnt; ++e)
                    v[e] += src[q * piece_stride + e];
        }
        if (d_dtype != 0) {
            float* dst = reinterpret_cast<float*>(d) + row * d_sm + col;
            if (vec_ok && cnt == 4) {
                v4f out = {v[0], v[1], v[2], v[3]};
                if (accumulate) out += *reinterpret_cast<const v4f*>(dst);
                *reinterpret_cast<v4f*>(dst) = out;
            } else {
                for (int e = 0; e < cnt; ++e)
                    dst[e] = accumulate ? dst[e] + v[e] : v[e];
            }
        } else {
            uint16_t* dst = reinterpret_cast<uint16_t*>(d) + row * d_sm + col;
            for (int e = 0; e < cnt; ++e) {
                float w = round_bf16(v[e]);
                if (accumulate) w = w + bf16_lo(static_cast<uint32_t>(dst[e]));
                dst[e] = static_cast<uint16_t>(pack_bf16(w, 0.f) & 0xffffu);
            }
        }
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// Generic path: any operand majorness / alignment / K tail, both SFB granularities.  128 x 128 tile, 4 waves,
// register-staged loads written into the same swizzled LDS image.  Correctness first.
// ---------------------------------------------------------------------------------------------------------------
// 16 consecutive elements along the unit-stride dimension starting at `src`; elements >= nvalid read as zero.
__device__ __forceinline__ uint4 fetch16(const uint8_t* src, int nvalid) {
    if (nvalid >= 16 && (reinterpret_cast<uintptr_t>(src) & 15) == 0)
        return *reinterpret_cast<const uint4*>(src);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
    for (int e = 0; e < 16; ++e)
        if (e < nvalid)
            w[e >> 2] |= static_cast<uint32_t>(src[e]) << ((e & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// Loads one 128-row x 128-K operand tile into registers (4 x 16 bytes per thread of a 256-thread block).
//   K-major  (stride_k == 1): item i -> row (tid + 256 i) / 8, chunk (tid + 256 i) % 8 (16 bytes along K)
//   MN-major (stride_mn == 1): item i -> k (tid + 256 i) / 8, row-chunk (tid + 256 i) % 8 (16 rows at one k)
struct StagedTile { uint4 v[4]; };

__device__ __forceinline__ StagedTile stage_load(const uint8_t* base, int64_t stride_mn, int64_t stride_k,
                                                 int rows_valid, int k_valid, bool permute_rows) {
    StagedTile s;
    const int tid = threadIdx.x;
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = tid + 256 * i;
        if (stride_k == 1) {
            const int p = item >> 3, chunk = item & 7;
            const int row = permute_rows ? b_row_perm<64>(p) : p;
            const int nvalid = row < rows_valid ? imin(imax(k_valid - chunk * 16, 0), 16) : 0;
            s.v[i] = nvalid > 0 ? fetch16(base + static_cast<int64_t>(row) * stride_mn + chunk * 16, nvalid)
                                : make_uint4(0u, 0u, 0u, 0u);
        } else {
            const int kk = item >> 3, row0 = (item & 7) * 16;
            const int nvalid = kk < k_valid ? imin(imax(rows_valid - row0, 0), 16) : 0;
            s.v[i] = nvalid > 0 ? fetch16(base + static_cast<int64_t>(kk) * stride_k + row0, nvalid)
                                : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    return s;
}

// inverse of b_row_perm<64> restricted to one 64-row wave range is not needed: MN-major loads write each byte to
// the LDS row position whose permuted row equals the loaded row, found through a small search-free formula below.
template <int WN>
__device__ __forceinline__ int b_row_perm_inv(int n_local) {
    const int w = n_local / WN, q = n_local % WN;
    const int pair = q >> 5, lg = (q >> 3) & 3, odd = (q >> 2) & 1, r = q & 3;
    return w * WN + (pair * 2 + odd) * 16 + lg * 4 + r;
}

__device__ __forceinline__ void stage_store(uint8_t* tile, const StagedTile& s, int64_t stride_k, bool permute_rows) {
    const int tid = threadIdx.x;
    #pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int item = tid + 256 * i;
        if (stride_k == 1) {
            const int p = item >> 3, chunk = item & 7;
            *reinterpret_cast<uint4*>(tile + lds_chunk_offset(p, chunk)) = s.v[i];
        } else {
            const int kk = item >> 3, row0 = (item & 7) * 16;
            const uint32_t w[4] = {s.v[i].x, s.v[i].y, s.v[i].z, s.v[i].w};
            #pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row0 + e;
                const int p = permute_rows ? b_row_perm_inv<64>(row) : row;
                tile[lds_chunk_offset(p, kk >> 4) + (kk & 15)] = static_cast<uint8_t>(w[e >> 2] >> ((e & 3) * 8));
            }
        }
    }
}

#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_fp8_gemm_generic_kernel(const GemmParams p) {
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, MS = 4, NS = 4;
    __shared__ __attribute__((aligned(1024))) uint8_t lds[2 * 128 * 128];
    uint8_t* a_lds = lds;
    uint8_t* b_lds = lds + BM * 128;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lg = lane >> 4;
    const int num_kb = (p.k + 127) / 128;
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
            const int rows_a = t.m_end - t.m0, rows_b = p.n - t.n0;

            int sfa_row[MS];
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                sfa_row[ms] = t.m0 + imin(wm * WM + ms * 16 + (lane & 15), rows_a - 1);
            const float* sfa_g = p.sfa + ad_group * p.sfa_sg;
            const float* sfb_g = p.sfb + static_cast<int64_t>(t.group) * p.sfb_sg;

            for (int kb = 0; kb < num_kb; ++kb) {
                const int k_valid = imin(p.k - kb * 128, 128);
                const StagedTile sa = stage_load(a_base + static_cast<int64_t>(kb) * 128 * p.a_sk, p.a_sm, p.a_sk,
                                                 rows_a, k_valid, false);
                const StagedTile sb = stage_load(b_base + static_cast<int64_t>(kb) * 128 * p.b_sk, p.b_sn, p.b_sk,
                                                 rows_b, k_valid, p.b_sk == 1);
                __syncthreads();                     // previous K block's fragments are consumed
                stage_store(a_lds, sa, p.a_sk, false);
                stage_store(b_lds, sb, p.b_sk, true);
                __syncthreads();

                float sa_v[MS];
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms)
                    sa_v[ms] = sfa_g[static_cast<int64_t>(sfa_row[ms]) * p.sfa_sm + static_cast<int64_t>(kb) * p.sfa_sk];

                const uint8_t* a_tile = a_lds + (wm * WM) * 128;
                const uint8_t* b_tile = b_lds + (wn * WN) * 128;
                v8i bf[NS];
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);

                if (p.sfb_gran_n == 128) {
                    const float sb_v = sfb_g[static_cast<int64_t>((t.n0 + wn * WN) / 128) * p.sfb_sn +
                                             static_cast<int64_t>(kb) * p.sfb_sk];
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        const v8i af = load_fragment(a_tile + ms * 2048, frag_off);
                        const float scale = sa_v[ms] * sb_v;
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns)
                            acc[ms][ns] += scale * mfma_fp8_k128(bf[ns], af);
                    }
                } else {
                    // Per-column SFB (recipe (1,1,128)): reference sm90_fp8_gemm_1d1d.cuh:303-311.
                    float sb_n[NS][4];
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        #pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int col = imin(t.n0 + wn * WN + (ns >> 1) * 32 + lg * 8 + (ns & 1) * 4 + r, p.n - 1);
                            sb_n[ns][r] = sfb_g[static_cast<int64_t>(col) * p.sfb_sn + static_cast<int64_t>(kb) * p.sfb_sk];
                        }
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        const v8i af = load_fragment(a_tile + ms * 2048, frag_off);
                        #pragma unroll
                        for (int ns = 0; ns < NS; ++ns) {
                            const v4f part = mfma_fp8_k128(bf[ns], af);
                            #pragma unroll
                            for (int r = 0; r < 4; ++r)
                                acc[ms][ns][r] += (sa_v[ms] * sb_n[ns][r]) * part[r];
                        }
                    }
                }
            }
        }
        __syncthreads();
        store_tile<MS, NS>(p, t, ad_group * p.d_sg, acc, t.m0 + wm * WM, t.n0 + wn * WN);
    }
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// Skinny kernel: dense GEMMs with M <= 16 * MS rows (batch-1 ... batch-32 decode: the m = 1 rows of the reference's dense sweep,
// tests/generators.py:119-121).  The tile kernels cannot win here: a 64-row tile wastes the matrix core, and with n / 32 ... n / 128
// workgroups of ONE latency chain each (0.27 - 0.66 us per K block) a CU has far too few weight bytes in flight.  This kernel is a
// weight stream: one workgroup owns 16 output columns over the WHOLE K, its 8 waves split K into 8 contiguous ranges, and every
// wave pulls its B fragments (16 weight rows x 128 K bytes = 2 KiB per K block) straight from HBM into registers -- no LDS ring, no
// barriers in the loop, eight K blocks of loads in flight per wave (128 KiB per CU).  A (<= 32 rows, L2-resident) is read the same
// way.  Operand roles as everywhere: weight rows feed the MFMA's row slot, so a lane owns ONE m and four n; promotion in K-block
// order inside a wave's range, the eight partial 16 x 16 tiles are summed in wave order through 8 KiB of LDS (bit-repeatable), and the
// result leaves through the shared epilogue (natural column order).  K permutation inside a block: lane group g supplies the
// 16-byte chunks g and g + 4 of both operands (the same bijection on both sides leaves the contraction unchanged), so each of the
// two loads of a fragment covers 64 contiguous bytes per row.
// ---------------------------------------------------------------------------------------------------------------
// (Round 3, negative: MS = 4 / 8 with CH = 2 / 1 K blocks per chunk for M <= 64 / 128 -- every workgroup re-reads the whole A through
// L2, 16 half-used cache lines per load instruction: 64 x 4096 x 7168 22.5 us against 21.0 on the stream tiles, 128 x 4096 x 7168 41.9
// against 21.3.  Not instantiated.)
// NSUB = 2 (round 3): a workgroup owns `p.skinny_cols` (17 .. 32) columns as two N-subtiles at n0 and n0 + 16, so that n / 16 column
// tiles just above the CU count (m = 1, 7168 x 16384: 448 tiles = 1.75 rounds) become ONE round of 28-column tiles.  The second subtile
// reaches into the next workgroup's columns: both compute the same bits for them (non-accumulating outputs only).
// COAL (round 5): the weight loads in a COALESCED lane order -- lane l fetches 16-byte chunk l & 7 of weight row (l >> 3) + 8 h, so one load
// instruction covers 8 rows x 128 contiguous bytes (8 cache lines, each asked for once) instead of 16 rows x 64 bytes with the 16 lanes of
// every quarter-wave on 16 different lines (64 requests per instruction) -- and reach the MFMA's operand layout (lane (r, g): row r, chunks
// g and g + 4) through 2 KiB of wave-private LDS at the moment they are consumed: two ds_write_b128 + two ds_read_b128 per K block in the
// chunk-swizzled image of the tile kernels (lds_chunk_offset), no barrier (a wave's LDS operations complete in order).  Same operands, same
// bits as the register-direct form.
// ACOAL: the same for the activation rows (they come from the L2, but their load instructions were as scattered: at m = 16 A is as many bytes
// per workgroup as the weights).
// E8 (round 6): packed UE8M0 scale words -- one per ROW of A and per weight ROW and four K blocks (G32: per 128-K block, byte g = MX block g) -- with
// the hardware-scaled MFMA accumulating a wave's K range in place (no promotion): the batch-1 .. 32 decode of the packed-scale path, which ran
// the 64 x 32 stream tile (15.5 us for 29 MB of weights against 7.6 here with FP32 scales).  A lane (r, g) holds row r of both operands'
// fragments, so its two scale words are those of A row 16 ms + r and of weight row n0 + 16 s + r.
template <int MS, int CH = 4, int NSUB = 1, bool COAL = false, bool ACOAL = false, bool E8 = false, bool G32 = false>
__global__ __launch_bounds__(512)
void dg_fp8_gemm_skinny_kernel(const GemmParams p) {
    static_assert(!ACOAL || COAL, "ACOAL shares the staging buffers' geometry with COAL");
    static_assert(!G32 || E8, "G32: a form of the packed-scale kernel");
    using SfT = std::conditional_t<E8, int, float>;
    constexpr int NW = 8;                                       // CH: K blocks per software-pipeline chunk (two chunks in flight)
    static_assert(MS * NSUB <= NW, "one wave per (M-subtile, N-subtile) sums the partial tiles");
    __shared__ float red[NW][MS * NSUB][256];
    __shared__ __attribute__((aligned(16))) uint8_t staging[COAL ? NW : 1][COAL ? 2 : 1][COAL ? 2048 : 16];
    __shared__ __attribute__((aligned(16))) uint8_t staging_a[ACOAL ? NW : 1][ACOAL ? 2 : 1][ACOAL ? 2048 : 16];
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * (NSUB == 1 || p.skinny_cols == 0 ? 16 * NSUB : p.skinny_cols);
    const int num_kb = p.k / 128;
    const int kb_begin = wave * num_kb / NW, kb_end = (wave + 1) * num_kb / NW;
    const uint8_t* b_ptr[NSUB];
    [[maybe_unused]] const uint8_t* b_ptr_hi[NSUB];             // COAL: rows (l >> 3) and (l >> 3) + 8
    const SfT* sfb_ptr[NSUB];
    [[maybe_unused]] const int st_write = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);        // + 1024 for the upper eight rows
    // fragment chunks of lane group g: g and g + 4 -- the matrix core's natural K order (registers 0-3 = K bytes 16 g .., 4-7 = 64 + 16 g ..), in
    // which the scale byte SUPPLIED by lane group g applies to MX block g of the row (profiles/r06_probe/g32_scale_byte_mapping.log): G32 as it is
    [[maybe_unused]] const int st_read_lo = r * 128 + ((g ^ (r & 7)) << 4), st_read_hi = r * 128 + (((g + 4) ^ (r & 7)) << 4);
    #pragma unroll
    for (int s = 0; s < NSUB; ++s) {
        if constexpr (COAL) {
            b_ptr[s] = p.b + static_cast<int64_t>(imin(n0 + s * 16 + (lane >> 3), p.n - 1)) * p.b_sn + (lane & 7) * 16;
            b_ptr_hi[s] = p.b + static_cast<int64_t>(imin(n0 + s * 16 + 8 + (lane >> 3), p.n - 1)) * p.b_sn + (lane & 7) * 16;
        } else {
            b_ptr[s] = p.b + static_cast<int64_t>(imin(n0 + s * 16 + r, p.n - 1)) * p.b_sn + g * 16;
        }
        // a lane's four output columns are n0 + 16 s + 4 g .. + 3 (n0 is a multiple of 4: they never straddle a 128-column scale block,
        // the 16-column subtile of the two-subtile form may)
        if constexpr (E8)
            sfb_ptr[s] = reinterpret_cast<const SfT*>(p.sfb) + static_cast<int64_t>(imin(n0 + s * 16 + r, p.n - 1)) * p.sfb_sn;
        else
            sfb_ptr[s] = reinterpret_cast<const SfT*>(p.sfb) + static_cast<int64_t>(imin(n0 + s * 16 + 4 * g, p.n - 1) / 128) * p.sfb_sn;
    }
    const uint8_t* a_ptr[MS];
    [[maybe_unused]] const uint8_t* a_ptr_hi[MS];               // ACOAL: rows (l >> 3) and (l >> 3) + 8 of the subtile, chunk l & 7
    const SfT* sfa_ptr[MS];
    #pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        const int row = imin(ms * 16 + r, p.m - 1);             // rows past m: a valid row's bytes, the result is never stored
        if constexpr (ACOAL) {
            a_ptr[ms] = p.a + static_cast<int64_t>(imin(ms * 16 + (lane >> 3), p.m - 1)) * p.a_sm + (lane & 7) * 16;
            a_ptr_hi[ms] = p.a + static_cast<int64_t>(imin(ms * 16 + 8 + (lane >> 3), p.m - 1)) * p.a_sm + (lane & 7) * 16;
        } else {
            a_ptr[ms] = p.a + static_cast<int64_t>(row) * p.a_sm + g * 16;
        }
        sfa_ptr[ms] = reinterpret_cast<const SfT*>(p.sfa) + static_cast<int64_t>(row) * p.sfa_sm;
    }

    struct Chunk { v4i b[CH][NSUB][2]; v4i a[MS][CH][2]; SfT sa[MS][CH]; SfT sb[CH][NSUB]; };
    auto load_chunk = [&](Chunk& c, int kb0) {
        #pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int kb = imin(kb0 + j, num_kb - 1);          // past the range: re-read the last block (never used)
            const int64_t off = static_cast<int64_t>(kb) * 128;
            const int64_t ksf = E8 && !G32 ? kb >> 2 : kb;     // row of the scale tensors along K (packed words: one per K quad; G32: per K block)
            #pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                c.b[j][s][0] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(b_ptr[s] + off));
                if constexpr (COAL) c.b[j][s][1] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(b_ptr_hi[s] + off));
                else c.b[j][s][1] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(b_ptr[s] + off + 64));
                c.sb[j][s] = sfb_ptr[s][ksf * p.sfb_sk];
            }
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms) {
                c.a[ms][j][0] = *reinterpret_cast<const v4i*>(a_ptr[ms] + off);
                if constexpr (ACOAL) c.a[ms][j][1] = *reinterpret_cast<const v4i*>(a_ptr_hi[ms] + off);
                else c.a[ms][j][1] = *reinterpret_cast<const v4i*>(a_ptr[ms] + off + 64);
                c.sa[ms][j] = sfa_ptr[ms][ksf * p.sfa_sk];
            }
        }
    };
    v4f acc[MS][NSUB];
    #pragma unroll
    for (int ms = 0; ms < MS; ++ms)
        #pragma unroll
        for (int s = 0; s < NSUB; ++s)
            acc[ms][s] = v4f{0.f, 0.f, 0.f, 0.f};
    auto compute_chunk = [&](const Chunk& c, int kb0) {
        #pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (kb0 + j < kb_end) {                             // wave-uniform
                v8i afs[MS];
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    v4i a_lo = c.a[ms][j][0], a_hi = c.a[ms][j][1];
                    if constexpr (ACOAL) {
                        uint8_t* st = staging_a[wave][(j * MS + ms) & 1];
                        *reinterpret_cast<v4i*>(st + st_write) = a_lo;
                        *reinterpret_cast<v4i*>(st + st_write + 1024) = a_hi;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        a_lo = *reinterpret_cast<const v4i*>(st + st_read_lo);
                        a_hi = *reinterpret_cast<const v4i*>(st + st_read_hi);
                    }
                    afs[ms] = __builtin_shufflevector(a_lo, a_hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
                #pragma unroll
                for (int s = 0; s < NSUB; ++s) {
                    v4i b_lo = c.b[j][s][0], b_hi = c.b[j][s][1];
                    if constexpr (COAL) {
                        uint8_t* st = staging[wave][(j * NSUB + s) & 1];
                        *reinterpret_cast<v4i*>(st + st_write) = b_lo;
                        *reinterpret_cast<v4i*>(st + st_write + 1024) = b_hi;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // other lanes' bytes: keep the reads behind the writes
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        b_lo = *reinterpret_cast<const v4i*>(st + st_read_lo);
                        b_hi = *reinterpret_cast<const v4i*>(st + st_read_hi);
                    }
                    const v8i bf = __builtin_shufflevector(b_lo, b_hi, 0, 1, 2, 3, 4, 5, 6, 7);
                    #pragma unroll
                    for (int ms = 0; ms < MS; ++ms) {
                        if constexpr (E8) {
                            // this block's byte of the words (G32: the lane group's byte of the block's words) into byte 0; in-place accumulation
                            const int shift = G32 ? g * 8 : ((kb0 + j) & 3) * 8;
                            acc[ms][s] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                                bf, afs[ms], acc[ms][s], 0, 0, 0, static_cast<int>(static_cast<unsigned>(c.sb[j][s]) >> shift), 0,
                                static_cast<int>(static_cast<unsigned>(c.sa[ms][j]) >> shift));
                        } else {
                        const v4f part = mfma_fp8_k128(bf, afs[ms]);
                        const float scale = c.sa[ms][j] * c.sb[j][s];
                        #pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[ms][s][e] = __builtin_fmaf(scale, part[e], acc[ms][s][e]);
                        }
                    }
                }
            }
        }
    };
    Chunk c0, c1;
    if (kb_begin < kb_end)
        load_chunk(c0, kb_begin);
    for (int kb = kb_begin; kb < kb_end; kb += 2 * CH) {
        if (kb + CH < kb_end) load_chunk(c1, kb + CH);
        compute_chunk(c0, kb);
        if (kb + 2 * CH < kb_end) load_chunk(c0, kb + 2 * CH);
        if (kb + CH < kb_end) compute_chunk(c1, kb + CH);
    }
    #pragma unroll
    for (int ms = 0; ms < MS; ++ms)
        #pragma unroll
        for (int s = 0; s < NSUB; ++s)
            *reinterpret_cast<v4f*>(&red[wave][ms * NSUB + s][lane * 4]) = acc[ms][s];
    __syncthreads();
    if (wave < MS * NSUB) {
        v4f sum = *reinterpret_cast<const v4f*>(&red[0][wave][lane * 4]);
        #pragma unroll
        for (int w = 1; w < NW; ++w)
            sum += *reinterpret_cast<const v4f*>(&red[w][wave][lane * 4]);
        const int ms = wave / NSUB, sub = wave % NSUB;
        if (n0 + sub * 16 < p.n) {
            Tile t;
            t.m0 = 0; t.n0 = n0 + sub * 16; t.group = 0; t.m_begin = 0; t.m_end = p.m; t.zero_from = t.zero_to = p.m; t.valid = true; t.second_pass = false;
            v4f out[1][1] = {{sum}};
            store_tile<1, 1, false, false, true>(p, t, 0, out, ms * 16, n0 + sub * 16);
        }
    }
}

// Tile tables of the contiguous layout (M alignment 128): one pass over grouped_layout's 128-row blocks (the group of a block is read from
// its first row, reference scheduler/gemm.cuh:160-162) cuts every group's run of blocks into 256-row tiles plus at most one 128-row
// remainder; blocks of padding rows (-1) go with the remainders (they are stored as zeros).  big[0] / rem[0] = counts, [1 + i] = first
// rows.  One workgroup; the block ids pass through LDS so that the serial scan does not chain dependent global loads.  Runs on the
// GEMM's stream in front of it: the 256-row tiles then never straddle two groups (the fixed grid walked such tiles twice), and the
// remainders -- too few to fill the chip -- are cut along K (launch_contiguous_tabled in dg_api.hip).
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_build_contiguous_tile_table_kernel(const int32_t* __restrict__ layout, int m, int32_t* __restrict__ big, int32_t* __restrict__ rem) {
    __shared__ int group_of[512];
    const int nb = (m + 127) / 128;
    for (int b = threadIdx.x; b < nb; b += 256)
        group_of[b] = layout[b * 128];
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    int big_n = 0, rem_n = 0, b = 0;
    while (b < nb) {
        const int g = group_of[b];
        if (g < 0) {
            rem[1 + rem_n++] = b * 128;
            ++b;
            continue;
        }
        int run = 1;
        while (b + run < nb && group_of[b + run] == g)
            ++run;
        for (int j = 0; j + 1 < run; j += 2)
            big[1 + big_n++] = (b + j) * 128;
        if (run & 1)
            rem[1 + rem_n++] = (b + run - 1) * 128;
        b += run;
    }
    big[0] = big_n;
    rem[0] = rem_n;
}
#endif

// SF layout kernel: [batches, mn, sf_k] row-major FP32 -> MN-major with mn padded to a multiple of 4 floats
// (semantics of transpose_fp32, deep_gemm/include/deep_gemm/impls/smxx_layout.cuh:12-50).  One block moves a
// 64 (mn) x 64 (sf_k) patch through LDS so that both the read (along sf_k) and the write (along mn) are coalesced.
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_transpose_sf_fp32_kernel(const float* __restrict__ sf, float* __restrict__ out, int mn, int sf_k, int aligned_mn) {
    __shared__ float patch[64][65];
    const int batch = blockIdx.z;
    const int mn0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const float* src = sf + static_cast<int64_t>(batch) * mn * sf_k;
    float* dst = out + static_cast<int64_t>(batch) * aligned_mn * sf_k;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int row = mn0 + i, col = k0 + tx;
        if (row < mn && col < sf_k)
            patch[i][tx] = src[static_cast<int64_t>(row) * sf_k + col];
    }
    __syncthreads();
    for (int j = ty; j < 64; j += 4) {
        const int col = k0 + j, row = mn0 + tx;
        if (row < mn && col < sf_k)
            dst[static_cast<int64_t>(col) * aligned_mn + row] = patch[tx][j];
    }
}
#endif

// The same transpose for the common case sf_k % 4 == 0 with 16-byte aligned rows (K a multiple of 512: every DeepSeek-V3 shape): one
// thread takes FOUR consecutive K blocks of one row with a single 16-byte load and writes them to four MN-major rows -- every load of the
// launch is independent (the patch kernel above walks 16 dependent iterations per thread and took ~7 us for 0.9 MB in front of a 90 us
// GEMM: this launch is pure latency), the writes run along mn (coalesced).  Round 5.
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_transpose_sf_fp32_vec4_kernel(const float* __restrict__ sf, float* __restrict__ out, int mn, int sf_k, int aligned_mn) {
    const int batch = blockIdx.z;
    const int row = blockIdx.x * 256 + threadIdx.x, kq = blockIdx.y;
    if (row >= mn)
        return;
    const float* src = sf + static_cast<int64_t>(batch) * mn * sf_k + static_cast<int64_t>(row) * sf_k + 4 * kq;
    float* dst = out + static_cast<int64_t>(batch) * aligned_mn * sf_k + static_cast<int64_t>(4 * kq) * aligned_mn + row;
    const v4f v = *reinterpret_cast<const v4f*>(src);
    dst[0] = v[0];
    dst[aligned_mn] = v[1];
    dst[2 * static_cast<int64_t>(aligned_mn)] = v[2];
    dst[3 * static_cast<int64_t>(aligned_mn)] = v[3];
}
#endif

// SF packing kernel: FP32 power-of-two scales [batches, ceil(mn / gran_mn), sf_k] (any strides) -> packed UE8M0 words, MN-major:
// word (row, kq) = exponent bytes of K blocks 4 kq .. 4 kq + 3 of source row `row / gran_mn` (byte j = bits 30..23 of
// sf[row / gran_mn][4 kq + j], blocks past sf_k are zero bytes), stored at out[batch * packed_k * aligned_mn + kq * aligned_mn + row].
// Semantics of the reference's transpose_and_pack_fp32_into_ue8m0 / pack_fp32_into_ue8m0 (impls/smxx_layout.cuh:56,148) and of its
// torch twin (jit_kernels/impls/smxx_layout.hpp:156-179).  gran_mn > 1 fuses the row broadcast the reference performs with
// index_select in front of the pack (csrc/apis/layout.hpp:52-54: per-128-row scales -> one word per row) -- no temporary.
// psum_layout != nullptr (smxx_layout.cuh:76-94): rows outside every group's range [align(end[g-1], m_alignment), end[g]) -- the
// uninitialised gap rows of the psum contiguous layout -- get zero words (a finite scale code; 0xff would be NaN).
// A block packs a 64 (mn) x 64 (sf_k) patch through LDS: the read runs along whichever input axis has unit stride, the write along mn.
struct PackSfArgs {
    const float* sf; int32_t* out;
    int mn, sf_k, aligned_mn;
    int64_t stride_b, stride_mn, stride_k;
    int gran_mn;
    const int32_t* psum_layout; int num_psum_groups, m_alignment;
    int blocks_mn, batches;                     // work items of this tensor: blocks_mn (256 rows each) x packed_k x batches
};

// One output word per thread, 256 consecutive rows per block (the write runs along mn: coalesced; the four reads of a word are four
// independent loads -- this kernel sits in front of a GEMM on the same stream and its cost is latency: the first version, a 64 x 64 patch
// through LDS with one dependent load per loop iteration, took 8 us for 0.9 MB).
__device__ __forceinline__ void pack_sf_ue8m0_words(const PackSfArgs& a, int block_mn, int kq, int batch) {
    const int row = block_mn * 256 + static_cast<int>(threadIdx.x);
    if (row >= a.mn)
        return;
    const int packed_k = (a.sf_k + 3) / 4;
    const float* src = a.sf + static_cast<int64_t>(batch) * a.stride_b + static_cast<int64_t>(row / a.gran_mn) * a.stride_mn;
    uint32_t bits[4];
    #pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = 4 * kq + j;
        bits[j] = col < a.sf_k ? __float_as_uint(src[static_cast<int64_t>(col) * a.stride_k]) : 0u;
    }
    bool valid = true;
    if (a.psum_layout != nullptr) {
        valid = false;
        int start = 0;
        for (int g = 0; g < a.num_psum_groups; ++g) {
            const int end = a.psum_layout[g];
            valid = valid || (row >= start && row < end);
            start = (end + a.m_alignment - 1) / a.m_alignment * a.m_alignment;
        }
    }
    const uint32_t word = ((bits[0] >> 23) & 0xffu) | (((bits[1] >> 23) & 0xffu) << 8) | (((bits[2] >> 23) & 0xffu) << 16) | ((bits[3] >> 23) << 24);
    a.out[(static_cast<int64_t>(batch) * packed_k + kq) * a.aligned_mn + row] = valid ? static_cast<int32_t>(word) : 0;
}

// One launch packs up to two scale tensors (the SFA / SFB pair of a GEMM call in the 'sm100' scaling-factor mode: one kernel boundary in
// front of the GEMM instead of two): block ids [0, work items of a) belong to `a`, the rest to `b`.
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_pack_sf_ue8m0_kernel(const PackSfArgs a, const PackSfArgs b) {
    const int items_a = a.blocks_mn * ((a.sf_k + 3) / 4) * a.batches;
    const bool second = static_cast<int>(blockIdx.x) >= items_a;
    const PackSfArgs& t = second ? b : a;
    int id = static_cast<int>(blockIdx.x) - (second ? items_a : 0);
    const int bx = id % t.blocks_mn;
    id /= t.blocks_mn;
    const int packed_k = (t.sf_k + 3) / 4;
    pack_sf_ue8m0_words(t, bx, id % packed_k, id / packed_k);
}
#endif

// Fused per-token quantiser (the producer side of operand A): BF16 [m, n] -> e4m3fn [m, n] + one FP32 scale per
// 1 x 128 block, the arithmetic of per_token_cast_to_fp8 (deep_gemm/utils/math.py:26-38): amax over the block (ragged
// tail zero-padded), sf = max(amax, 1e-4) / 448, optionally rounded up to a power of two (ceil_to_ue8m0, :13-16),
// q = e4m3fn_rne(float(x) * (1.0f / sf)).  HBM-bound: 2 bytes read + 1 byte written per element, one pass (the torch
// expression makes five).  16 lanes own one block (8 elements = one 16-byte load each), a wave four blocks; the amax is a
// 4-step DPP-width butterfly inside the 16-lane row.  The scale lands either row-major [m, ceil(n/128)] (reference
// return value) or directly in the GEMM's MN-major layout, which saves the transpose launch of the GEMM call (PMC on
// 16384 x 7168: FETCH 229 MB = the input; WRITE 142 MB vs 121 MB of payload -- the 4-byte MN-major scale writes cost
// partial lines; walking down K-block columns instead to make them adjacent was slower, 79 vs 68 us: the reads lose
// their row locality).
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_per_token_cast_to_fp8_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ q, float* __restrict__ sf,
                                     int m, int n, int64_t x_sm, int64_t q_sm, int64_t sf_sm, int64_t sf_sk, int use_ue8m0) {
    const int lane = threadIdx.x & 63, sub = lane & 15;
    const int blocks_per_row = (n + 127) / 128;
    const int64_t total = static_cast<int64_t>(m) * blocks_per_row;
    const int64_t wave_first = (static_cast<int64_t>(blockIdx.x) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
    const int64_t step = static_cast<int64_t>(gridDim.x) * (blockDim.x >> 6) * 4;
    const bool vec_ok = (n % 8 == 0) && (x_sm % 8 == 0) && (q_sm % 8 == 0) &&
                        (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(q) % 8 == 0);
    for (int64_t base = wave_first; base < total; base += step) {
        // every lane of the wave runs the same number of iterations (the butterfly needs all 16 lanes of a row alive)
        const int64_t c = base + (lane >> 4);
        const bool live = c < total;
        const int row = live ? static_cast<int>(c / blocks_per_row) : 0;
        const int kb = live ? static_cast<int>(c % blocks_per_row) : 0;
        const int col = kb * 128 + sub * 8;
        float v[8];
        if (live && vec_ok && col + 8 <= n) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + row * x_sm + col);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = __uint_as_float(w[i] << 16);
                v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            }
        } else {
            #pragma unroll
            for (int i = 0; i < 8; ++i)
                v[i] = (live && col + i < n) ? __uint_as_float(static_cast<uint32_t>(x[row * x_sm + col + i]) << 16) : 0.f;
        }
        float amax = 0.f;
        #pragma unroll
        for (int i = 0; i < 8; ++i)
            amax = fmaxf(amax, fabsf(v[i]));
        #pragma unroll
        for (int d = 1; d < 16; d <<= 1)
            amax = fmaxf(amax, __shfl_xor(amax, d, 64));
        float scale = fmaxf(amax, 1e-4f) / 448.0f;
        if (use_ue8m0) {
            const uint32_t bits = __float_as_uint(scale);
            uint32_t e = ((bits >> 23) & 0xffu) + ((bits & 0x7fffffu) != 0 ? 1u : 0u);
            e = e < 1u ? 1u : (e > 254u ? 254u : e);
            scale = __uint_as_float(e << 23);
        }
        const float inv = 1.0f / scale;
        if (!live)
            continue;
        uint32_t packed[2];
        #pragma unroll
        for (int i = 0; i < 2; ++i) {
            int word = 0;
            word = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i] * inv, v[4 * i + 1] * inv, word, false);
            word = __builtin_amdgcn_cvt_pk_fp8_f32(v[4 * i + 2] * inv, v[4 * i + 3] * inv, word, true);
            packed[i] = static_cast<uint32_t>(word);
        }
        if (vec_ok && col + 8 <= n) {
            *reinterpret_cast<uint2*>(q + row * q_sm + col) = make_uint2(packed[0], packed[1]);
        } else {
            #pragma unroll
            for (int i = 0; i < 8; ++i)
                if (col + i < n)
                    q[row * q_sm + col + i] = static_cast<uint8_t>(packed[i >> 2] >> (8 * (i & 3)));
        }
        if (sub == 0)
            sf[row * sf_sm + kb * sf_sk] = scale;
    }
}
#endif

// Fused block quantisers of the weight / wgrad side: one 256-thread workgroup casts a 128 x 128 patch of a BF16 matrix in
// one pass (each thread holds 8 rows x 8 columns).  PER_CHANNEL = false: one scale per 128 x 128 block
// (per_block_cast_to_fp8, deep_gemm/utils/math.py:51-61; ragged edges count as zeros); PER_CHANNEL = true: one scale per
// column per 128-row block (per_channel_cast_to_fp8, :41-48: the operand form of the K-grouped GEMM).  Same arithmetic as
// the per-token kernel above; HBM-bound, 3 bytes per element.
template <bool PER_CHANNEL>
__global__ __launch_bounds__(256)
void dg_block_cast_to_fp8_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ q, float* __restrict__ sf,
                                 int rows, int cols, int64_t x_sr, int64_t q_sr, int64_t sf_sr, int64_t sf_sc, int use_ue8m0) {
    __shared__ float red[4][128];
    const int tid = threadIdx.x, cg = tid & 15, rg = tid >> 4;           // column group (8 columns), row group (of 16)
    const int row0 = blockIdx.y * 128, col0 = blockIdx.x * 128 + cg * 8;
    const bool vec_ok = (cols % 8 == 0) && (x_sr % 8 == 0) && (q_sr % 8 == 0) &&
                        (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(q) % 8 == 0);
    float v[8][8];
    #pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = row0 + i * 16 + rg;
        if (row < rows && vec_ok && col0 + 8 <= cols) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + row * x_sr + col0);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
            #pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[i][2 * j] = __uint_as_float(w[j] << 16);
                v[i][2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
            }
        } else {
            #pragma unroll
            for (int j = 0; j < 8; ++j)
                v[i][j] = (row < rows && col0 + j < cols) ? __uint_as_float(static_cast<uint32_t>(x[row * x_sr + col0 + j]) << 16) : 0.f;
        }
    }
    // per-column amax over this thread's 8 rows, then over the 16 row groups: lanes 16 / 32 apart, then the 4 waves
    float cmax[8];
    #pragma unroll
    for (int j = 0; j < 8; ++j) {
        float a = 0.f;
        #pragma unroll
        for (int i = 0; i < 8; ++i)
            a = fmaxf(a, fabsf(v[i][j]));
        a = fmaxf(a, __shfl_xor(a, 16, 64));
        a = fmaxf(a, __shfl_xor(a, 32, 64));
        cmax[j] = a;
    }
    if ((tid & 63) < 16) {
        #pragma unroll
        for (int j = 0; j < 8; ++j)
            red[tid >> 6][cg * 8 + j] = cmax[j];
    }
    __syncthreads();
    float scale[8];
    if constexpr (PER_CHANNEL) {
        #pragma unroll
        for (int j = 0; j < 8; ++j)
            scale[j] = fmaxf(fmaxf(red[0][cg * 8 + j], red[1][cg * 8 + j]), fmaxf(red[2][cg * 8 + j], red[3][cg * 8 + j]));
    } else {
        float a = 0.f;
        for (int c = tid & 63; c < 512; c += 64)            // every wave reduces all 4 x 128 partial maxima
            a = fmaxf(a, red[c >> 7][c & 127]);
        #pragma unroll
        for (int d = 1; d < 64; d <<= 1)
            a = fmaxf(a, __shfl_xor(a, d, 64));
        #pragma unroll
        for (int j = 0; j < 8; ++j)
            scale[j] = a;
    }
    float inv[8];
    #pragma unroll
    for (int j = 0; j < 8; ++j) {
        float sc = fmaxf(scale[j], 1e-4f) / 448.0f;
        if (use_ue8m0) {
            const uint32_t bits = __float_as_uint(sc);
            uint32_t e = ((bits >> 23) & 0xffu) + ((bits & 0x7fffffu) != 0 ? 1u : 0u);
            e = e < 1u ? 1u : (e > 254u ? 254u : e);
            sc = __uint_as_float(e << 23);
        }
        scale[j] = sc;
        inv[j] = 1.0f / sc;
    }
    #pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = row0 + i * 16 + rg;
        if (row >= rows)
            continue;
        uint32_t packed[2];
        #pragma unroll
        for (int h = 0; h < 2; ++h) {
            int word = 0;
            word = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4 * h] * inv[4 * h], v[i][4 * h + 1] * inv[4 * h + 1], word, false);
            word = __builtin_amdgcn_cvt_pk_fp8_f32(v[i][4 * h + 2] * inv[4 * h + 2], v[i][4 * h + 3] * inv[4 * h + 3], word, true);
            packed[h] = static_cast<uint32_t>(word);
        }
        if (vec_ok && col0 + 8 <= cols) {
            *reinterpret_cast<uint2*>(q + row * q_sr + col0) = make_uint2(packed[0], packed[1]);
        } else {
            #pragma unroll
            for (int j = 0; j < 8; ++j)
                if (col0 + j < cols)
                    q[row * q_sr + col0 + j] = static_cast<uint8_t>(packed[j >> 2] >> (8 * (j & 3)));
        }
    }
    if constexpr (PER_CHANNEL) {
        if (rg == 0) {
            #pragma unroll
            for (int j = 0; j < 8; ++j)
                if (col0 + j < cols)
                    sf[blockIdx.y * sf_sr + (col0 + j) * sf_sc] = scale[j];
        }
    } else {
        if (tid == 0)
            sf[blockIdx.y * sf_sr + blockIdx.x * sf_sc] = scale[0];
    }
}

// Operand re-majoring: dst[c][r] = src[r][c] for 1-byte elements (an MN-major FP8 operand -> the K-major form the
// LDS-DMA kernels consume).  64 x 64 byte patches through LDS; both the global read (16 bytes along c per lane) and the
// global write (16 bytes along r per lane) are coalesced 16-byte vectors.  HBM-bound: 2 bytes of traffic per element.
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_transpose_bytes_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int rows, int cols,
                               int64_t src_ld, int64_t dst_ld, int64_t src_batch, int64_t dst_batch) {
    __shared__ uint8_t patch[64][64 + 16];                  // +16: rows stay 16-byte aligned, column walks spread over banks
    const uint8_t* s = src + static_cast<int64_t>(blockIdx.z) * src_batch;
    uint8_t* d = dst + static_cast<int64_t>(blockIdx.z) * dst_batch;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tr = threadIdx.x >> 2, tc = (threadIdx.x & 3) * 16;          // 64 rows x 4 vectors of 16 bytes
    {
        const int r = r0 + tr, c = c0 + tc;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        const uint8_t* ptr = s + static_cast<int64_t>(r) * src_ld + c;
        if (r < rows) {
            if (c + 16 <= cols && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0) {
                v = *reinterpret_cast<const uint4*>(ptr);
            } else {
                uint32_t w[4] = {0u, 0u, 0u, 0u};
                for (int e = 0; e < 16; ++e)
                    if (c + e < cols)
                        w[e >> 2] |= static_cast<uint32_t>(ptr[e]) << ((e & 3) * 8);
                v = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        *reinterpret_cast<uint4*>(&patch[tr][tc]) = v;
    }
    __syncthreads();
    {
        // output row = source column c0 + tr, 16 consecutive source rows r0 + tc .. + 15
        const int c = c0 + tr, r = r0 + tc;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        #pragma unroll
        for (int e = 0; e < 16; ++e)
            w[e >> 2] |= static_cast<uint32_t>(patch[tc + e][tr]) << ((e & 3) * 8);
        if (c < cols) {
            uint8_t* ptr = d + static_cast<int64_t>(c) * dst_ld + r;
            if (r + 16 <= rows && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0) {
                *reinterpret_cast<uint4*>(ptr) = make_uint4(w[0], w[1], w[2], w[3]);
            } else {
                for (int e = 0; e < 16; ++e)
                    if (r + e < rows)
                        ptr[e] = static_cast<uint8_t>(w[e >> 2] >> ((e & 3) * 8));
            }
        }
    }
}
#endif

}  // namespace dg
