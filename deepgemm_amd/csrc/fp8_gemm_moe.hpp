// Fused expert MLP hand-off (the single-GPU half of the reference's Mega-MoE kernel): the first grouped GEMM of an expert MLP with the
// SwiGLU activation and the per-token FP8 re-quantisation for the SECOND GEMM in its epilogue -- the BF16 intermediate [rows, 2 I] never
// goes to memory; what leaves the kernel is GEMM2's operand pair (A_fp8 [G, M, I], SFA [G, I / 128, M] FP32, MN-major).
//
// Reference: the L1 -> L2 hand-off inside deep_gemm/include/deep_gemm/impls/sm100_fp8_fp4_mega_moe.cuh (GEMM1 epilogue: SwiGLU on the
// gate / up halves, amax, re-quantisation, L2 operand written for the second UMMA pipeline), weight layout
// deep_gemm/mega/__init__.py:115-151 (`transform_weights_for_mega_moe`: gate and up rows interleaved so that one output tile holds
// both halves of the same intermediate columns).  The dispatch / combine over NVLink of that kernel is NOT here (it needs a multi-GPU
// node: deepgemm_amd/ep.py does the exchange with RCCL all-to-alls around this operator).
//
// MI355X form.  At decode sizes this operator is a WEIGHT STREAM (8 experts x 4096 x 7168 bytes for <= 512 tokens): what matters is
// HBM bytes in flight on ALL 256 CUs -- one CU sustains about 30 GB/s from HBM whatever the kernel (a 64 x 256 tile version of this
// kernel with 128 workgroups ran at 3.0 TB/s, the plain 64 x 128 stream kernel with 256 workgroups and a 6-deep ring at 5.0 TB/s;
// profiles/r03_mlp/NOTES.md).  So the tile is the plain stream kernel's 64 x 128 with its 6-stage ring, and the weight layout
// (deepgemm_amd/mega.py) interleaves gate and up rows in BLOCKS OF 64: tile columns [0, 64) = gate rows 64 j .. 64 j + 63, [64, 128) =
// the up rows of the same j.  A 1 x 128 quantisation block of the intermediate then spans TWO tiles (j = 2 i and 2 i + 1), computed
// by two workgroups on two CUs: they exchange their per-row amax through a small global workspace (one 32-bit slot per row and tile:
// bit 31 = valid, the reader clears the slot again, so the workspace is zeroed once and then serves every launch and every hipGraph
// replay).  The partner of tile t is tile t ^ 1 -- adjacent in launch order, handled by the neighbouring workgroup in the same
// iteration of the persistent loop; hardware dispatches a kernel's workgroups in order, so a waiting workgroup's partner is already
// resident or is the next one to be placed: no deadlock, whatever else runs on the device.  The FP32 weight scales stay the 128 x 128
// blocks the weights were quantised with: gate rows 64 j .. belong to scale block j / 2 of the gate half, likewise up -- two SFB
// values per tile, as laid out by mega.py (scale rows interleaved [gate 0, up 0, gate 1, up 1, ...]).
// Kernel: 4 waves, wave tile 64 x 32 (waves 0, 1 hold gate columns, waves 2, 3 the matching up columns), K loop of
// stream_kernel_body (fp8_gemm_kernels.hpp): promotion by FMA in K-block order, bit-identical to the plain masked kernel.  Epilogue,
// after the K loop has released the LDS:
//   1. every accumulator is rounded to BF16 (the value the unfused pipeline stores and reloads);
//   2. gate and up waves swap halves through the LDS (a gate wave finishes M-subtiles 0, 1 of its columns, the up wave of the same columns
//      subtiles 2, 3 -- round 5: the arithmetic on all four SIMDs) and read them at the same (lane, register) position: y = silu(g) * u in
//      FP32 (optionally clamped: g <= c, |u| <= c), rounded to BF16;
//   3. amax over the row's 64 values of this tile: 8 in a lane, 4 lanes of a row (lane bits 4, 5), 2 gate waves (LDS); then the
//      exchange with the partner tile (wave 0, one lane per row): amax over the 128-wide block;
//   4. scale = max(amax, 1e-4) * (1 / 448) (optionally rounded up to a power of two), q = e4m3(y * (1 / scale)) -- the arithmetic of
//      per_token_cast_to_fp8 (deep_gemm/utils/math.py:26-38) as torch executes it on a device;
//   5. one 8-byte store per lane and M-subtile (8 consecutive intermediate columns), one scale per row and PAIR of tiles.
// Bit-exact against "masked GEMM -> BF16 -> torch SwiGLU -> reference per_token_cast_to_fp8" (tests/test_mega_gpu.py).
#pragma once
#include "fp8_gemm_kernels.hpp"

#ifndef DG_SWIGLU_B_AUX
#define DG_SWIGLU_B_AUX 2
#endif

namespace dg {

struct SwigluOut {
    uint8_t* q;             // [G, m_max, I] e4m3
    float* sf;              // [G, I / 128, sf_sk] FP32: element (g, kb, m) at sf[g * sf_sg + kb * sf_sk + m]
    int64_t q_sg, q_sm, sf_sg, sf_sk;
    float clamp;            // <= 0: none
    int use_ue8m0;
    uint32_t* amax_ws;      // [tiles][64] exchange slots, all zero between launches
    uint32_t* errors;       // workspace header word 0: number of exchange waits that timed out (the rows involved get NaN scales)
    long long timeout_ticks;    // bound of the partner wait in s_memrealtime ticks (100 MHz); reference: comm/barrier.cuh:12,36-40 (60 s)
    const float* row_weight;    // optional [G, m_max] FP32 (element (g, m) at row_weight[g * rw_sg + m]): the row's top-k routing weight, applied
    int64_t rw_sg;              // to the SwiGLU output before the re-quantisation as the reference kernel does (sm100_fp8_fp4_mega_moe.cuh:1019)
    int fault;              // test hook (environment DG_TEST_SWIGLU_FAULT, tests only): 1 = odd tiles never publish their amax -> their partners time out
};

template <int STAGES>
__device__ __forceinline__ void stream_swiglu_kernel_body(const GemmParams& p, const SwigluOut& o) {
    constexpr int BM = 64, BN = 128, NW = 4, WM = 64, WN = 32, MS = 4, NS = 2;
    constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128;
    // scales: one 16-byte-per-lane piece = the 64 row scales of FOUR K blocks, one 4-byte piece = their (gate, up) SFB pairs, in a ring
    // of four group slots behind the stages (stream_kernel_body says why); not counted in PIECES (the counted waits get stricter)
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int SFG_OFF = STAGES * STAGE_BYTES, SFG_SLOT = 1024 + 256, SFG_SLOTS = 4;
    constexpr int LDS_BYTES = SFG_OFF + SFG_SLOTS * SFG_SLOT;
    constexpr int A_ITERS = BM / 8 / NW, B_ITERS = BN / 8 / NW;
    constexpr int PIECES = A_ITERS + B_ITERS;
    static_assert(STAGES <= 4 * (SFG_SLOTS - 1), "a group slot is refilled only after its last reader");
    constexpr unsigned OOB = 0x80000000u;
    static_assert((STAGES - 1) * PIECES < 64 && LDS_BYTES <= 160 * 1024 && STAGES >= 3, "ring geometry");
    static_assert(2 * (MS * NS * 4) * 64 * 4 + 3 * 64 * 4 <= LDS_BYTES, "the epilogue exchange fits in the ring's LDS");

    __shared__ __attribute__((aligned(1024))) uint8_t lds[LDS_BYTES];

    const int lane = threadIdx.x & 63, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wn = wave;
    const int num_kb = p.k / 128;
    const int piece_row = lane >> 3;
    const int src_chunk = (lane & 7) ^ piece_row;
    const int frag_off = (lane & 15) * 128 + ((((lane >> 4) ^ (lane & 7))) << 4);
    const int lda = static_cast<int>(p.a_sm), ldb = static_cast<int>(p.b_sn);
    auto a_unit_row = [](int u) { return (u / (WM / 8)) * WM + (u & 1) * 8 * MS + ((u % (WM / 8)) >> 1); };
    const int a_voff = piece_row * MS * lda + src_chunk * 16;
    const int b_voff = b_row_perm<WN>(wave * 8 + piece_row) * ldb + src_chunk * 16;

    // Persistent walk over the groups (masked_m lives on the device).  Order inside a group: the two tiles of a quantisation block
    // adjacent (tile_id ^ 1 is the partner; every group holds an even number of tiles), then M tiles, then blocks.
    struct { int m0, n0, m_end; } t;
    int walk_group = 0, walk_base = 0;
    const int num_launched = gridDim.x;
    for (int tile_id = blockIdx.x;; tile_id += num_launched) {
        int nmt;
        while (true) {
            if (walk_group >= p.num_groups)
                return;
            t.m_end = imin(p.layout[walk_group], p.m);
            nmt = (t.m_end + BM - 1) / BM;
            if (tile_id < walk_base + nmt * p.num_n_tiles)
                break;
            walk_base += nmt * p.num_n_tiles;
            ++walk_group;
        }
        {
            const int local = tile_id - walk_base, rest = local >> 1;
            t.m0 = (rest % nmt) * BM;
            t.n0 = (2 * (rest / nmt) + (local & 1)) * BN;
        }
        const int64_t group = walk_group;

        v4f acc[MS][NS];
        #pragma unroll
        for (int ms = 0; ms < MS; ++ms)
            #pragma unroll
            for (int ns = 0; ns < NS; ++ns)
                acc[ms][ns] = v4f{0.f, 0.f, 0.f, 0.f};

        if (t.m_end > t.m0) {
            const uint8_t* a_base = p.a + group * p.a_sg + static_cast<int64_t>(t.m0) * p.a_sm;
            const uint8_t* b_base = p.b + group * p.b_sg + static_cast<int64_t>(t.n0) * p.b_sn;
            const int a_rows = imin(t.m_end - t.m0, BM);
            const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a_base), 0, (a_rows - 1) * lda + p.k, 0x00020000);
            const auto b_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(b_base), 0, (BN - 1) * ldb + p.k, 0x00020000);
            const int sfa_kb_stride = static_cast<int>(p.sfa_sk) * 4, sfb_kb_stride = static_cast<int>(p.sfb_sk) * 4;
            float* sfa_tile = const_cast<float*>(p.sfa) + group * p.sfa_sg + t.m0;
            const int sfa_rows = imin(p.m - t.m0, BM);
            const auto sfa_rsrc = __builtin_amdgcn_make_buffer_rsrc(sfa_tile, 0, (num_kb - 1) * sfa_kb_stride + (sfa_rows + 3) / 4 * 16, 0x00020000);
            const int sfg_a_voff = (lane >> 4) * sfa_kb_stride + (lane & 15) * 16;
            // two SFB values per tile: the 128 x 128 scale blocks its 64 gate rows and its 64 up rows were quantised in
            float* sfb_tile = const_cast<float*>(p.sfb) + group * p.sfb_sg + static_cast<int64_t>(t.n0 / 256) * 2 * p.sfb_sn;
            const int sfb_sn_bytes = static_cast<int>(p.sfb_sn) * 4;
            const auto sfb_rsrc = __builtin_amdgcn_make_buffer_rsrc(sfb_tile, 0, (num_kb - 1) * sfb_kb_stride + sfb_sn_bytes + 4, 0x00020000);
            // SFB piece: lane l -> K block (l & 3) of the group, gate (l & 4 == 0) or up scale row: [gate x 4][up x 4] floats in the slot
            const int sfg_b_voff = (lane & 3) * sfb_kb_stride + ((lane >> 2) & 1) * sfb_sn_bytes;

            auto issue_block = [&](int slot_off, int j) {
                const unsigned oob = j < num_kb ? 0u : OOB;
                uint8_t* stage = lds + slot_off;
                if ((j & 3) == 0) {                             // the scales of K blocks j .. j + 3
                    uint8_t* slot = lds + SFG_OFF + ((j >> 2) & (SFG_SLOTS - 1)) * SFG_SLOT;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        sfa_rsrc, (__attribute__((address_space(3))) void*)slot, 16,
                        static_cast<int>(static_cast<unsigned>(sfg_a_voff) | oob), j * sfa_kb_stride, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        sfb_rsrc, (__attribute__((address_space(3))) void*)(slot + 1024), 4,
                        static_cast<int>(static_cast<unsigned>(sfg_b_voff) | oob), j * sfb_kb_stride, 0, 0);
                }
                #pragma unroll
                for (int q = 0; q < A_ITERS; ++q) {
                    const int unit = wave + NW * q;
                    const int voff = static_cast<int>(static_cast<unsigned>(a_voff) + (static_cast<unsigned>(a_unit_row(unit) * lda) | oob));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        a_rsrc, (__attribute__((address_space(3))) void*)(stage + unit * 1024), 16, voff, j * 128, 0, 0);
                }
                #pragma unroll
                for (int q = 0; q < B_ITERS; ++q) {
                    const int unit = wave + NW * q;
                    const int voff = static_cast<int>(static_cast<unsigned>(b_voff) +
                                                      (static_cast<unsigned>(b_row_perm<WN>(q * (NW * 8)) * ldb) | oob));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        b_rsrc, (__attribute__((address_space(3))) void*)(stage + A_BYTES + unit * 1024), 16, voff, j * 128, 0, DG_SWIGLU_B_AUX);   // nt: weights stream once
                }
            };
            #pragma unroll
            for (int j = 0; j < STAGES - 1; ++j)
                issue_block(j * STAGE_BYTES, j);

            int cur = 0, fill = (STAGES - 1) * STAGE_BYTES;
            for (int kb = 0; kb < num_kb; ++kb) {
                asm volatile("s_waitcnt vmcnt(%c0)" :: "i"((STAGES - 2) * PIECES) : "memory");
                raw_barrier();
                issue_block(fill, kb + STAGES - 1);
                const uint8_t* stage = lds + cur;
                const uint8_t* sfg = lds + SFG_OFF + ((kb >> 2) & (SFG_SLOTS - 1)) * SFG_SLOT;      // [4 blocks][64 rows], then [gate x 4][up x 4]
                const v4f q = *reinterpret_cast<const v4f*>(sfg + (kb & 3) * 256 + ((lane & 15) * MS) * 4);
                const float sa[MS] = {q[0], q[1], q[2], q[3]};
                const float sb = *reinterpret_cast<const float*>(sfg + 1024 + (wn >> 1) * 16 + (kb & 3) * 4);
                const uint8_t* b_tile = stage + A_BYTES + (wn * WN) * 128;
                v8i bf[NS];
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    bf[ns] = load_fragment(b_tile + ns * 2048, frag_off);
                #pragma unroll
                for (int ms = 0; ms < MS; ++ms) {
                    const v8i af = load_fragment(stage + ms * 2048, frag_off);
                    const float scale = sa[ms] * sb;
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns) {
                        const v4f part = mfma_fp8_k128(bf[ns], af);
                        #pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[ms][ns][e] = __builtin_fmaf(scale, part[e], acc[ms][ns][e]);      // the promotion of every other kernel: fused multiply-add
                    }
                }
                fill = cur;
                cur = (cur == (STAGES - 1) * STAGE_BYTES) ? 0 : cur + STAGE_BYTES;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                        // the ring is free: the epilogue exchange lives in its first 17 KiB

            // ---- 1. BF16 rounding; 2. every wave hands the half of its values that its partner wave finishes to the LDS ----
            float* xch = reinterpret_cast<float*>(lds);                 // [4 waves][MS / 2 * NS * 4 registers][64 lanes]
            float* row_max = reinterpret_cast<float*>(lds + 2 * (MS * NS * 4) * 64 * 4);    // [2 column halves][64 rows], then [64] block amax
            #pragma unroll
            for (int ms = 0; ms < MS; ++ms)
                #pragma unroll
                for (int ns = 0; ns < NS; ++ns)
                    #pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[ms][ns][e] = round_bf16(acc[ms][ns][e]);
            // round 5: the SiLU arithmetic is shared by all four waves (it used to run on the two gate waves alone -- two of a CU's four SIMDs,
            // with both co-resident workgroups' gate waves on the same two: 1.7 us of the call, profiles/r05_probe/swiglu_epilogue_ablation.log).
            // A gate wave keeps M-subtiles 0, 1 and hands its g of subtiles 2, 3 to the up wave of the same columns; the up wave hands over its u
            // of subtiles 0, 1 and keeps 2, 3: every (row, column) is still computed once, by the same expression.
            constexpr int HALF = MS / 2;
            const bool gate_wave = wn < 2;
            const int my_ms0 = gate_wave ? 0 : HALF;                 // the M-subtiles this wave finishes: my_ms0, my_ms0 + 1
            {
                float* mine_out = xch + wn * (HALF * NS * 4) * 64;   // [4 waves][HALF * NS * 4 registers][64 lanes]
                #pragma unroll
                for (int h = 0; h < HALF; ++h)                       // (the subtiles the partner wave finishes: gate -> HALF + h, up -> h)
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        #pragma unroll
                        for (int e = 0; e < 4; ++e)
                            mine_out[((h * NS + ns) * 4 + e) * 64 + lane] = gate_wave ? acc[HALF + h][ns][e] : acc[h][ns][e];
            }
            __syncthreads();
            float amax[HALF] = {0.f, 0.f};
            float y_out[HALF][NS][4];
            {
                const float* partner = xch + (wn ^ 2) * (HALF * NS * 4) * 64;
                float rw[HALF] = {1.f, 1.f};
                if (o.row_weight != nullptr) {
                    // (rows past m_end: whatever the buffer holds -- they are never stored and never meet another row's amax)
                    const v4f w4 = *reinterpret_cast<const v4f*>(o.row_weight + group * o.rw_sg + t.m0 + (lane & 15) * MS);
                    rw[0] = gate_wave ? w4[0] : w4[2]; rw[1] = gate_wave ? w4[1] : w4[3];
                }
                #pragma unroll
                for (int h = 0; h < HALF; ++h)
                    #pragma unroll
                    for (int ns = 0; ns < NS; ++ns)
                        #pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float other = partner[((h * NS + ns) * 4 + e) * 64 + lane];
                            float g = gate_wave ? acc[h][ns][e] : other;
                            float u = gate_wave ? other : acc[HALF + h][ns][e];
                            if (o.clamp > 0.f) {
                                g = fminf(g, o.clamp);
                                u = fminf(fmaxf(u, -o.clamp), o.clamp);
                            }
#if defined(DG_SWIGLU_ABL) && (DG_SWIGLU_ABL & 1)         // timing ablation (tuning builds only): no exponential, no division -- wrong values
                            float y = g * u;
#else
                            float y = (g / (1.0f + expf(-g))) * u;                       // silu(g.float()) * u.float()
#endif
                            // without a routing weight the operator stands for "... -> BF16 intermediate -> per_token_cast_to_fp8": round to
                            // BF16 as the unfused pipeline stores it; with one it is the reference kernel's epilogue, which keeps
                            // silu(gate) * up * weight in FP32 up to the amax and the FP8 cast (sm100_fp8_fp4_mega_moe.cuh:1001-1020)
                            y = o.row_weight != nullptr ? y * rw[h] : round_bf16(y);
                            y_out[h][ns][e] = y;
                            amax[h] = fmaxf(amax[h], fabsf(y));
                        }
                // 3. the row's 4 lanes (lane bits 4, 5), then the two waves that hold the row's two column halves
                #pragma unroll
                for (int h = 0; h < HALF; ++h) {
                    amax[h] = fmaxf(amax[h], __shfl_xor(amax[h], 16, 64));
                    amax[h] = fmaxf(amax[h], __shfl_xor(amax[h], 32, 64));
                    if (lg == 0)
                        row_max[(wn & 1) * 64 + (lane & 15) * MS + my_ms0 + h] = amax[h];     // tile row of (lane, ms): interleaved rows
                }
            }
            __syncthreads();
            if (wave == 0) {
                // 3b. the other half of the 128-wide block lives in tile_id ^ 1: publish this tile's row amax, take the partner's, clear its slot
                const float mine = fmaxf(row_max[lane], row_max[64 + lane]);
                if (!(o.fault == 1 && (tile_id & 1)))
                    __hip_atomic_store(o.amax_ws + static_cast<int64_t>(tile_id) * 64 + lane, 0x80000000u | __float_as_uint(mine),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t* theirs = o.amax_ws + static_cast<int64_t>(tile_id ^ 1) * 64 + lane;
                // The partner is resident or already done (adjacent in dispatch order, see the file header), so this wait is short -- but it
                // is BOUNDED all the same: a workspace that is not all-zero (an aborted launch, two launches sharing one workspace) or a lost
                // partner must end in an error the host can see and NaN scales, not in a hung device (reference: comm/barrier.cuh:36-40).
                uint32_t v;
                const long long t_wait = __builtin_amdgcn_s_memrealtime();
                bool timed_out = false;
#if defined(DG_SWIGLU_ABL) && (DG_SWIGLU_ABL & 2)         // timing ablation (tuning builds only): nobody waits for a partner -- wrong scales
                v = 0x80000000u | __float_as_uint(mine);
                if (false)
#endif
                do {
                    v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!(v & 0x80000000u)) {
                        __builtin_amdgcn_s_sleep(1);
                        timed_out = __builtin_amdgcn_s_memrealtime() - t_wait > o.timeout_ticks;
                    }
                } while (!(v & 0x80000000u) && !timed_out);
                if (v & 0x80000000u) {
                    __hip_atomic_store(theirs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    row_max[128 + lane] = fmaxf(mine, __uint_as_float(v & 0x7fffffffu));
                } else {
                    row_max[128 + lane] = __uint_as_float(0x7fc00000u);         // NaN: poisons the row's scale below
                    // take this tile's own slot back: nobody may find a valid bit from an aborted exchange in the workspace of the next launch
                    // (a partner that was merely slow then times out as well -- both rows are poisoned and counted)
                    __hip_atomic_store(o.amax_ws + static_cast<int64_t>(tile_id) * 64 + lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == 0)
                        atomicAdd(o.errors, 1u);
                }
            }
            __syncthreads();
            {
                const int kb2 = t.n0 / (2 * BN);                         // the intermediate's 128-block = GEMM2's K block
                #pragma unroll
                for (int h = 0; h < HALF; ++h) {
                    const int row_in_tile = (lane & 15) * MS + my_ms0 + h;
                    const int row = t.m0 + row_in_tile;
                    const float row_amax = row_max[128 + row_in_tile];
                    float scale = fmaxf(row_amax, 1e-4f) * (1.0f / 448.0f);     // the reference kernel (math.cuh:93) and torch's `/ 448.0` on a device both multiply
                    if (row_amax != row_amax)
                        scale = row_amax;                                       // the exchange timed out: NaN scale, NaN bytes (fmaxf drops a NaN)
                    if (o.use_ue8m0) {
                        const uint32_t bits = __float_as_uint(scale);
                        uint32_t e = ((bits >> 23) & 0xffu) + ((bits & 0x7fffffu) != 0 ? 1u : 0u);
                        e = e < 1u ? 1u : (e > 254u ? 254u : e);
                        scale = __uint_as_float(e << 23);
                    }
                    const float inv = 1.0f / scale;
                    if (row >= t.m_end)
                        continue;
                    uint8_t* qrow = o.q + group * o.q_sg + static_cast<int64_t>(row) * o.q_sm + (t.n0 / BN) * 64 + (wn & 1) * 32 + lg * 8;
                    int w0 = 0, w1 = 0;                                  // N-subtiles 0 and 1: 8 consecutive intermediate columns
                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y_out[h][0][0] * inv, y_out[h][0][1] * inv, w0, false);
                    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(y_out[h][0][2] * inv, y_out[h][0][3] * inv, w0, true);
                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y_out[h][1][0] * inv, y_out[h][1][1] * inv, w1, false);
                    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(y_out[h][1][2] * inv, y_out[h][1][3] * inv, w1, true);
                    *reinterpret_cast<uint2*>(qrow) = make_uint2(static_cast<uint32_t>(w0), static_cast<uint32_t>(w1));
                    if ((wn & 1) == 0 && lg == 0 && (tile_id & 1) == 0)
                        o.sf[group * o.sf_sg + static_cast<int64_t>(kb2) * o.sf_sk + row] = scale;
                }
            }
            __syncthreads();                        // the next tile's prologue refills the LDS
        }
    }
}

template <int STAGES>
__global__ __launch_bounds__(256)
void dg_fp8_gemm_stream_swiglu_kernel(const GemmParams p, const SwigluOut o) {
    stream_swiglu_kernel_body<STAGES>(p, o);
}

// ---------------------------------------------------------------------------------------------------------------
// World-size-1 routing around the fused expert MLP (the reference-shaped entry fp8_mega_moe, deepgemm_amd/mega.py; reference:
// the dispatch / combine stages of sm100_fp8_fp4_mega_moe.cuh:357-405, 523-595 -- there they pull / push token rows over NVLink; with
// one rank they degenerate to a scatter into the masked layout and a gather-sum back).
// dg_moe_scatter_kernel: one workgroup per token.  For every top-k entry with a valid expert the token claims the next row slot of
// that expert (atomic counter = masked_m), copies its FP8 row and its K / 128 scales (into the MN-major SF layout the GEMMs read
// zero-copy), stores the routing weight of the slot and remembers the slot for the combine.  Slot ORDER inside an expert depends on the
// arrival order of the claims; no RESULT does: every row of the grouped GEMMs and of the per-token re-quantisation is computed
// independently of its neighbours, and the combine sums a token's rows in top-k order.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMoeMaxTopk = 32;      // top-k entries per token of the scatter kernel (their slot claims go out together, one per thread)
struct MoeRoute {
    const uint8_t* x; const float* x_sf; const void* topk_idx; const float* topk_w;
    int tokens, hidden, topk, num_experts, max_m, idx64;
    int64_t x_sm, xsf_sm;
    uint8_t* a; float* sfa; float* rw; int32_t* slot; int32_t* counts; uint32_t* errors;
    int64_t a_sg, a_sm, sfa_sg, sfa_sk, rw_sg;
};

#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_moe_scatter_kernel(const MoeRoute r) {
    // (round 6: the token's top-k entries claim their slots TOGETHER -- one atomic round trip instead of top-k in a row -- and the row is read once)
    __shared__ int s_pos[kMoeMaxTopk], s_e[kMoeMaxTopk];
    const int t = blockIdx.x;
    if (threadIdx.x < r.topk) {
        const int j = threadIdx.x;
        const int64_t e64 = r.idx64 ? static_cast<const int64_t*>(r.topk_idx)[static_cast<int64_t>(t) * r.topk + j]
                                    : static_cast<int64_t>(static_cast<const int32_t*>(r.topk_idx)[static_cast<int64_t>(t) * r.topk + j]);
        const bool valid = e64 >= 0 && e64 < r.num_experts;          // (-1 = no expert for this entry, as the reference's masked top-k)
        const int e = valid ? static_cast<int>(e64) : 0;
        int pos = -1;
        if (valid) {
            pos = atomicAdd(r.counts + e, 1);
            if (pos >= r.max_m) {                                    // more rows than the buffer was sized for: dropped, counted, visible
                atomicAdd(r.errors, 1u);
                atomicSub(r.counts + e, 1);
                pos = -1;
            }
        }
        s_pos[j] = pos; s_e[j] = e;
        r.slot[static_cast<int64_t>(t) * r.topk + j] = pos < 0 ? -1 : e * r.max_m + pos;
        if (pos >= 0)
            r.rw[e * r.rw_sg + pos] = r.topk_w[static_cast<int64_t>(t) * r.topk + j];
    }
    __syncthreads();
    const uint4* src = reinterpret_cast<const uint4*>(r.x + static_cast<int64_t>(t) * r.x_sm);
    for (int c = threadIdx.x; c < r.hidden / 16; c += 256) {
        const uint4 v = src[c];
        for (int j = 0; j < r.topk; ++j)
            if (s_pos[j] >= 0)
                reinterpret_cast<uint4*>(r.a + s_e[j] * r.a_sg + static_cast<int64_t>(s_pos[j]) * r.a_sm)[c] = v;
    }
    for (int kb = threadIdx.x; kb < r.hidden / 128; kb += 256) {
        const float sf = r.x_sf[static_cast<int64_t>(t) * r.xsf_sm + kb];
        for (int j = 0; j < r.topk; ++j)
            if (s_pos[j] >= 0)
                r.sfa[s_e[j] * r.sfa_sg + kb * r.sfa_sk + s_pos[j]] = sf;
    }
}
#endif

// y[t, :] = bf16( sum_j float(y2[slot(t, j), :]) ), j in top-k order, FP32 accumulation, entries without a slot skipped.
#ifndef DG_SHARD_TU   // (a plain kernel: defined once, in the dg_api.hip translation unit -- see kernel_instances.inc)
__global__ __launch_bounds__(256)
void dg_moe_combine_kernel(const uint16_t* y2, const int32_t* slot, int tokens, int topk, int hidden, int64_t y2_row_stride, uint16_t* y,
                           int64_t y_sm) {
    // grid (tokens, ceil(hidden / 2048)), round 6: a thread owns 8 columns of one token and has the rows of all its entries in flight together
    // (one token per workgroup with one row at a time: top-k x hidden / 2048 dependent (slot, row) load pairs per thread, 10.4 us for 64 tokens)
    const int t = blockIdx.x;
    const int c = (blockIdx.y * 256 + threadIdx.x) * 8;
    if (c >= hidden)
        return;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < topk; j0 += 8) {
        int s[8];
        uint4 v[8];
        #pragma unroll
        for (int u = 0; u < 8; ++u)
            s[u] = j0 + u < topk ? slot[static_cast<int64_t>(t) * topk + j0 + u] : -1;
        #pragma unroll
        for (int u = 0; u < 8; ++u)
            if (s[u] >= 0)
                v[u] = *reinterpret_cast<const uint4*>(y2 + static_cast<int64_t>(s[u]) * y2_row_stride + c);
        #pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (s[u] < 0)
                continue;
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
                sum[2 * i] += bf16_lo(w[i]);
                sum[2 * i + 1] += bf16_hi(w[i]);
            }
        }
    }
    uint4 out;
    out.x = pack_bf16(sum[0], sum[1]); out.y = pack_bf16(sum[2], sum[3]);
    out.z = pack_bf16(sum[4], sum[5]); out.w = pack_bf16(sum[6], sum[7]);
    *reinterpret_cast<uint4*>(y + static_cast<int64_t>(t) * y_sm + c) = out;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// In-kernel dispatch / combine over peer-mapped memory (round 6): the communication half of the reference's Mega-MoE kernel -- token rows
// pulled / pushed between GPUs from inside the kernel with system-scope acquire / release (impls/sm100_fp8_fp4_mega_moe.cuh:357-405 dispatch,
// :523-595 remote pulls and write-back; comm/barrier.cuh:47-83 the NVLink barrier) -- rebuilt for peer-mapped HBM over xGMI.
//
// Every rank owns one SYMMETRIC REGION (same layout on every rank, P2pLayout) that all its peers map (hipIpcOpenMemHandle; on one node every
// GPU pair is one xGMI hop).  Per call, rank s:
//   1. dg_moe_p2p_dispatch_kernel: for every (token t, top-k entry j) with expert e owned by rank o = e / E_loc: claims the next row slot of
//      that expert with ONE system-scope fetch-add on o's counter, then writes the FP8 row, its K / 128 scales (straight into the MN-major
//      layout the grouped GEMM reads), its routing weight and its return address (rank, t, j) into o's masked-layout input -- PUSHED by the
//      sender, where the reference's receivers pull (a push needs no request / response round trip over a point-to-point link).  The last
//      workgroup to finish releases `arrived[s] = epoch` on every peer, then waits (bounded) for the epoch of every peer on its own flags
//      and publishes masked_m[e] = min(counter, capacity) for the GEMMs that follow on the stream.
//   2. fused L1 + L2 on the local experts (the kernels of the one-rank path, unchanged).
//   3. dg_moe_p2p_combine_kernel: every valid output row goes back to its return address (a BF16 row write into the token owner's y_rows);
//      the last workgroup zeroes the slot counters for the next call and releases `combined[o] = epoch` on every peer.
//   4. dg_moe_p2p_reduce_kernel: waits (bounded) for `combined[*] == epoch`, then y[t] = bf16(sum_j float(y_rows[t, j])) in top-k order.
// Five launches per step, no host round trip, no collective.  The epoch protocol needs no second buffer: a peer can only start dispatch
// e + 1 after its reduce e has seen combined[o] == e from every owner o, and o releases that flag after its combine kernel has read the last
// row of call e and zeroed its counters.  Slot order inside an expert depends on arrival order; no result does (every row of the grouped
// GEMMs and of the per-token re-quantisation is computed independently, the reduce sums in top-k order).
// Visibility: payload stores are written through (sc0 sc1) and fenced at system scope before the flag's release; the consumers of dispatched
// rows are LATER KERNELS on the owner's stream (kernel-boundary acquire); the reduce kernel reads the returned rows in the kernel that waited,
// with cache-bypassing loads behind a system-scope acquire.  The region is allocated fine-grained where the runtime allows it
// (dg_symm_alloc).  Every wait is bounded (timeout_ticks of the 100 MHz wall clock) and counted in `errors`: a lost peer ends in a flagged,
// wrong result -- never in a hung device (reference: comm/barrier.cuh:12,36-40).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMaxPeers = 16;
constexpr int kP2pMaxTopk = 32;      // top-k entries per token of the in-kernel dispatch (their slot claims go out together, one per thread)

struct P2pLayout {      // byte offsets inside a symmetric region (dg_moe_p2p_layout, the same on every rank)
    int64_t counts;     // uint32 [E_loc]: rows claimed per local expert (system-scope fetch-add by every sender; zeroed by the owner's combine)
    int64_t arrived;    // uint32 [R]: arrived[s] = epoch once rank s has pushed all its rows of that call
    int64_t combined;   // uint32 [R]: combined[o] = epoch once owner o has returned all rows of that call
    int64_t done;       // uint32 [2]: workgroup arrival counters of this rank's own dispatch / combine kernels
    int64_t l1_acts;    // e4m3 [E_loc][cap][H]
    int64_t l1_sf;      // float [E_loc][H / 128][cap]   (MN-major: what the grouped GEMM reads zero-copy)
    int64_t row_w;      // float [E_loc][cap]
    int64_t src_info;   // int32 [E_loc][cap]: (source rank << 24) | (token * topk + j)
    int64_t y_rows;     // bf16 [T][topk][H]: combine landing zone of the token owner
    int64_t bytes;
};

struct P2pArgs {
    uint8_t* peer[kMaxPeers];       // region base of every rank as mapped into THIS process (peer[rank] = the own region)
    P2pLayout lay;
    int world, rank;
    int tokens, hidden, topk, num_experts, local_experts, cap;
    unsigned epoch;
    long long timeout_ticks;
    uint32_t* errors;               // local uint32 [4]: 0 rows dropped over a capacity (counted by their SENDER), 1 partner waits of the fused L1 kernel,
                                    // 2 dispatch flag waits that timed out, 3 combine flag waits that timed out
    // dispatch
    const uint8_t* x; const float* x_sf; const void* topk_idx; const float* topk_w; int idx64;
    int64_t x_sm, xsf_sm;
    int32_t* masked_m;              // local int32 [E_loc]
    uint8_t* pair_ok;               // local uint8 [T * topk]: 1 = the pair was delivered (a row will come back)
    // combine
    const uint16_t* l2_out; int64_t l2_sg, l2_sm;
    // reduce
    uint16_t* y; int64_t y_sm;
    const uint32_t* swiglu_errors;  // optional: word 0 of the fused L1 kernel's workspace, copied into errors[1]
    long long* stamps;              // tuning aid (normally null; dg_set_debug_buffer): 100 MHz wall-clock stamps of the phases, [kernel 0..2][8]
};

#ifndef DG_SHARD_TU   // (plain kernels: defined once, in the dg_api.hip translation unit)
// 16 bytes at row + off, written through to memory (sc0 sc1): the line must not linger in this XCD's L2 when the flag goes out.  `row` is
// WAVE-UNIFORM (a row of the peer's region), the lane's part travels as the offset: a descriptor built from a per-lane pointer makes hipcc wrap
// the store in a 64-trip waterfall loop -- the first version of these kernels issued one lane per instruction (22 of the dispatch kernel's 28 us).
__device__ __forceinline__ void p2p_store16(uint8_t* row, int off, const uint4& v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), __builtin_amdgcn_make_buffer_rsrc(uniform_pointer(row), 0, 0x7ffffff0, 0x00020000), off, 0, 17);
}
__device__ __forceinline__ uint4 p2p_load16(const uint8_t* row, int off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(__builtin_amdgcn_make_buffer_rsrc(uniform_pointer(const_cast<uint8_t*>(row)), 0, 0x7ffffff0, 0x00020000), off, 0, 17));
}
// Bounded wait for `*flag == want`; false = timed out.  RELAXED polls (an acquire load would put a cache invalidate behind every poll): what the
// waiter reads afterwards is either read past the caches (p2p_load16, system-scope atomic loads) or behind the ONE acquire fence of its caller.
__device__ __forceinline__ bool p2p_wait_flag(const uint32_t* flag, unsigned want, long long timeout_ticks) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == want)
        return true;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
        if (wall_clock64() - t0 > timeout_ticks)
            return false;
        __builtin_amdgcn_s_sleep(2);
    }
    return true;
}
// "Every store this wave issued has reached memory": all payload stores of these kernels are written through (sc0 sc1) or system-scope atomic
// stores, so once they are acknowledged nothing of theirs sits dirty in a cache -- the write-back of the whole L2 that a system-scope RELEASE
// fence carries (buffer_wbl2: it also flushes every unrelated dirty line, e.g. the 3.7 MB the grouped GEMM just wrote) has nothing to add.
// Round 6 measurement (one rank, 64 tokens, top-4, hidden 7168; tools/mega_step_latency.py): with release fences and one-at-a-time claims /
// loads the three kernels took 34 + 38 + 51 us; in this form see profiles/r06_probe/mega_p2p_step_latency.log.
__device__ __forceinline__ void p2p_stamp(const P2pArgs& a, int kernel, int slot) {
    if (a.stamps != nullptr && threadIdx.x == 0)
        a.stamps[kernel * 8 + slot] = wall_clock64();
}
__device__ __forceinline__ void p2p_stores_done() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}


__global__ __launch_bounds__(256)
void dg_moe_p2p_dispatch_kernel(const P2pArgs a) {
    __shared__ int s_pos[kP2pMaxTopk], s_owner[kP2pMaxTopk], s_le[kP2pMaxTopk], s_last;
    const int t = blockIdx.x;
    uint8_t* self = a.peer[a.rank];
    if (blockIdx.x == 0) p2p_stamp(a, 0, 0);
    if (t < a.tokens) {
        // every entry of the token claims its row slot AT ONCE (one round trip to the owners' counters instead of top-k of them in a row)
        if (threadIdx.x < a.topk) {
            const int j = threadIdx.x;
            const int64_t e64 = a.idx64 ? static_cast<const int64_t*>(a.topk_idx)[static_cast<int64_t>(t) * a.topk + j]
                                        : static_cast<int64_t>(static_cast<const int32_t*>(a.topk_idx)[static_cast<int64_t>(t) * a.topk + j]);
            const bool valid = e64 >= 0 && e64 < a.num_experts;          // (-1 = no expert for this entry, as the reference's masked top-k)
            const int e = valid ? static_cast<int>(e64) : 0;
            const int owner = e / a.local_experts, le = e - owner * a.local_experts;
            int pos = -1;
            if (valid) {
                pos = static_cast<int>(__hip_atomic_fetch_add(reinterpret_cast<uint32_t*>(a.peer[owner] + a.lay.counts) + le, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
                if (pos >= a.cap) {                                      // more rows than the owner's buffer holds: dropped, counted HERE (by the sender)
                    atomicAdd(a.errors, 1u);
                    pos = -1;
                }
            }
            s_pos[j] = pos; s_owner[j] = owner; s_le[j] = le;
            a.pair_ok[static_cast<int64_t>(t) * a.topk + j] = pos >= 0 ? 1 : 0;
            if (pos >= 0) {
                uint8_t* dst = a.peer[owner];
                const int64_t row = static_cast<int64_t>(le) * a.cap + pos;
                __hip_atomic_store(reinterpret_cast<uint32_t*>(dst + a.lay.row_w) + row, __float_as_uint(a.topk_w[static_cast<int64_t>(t) * a.topk + j]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(reinterpret_cast<int32_t*>(dst + a.lay.src_info) + row, (a.rank << 24) | (t * a.topk + j), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        if (blockIdx.x == 0) p2p_stamp(a, 0, 1);                     // the claims are back
        // the token's row is read ONCE and pushed to every owner that granted a slot
        const uint4* src = reinterpret_cast<const uint4*>(a.x + static_cast<int64_t>(t) * a.x_sm);
        for (int c = threadIdx.x; c < a.hidden / 16; c += 256) {
            const uint4 v = src[c];
            for (int j = 0; j < a.topk; ++j)
                if (s_pos[j] >= 0)
                    p2p_store16(a.peer[s_owner[j]] + a.lay.l1_acts + (static_cast<int64_t>(s_le[j]) * a.cap + s_pos[j]) * a.hidden, c * 16, v);
        }
        if (blockIdx.x == 0) p2p_stamp(a, 0, 7);                     // (row pieces issued; the scales follow)
        for (int kb = threadIdx.x; kb < a.hidden / 128; kb += 256) {
            const unsigned sf = __float_as_uint(a.x_sf[static_cast<int64_t>(t) * a.xsf_sm + kb]);
            for (int j = 0; j < a.topk; ++j)
                if (s_pos[j] >= 0) {
                    float* dsf = reinterpret_cast<float*>(a.peer[s_owner[j]] + a.lay.l1_sf) + static_cast<int64_t>(s_le[j]) * (a.hidden / 128) * a.cap + s_pos[j];
                    __hip_atomic_store(reinterpret_cast<uint32_t*>(dsf + static_cast<int64_t>(kb) * a.cap), sf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
        }
    }
    // every store of this workgroup has reached memory before it counts itself done
    if (blockIdx.x == 0) p2p_stamp(a, 0, 2);                         // stores issued
    p2p_stores_done();
    __syncthreads();
    if (blockIdx.x == 0) p2p_stamp(a, 0, 3);                         // ... and acknowledged
    uint32_t* done = reinterpret_cast<uint32_t*>(self + a.lay.done);
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!s_last)
        return;
    // ---- the last workgroup of this rank's dispatch: announce, wait for everybody's rows, publish the counts ----
    p2p_stamp(a, 0, 4);                                              // the last workgroup knows it is the last
    if (threadIdx.x == 0)
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < a.world) {
        __hip_atomic_store(reinterpret_cast<uint32_t*>(a.peer[threadIdx.x] + a.lay.arrived) + a.rank, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (!p2p_wait_flag(reinterpret_cast<const uint32_t*>(self + a.lay.arrived) + threadIdx.x, a.epoch, a.timeout_ticks))
            atomicAdd(a.errors + 2, 1u);
    }
    __syncthreads();
    p2p_stamp(a, 0, 5);                                              // every peer has arrived
    // ONE system-scope acquire per call: lines of the region that an earlier step left in this GPU's caches (a coarse-grained region: the
    // previous step's GEMM reads) must not answer for what the peers have just written; the kernels that read the rows start behind it
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    for (int le = threadIdx.x; le < a.local_experts; le += 256) {
        const unsigned c = __hip_atomic_load(reinterpret_cast<const uint32_t*>(self + a.lay.counts) + le, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a.masked_m[le] = static_cast<int>(c < static_cast<unsigned>(a.cap) ? c : static_cast<unsigned>(a.cap));
    }
    p2p_stamp(a, 0, 6);
}

__global__ __launch_bounds__(256)
void dg_moe_p2p_combine_kernel(const P2pArgs a) {
    __shared__ int s_last;
    uint8_t* self = a.peer[a.rank];
    const int total = a.local_experts * a.cap;
    if (blockIdx.x == 0) p2p_stamp(a, 1, 0);
    for (int r = blockIdx.x; r < total; r += gridDim.x) {
        const int le = r / a.cap, slot = r - le * a.cap;
        if (slot >= a.masked_m[le])
            continue;
        const int info = __hip_atomic_load(reinterpret_cast<const int32_t*>(self + a.lay.src_info) + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const int src_rank = (info >> 24) & 0xff, pair = info & 0xffffff;
        const uint4* src = reinterpret_cast<const uint4*>(a.l2_out + le * a.l2_sg + static_cast<int64_t>(slot) * a.l2_sm);
        uint8_t* drow = a.peer[src_rank] + a.lay.y_rows + static_cast<int64_t>(pair) * a.hidden * 2;
        for (int c = threadIdx.x; c < a.hidden / 8; c += 256)
            p2p_store16(drow, c * 16, src[c]);
    }
    if (blockIdx.x == 0) p2p_stamp(a, 1, 1);
    p2p_stores_done();
    __syncthreads();
    if (blockIdx.x == 0) p2p_stamp(a, 1, 2);
    uint32_t* done = reinterpret_cast<uint32_t*>(self + a.lay.done) + 1;
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!s_last)
        return;
    // ---- the last workgroup: the slot counters are free for the next call, then everybody may know that this owner is done ----
    p2p_stamp(a, 1, 3);
    if (threadIdx.x == 0)
        __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int le = threadIdx.x; le < a.local_experts; le += 256)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(self + a.lay.counts) + le, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    p2p_stores_done();
    __syncthreads();
    if (threadIdx.x < a.world)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(a.peer[threadIdx.x] + a.lay.combined) + a.rank, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    p2p_stamp(a, 1, 4);
}

// grid (tokens, ceil(hidden / 2048)): a thread owns 8 columns of one token and has the rows of ALL its top-k entries in flight together (the
// returned rows are read past the caches: one at a time they were top-k x hidden / 2048 dependent ~2.5 us round trips per thread)
__global__ __launch_bounds__(256)
void dg_moe_p2p_reduce_kernel(const P2pArgs a) {
    uint8_t* self = a.peer[a.rank];
    if (blockIdx.x == 0 && blockIdx.y == 0) p2p_stamp(a, 2, 0);
    if (threadIdx.x < a.world)
        if (!p2p_wait_flag(reinterpret_cast<const uint32_t*>(self + a.lay.combined) + threadIdx.x, a.epoch, a.timeout_ticks) && blockIdx.x == 0 && blockIdx.y == 0)
            atomicAdd(a.errors + 3, 1u);
    __syncthreads();
    // (no acquire fence: every returned row is read with system-scope loads -- sc0 sc1, coherent by themselves like the polls of the flag; a
    //  cache invalidate in each of the (token, column block) workgroups cost the kernel 6.5 us of its 9; pair_ok is an earlier kernel's output)
    if (blockIdx.x == 0 && blockIdx.y == 0) p2p_stamp(a, 2, 1);
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && a.swiglu_errors != nullptr)
        a.errors[1] = *a.swiglu_errors;
    const int t = blockIdx.x;
    const int c = (blockIdx.y * 256 + threadIdx.x) * 8;
    if (t >= a.tokens || c >= a.hidden)
        return;
    const uint8_t* rows = self + a.lay.y_rows + static_cast<int64_t>(t) * a.topk * a.hidden * 2;
    const uint8_t* ok = a.pair_ok + static_cast<int64_t>(t) * a.topk;
    float sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};              // as dg_moe_combine_kernel: top-k order, FP32, absent entries skipped
    for (int j0 = 0; j0 < a.topk; j0 += 8) {
        uint4 v[8];
        bool have[8];
        #pragma unroll
        for (int u = 0; u < 8; ++u) {
            have[u] = j0 + u < a.topk && ok[j0 + u] != 0;
            if (have[u])
                v[u] = p2p_load16(rows + static_cast<int64_t>(j0 + u) * a.hidden * 2, c * 2);
        }
        #pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (!have[u])
                continue;
            const uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            #pragma unroll
            for (int i = 0; i < 4; ++i) {
                sum[2 * i] += bf16_lo(w[i]);
                sum[2 * i + 1] += bf16_hi(w[i]);
            }
        }
    }
    uint4 out;
    out.x = pack_bf16(sum[0], sum[1]); out.y = pack_bf16(sum[2], sum[3]);
    out.z = pack_bf16(sum[4], sum[5]); out.w = pack_bf16(sum[6], sum[7]);
    *reinterpret_cast<uint4*>(a.y + static_cast<int64_t>(t) * a.y_sm + c) = out;
    if (blockIdx.x == 0 && blockIdx.y == 0) p2p_stamp(a, 2, 2);
}
#endif  // DG_SHARD_TU

}  // namespace dg
