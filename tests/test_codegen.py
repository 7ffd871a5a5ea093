"""Static guard on what hipcc made of the production kernels (no GPU needed: the library cross-compiles here).

The fast kernels sit at the edge of the register file; a spill that lands inside the K loop costs far more than its
instruction (scratch traffic counts towards vmcnt and tightens every counted wait) and does not show up in any
correctness test -- only in sustained throughput.  It happened once (HISTORY.md, "Measure sustained, and watch the register
allocator"): this test keeps it from happening silently again."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'

PRODUCTION = [       # <BM, BN, WAVES_M, WAVES_N, PERSIST, B_MN, SPLITK, A_MN, K_TAIL, MERGED, STREAM_A, PC, SFA_RM> for the duo kernels
    'dg_fp8_gemm_duo_kernel<256,256,2,4,0,0,0,0,0,0,0,0,0>', 'dg_fp8_gemm_duo_kernel<256,256,2,4,1,0,0,0,0,0,0,0,0>', 'dg_fp8_gemm_duo_kernel<128,256,2,4,0,0,0,0,0,1,0,0,0>',
    'dg_fp8_gemm_duo_kernel<256,256,2,4,1,1,0,0,0,0,0,0,0>', 'dg_fp8_gemm_duo_kernel<128,256,2,4,0,1,0,0,0,1,0,0,0>',
    'dg_fp8_gemm_duo_kernel<128,256,2,4,1,0,1,0,0,1,0,0,0>', 'dg_fp8_gemm_duo_kernel<128,256,2,4,1,1,1,0,0,1,0,0,0>', 'dg_fp8_gemm_duo_kernel<256,256,2,4,1,0,0,1,0,0,0,0,0>', 'dg_fp8_gemm_duo_kernel<256,256,2,4,1,1,0,1,0,0,0,0,0>',
    'dg_fp8_gemm_duo_kernel<256,256,2,4,1,0,0,0,1,0,0,0,0>', 'dg_fp8_gemm_duo_kernel<128,256,2,4,0,0,0,0,1,1,0,0,0>', 'dg_fp8_gemm_duo_kernel<256,256,2,4,1,1,0,0,1,0,0,0,0>',
    'dg_fp8_gemm_duo_kernel<128,256,2,4,0,1,0,0,1,1,0,0,0>',
    'dg_fp8_gemm_duo_tab_fused_kernel<256,128,256,2,4>',                   # round 4: C4's 256-row walk and K-split remainder walk in one launch
    'dg_fp8_gemm_stream_kernel<64,128,1,4,6,0,1,0,0,0,0>', 'dg_fp8_gemm_stream_kernel<64,128,1,4,6,2,1,0,0,0,0>', 'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,0,0,0,0>',
    'dg_fp8_gemm_stream_kernel<64,128,1,4,6,0,1,1,0,0,0>', 'dg_fp8_gemm_stream_kernel<64,128,1,4,6,2,1,1,0,0,0>', 'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,1,0,0,0>',
    'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,0,4,0,0>',          # round 4: + four loader waves (the dense 64 x 32 pick)
    'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,1,4,0,0>',          # round 6: ... with packed scale words (the words of a K quad in the group ring)
    'dg_fp8_gemm_stream_kernel<64,128,1,4,6,0,1,0,0,1,0>',         # round 6: the in-kernel K split (stream_ks_64x128: dense 129 .. 256 rows)
    'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,0,0,1,0>',          # ... of the 64 x 32 tile (stream_ks_64x32: narrow layers at small M)
    'dg_fp8_gemm_stream_kernel<64,64,4,1,4,0,2,0,0,1,0>',          # ... of a 64 x 64 tile (stream_ks_64x64: three or more pieces at 33 .. 128 rows)
    'dg_fp8_gemm_stream_kernel<64,128,1,4,6,0,1,1,0,1,0>', 'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,1,0,1,0>',      # ... with packed scale words (e8_stream_ks_*)
    'dg_fp8_gemm_stream_kernel<64,128,1,4,6,0,1,1,0,1,1>', 'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,1,0,1,1>',
    # round 5: 3-stage ring, two workgroups per CU (FP32 scales / packed UE8M0; default and non-temporal weight policy)
    'dg_fp8_gemm_stream_kernel<64,128,1,4,3,0,1,0,0,0,0>', 'dg_fp8_gemm_stream_kernel<64,128,1,4,3,2,1,0,0,0,0>', 'dg_fp8_gemm_stream_kernel<64,128,1,4,3,0,1,1,0,0,0>',
    'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,1,0,0,1>', 'dg_fp8_gemm_stream_kernel<64,32,4,1,3,0,4,1,4,0,1>', 'dg_fp8_gemm_stream_kernel<64,128,1,4,3,0,1,1,0,0,1>', 'dg_fp8_gemm_stream_kernel<64,128,1,4,3,2,1,1,0,0,1>',      # round 6: granularity-32 stream tiles
   
    'dg_fp8_gemm_stream_kernel<64,128,1,4,3,2,1,1,0,0,0>', 'dg_fp8_gemm_stream_swiglu_kernel<3>',
    'dg_fp8_gemm_pipe_kernel<128,128,2,2,2>', 'dg_fp8_gemm_pipe_kernel<64,256,1,4,1>', 'dg_fp8_gemm_pipe_kernel<256,256,2,4,2>',
    'dg_fp8_gemm_pipe_kernel<128,256,2,4,2>', 'dg_fp8_gemm_pipe_kernel<32,256,1,4,0>', 'dg_fp8_gemm_pipe_kernel<16,256,1,4,0>',
    'dg_fp8_gemm_pipe_pc_kernel<256,256,2,4,1,0>', 'dg_fp8_gemm_pipe_pc_kernel<256,256,2,4,1,1>', 'dg_fp8_gemm_pipe_pc_kernel<192,256,2,4,1,0>',
    'dg_fp8_gemm_duo_e8_kernel<256,256,2,4,0,0,0>', 'dg_fp8_gemm_duo_e8_kernel<256,256,2,4,1,0,0>', 'dg_fp8_gemm_duo_e8_kernel<256,256,2,4,0,1,0>', 'dg_fp8_gemm_duo_e8_kernel<256,256,2,4,1,1,0>', 'dg_fp8_gemm_duo_e8_kernel<256,256,2,4,1,0,1>', 'dg_fp8_gemm_quad_e8_kernel<256,256,0,0,2,0,0,0,0,0,0>', 'dg_fp8_gemm_quad_e8_kernel<128,256,0,0,2,0,0,0,0,0,0>', 'dg_fp8_gemm_quad_e8_kernel<128,256,0,0,2,1,0,0,0,0,0>',
    'dg_fp8_gemm_quad_e8_kernel<256,256,0,0,2,0,1,0,0,0,0>', 'dg_fp8_gemm_quad_e8_kernel<256,256,0,0,2,0,2,0,0,0,0>', 'dg_fp8_gemm_quad_e8_kernel<128,256,0,0,2,0,0,1,0,0,0>',
    'dg_fp8_gemm_quad_e8_kernel<256,256,0,0,2,0,0,0,1,0,0>', 'dg_fp8_gemm_quad_e8_kernel<128,256,0,0,2,0,0,0,1,0,0>',       # round 6: scale granularity 32 (G32)
    'dg_fp8_gemm_quad_e8_kernel<256,256,0,0,2,0,0,0,0,1,0>', 'dg_fp8_gemm_quad_e8_kernel<128,256,0,0,2,0,0,0,0,1,0>',       # round 6: K-grouped, packed UE8M0 (KG)
    'dg_fp8_gemm_quad_e8_kernel<256,256,0,0,2,0,0,0,1,1,0>', 'dg_fp8_gemm_quad_e8_kernel<128,256,0,0,2,0,0,0,1,1,0>',
    'dg_fp8_gemm_quad_e8_kernel<256,256,0,0,2,0,0,0,0,1,1>', 'dg_fp8_gemm_quad_e8_kernel<256,256,0,0,2,0,0,0,1,1,1>',       # ... MN-major operands in place (MNK)
    'dg_fp8_gemm_skinny_kernel<1,4,1,0,0,0,0>', 'dg_fp8_gemm_skinny_kernel<2,4,1,0,0,0,0>', 'dg_fp8_gemm_skinny_kernel<1,4,2,0,0,0,0>',
    'dg_fp8_gemm_skinny_kernel<1,4,1,1,0,0,0>', 'dg_fp8_gemm_skinny_kernel<2,4,1,1,0,0,0>', 'dg_fp8_gemm_skinny_kernel<1,4,2,1,0,0,0>', 'dg_fp8_gemm_skinny_kernel<1,4,1,1,1,0,0>', 'dg_fp8_gemm_skinny_kernel<2,3,1,1,1,0,0>', 'dg_fp8_gemm_skinny_kernel<1,4,1,1,1,1,0>', 'dg_fp8_gemm_skinny_kernel<2,3,1,1,1,1,0>', 'dg_fp8_gemm_skinny_kernel<1,4,1,1,1,1,1>', 'dg_fp8_gemm_skinny_kernel<2,3,1,1,1,1,1>', 'dg_fp8_gemm_stream_swiglu_kernel<6>',
]


def _report_module():
    spec = importlib.util.spec_from_file_location('codegen_report', os.path.join(ROOT, 'tools', 'codegen_report.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_landing_hazard_checker_on_synthetic_streams():
    """The checker itself: a read, an overwrite or an address use of a landing register before a covering wait is flagged; a
    counted wait that leaves only YOUNGER operations outstanding releases the registers; LDS-DMA pieces have no landing registers."""
    check = _report_module().landing_hazards
    load = 'buffer_load_dwordx4 v[8:11], v2, s[4:7], 0 offen'
    dma = 'buffer_load_dwordx4 v3, s[8:11], s20 offen lds'
    assert check([load, dma, dma, 's_waitcnt vmcnt(2)', 'v_mul_f32_e32 v12, v8, v9']) == []
    assert [h[1] for h in check([load, dma, 's_waitcnt vmcnt(2)', 'v_mul_f32_e32 v12, v8, v9'])] == ['touch']      # wait too weak
    assert [h[1] for h in check([load, 'v_mov_b32_e32 v20, v10', 's_waitcnt vmcnt(0)'])] == ['touch']                 # early copy
    assert [h[1] for h in check([load, 'v_mov_b32_e32 v9, v1', 's_waitcnt vmcnt(0)'])] == ['touch']                  # early overwrite
    assert [h[1] for h in check([load, 'buffer_load_dword v30, v8, s[4:7], 0 offen', 's_waitcnt vmcnt(0)'])] == ['touch']
    assert [h[1] for h in check([load, 's_cbranch_scc1 12', 's_waitcnt vmcnt(0)'])] == ['branch']
    assert check([load, 's_waitcnt vmcnt(0) lgkmcnt(0)', 'v_mov_b32_e32 v20, v10', 's_cbranch_scc1 12']) == []
    assert check([load, 's_waitcnt lgkmcnt(0)', 'v_add_f32_e32 v1, v2, v3', 's_waitcnt vmcnt(0)', 'v_mov_b32_e32 v20, v8']) == []


def test_sgpr_vmem_hazard_checker_on_synthetic_streams():
    """VALU-written SGPR -> vector-memory read needs 5 wait states; nothing pads an inline-asm string (the round-3 wrong-tile bug)."""
    check = _report_module().sgpr_vmem_hazards
    load = 'buffer_load_dwordx4 v[108:111], v151, s[68:71], s6 offen'
    assert [h[1] for h in check(['v_readlane_b32 s6, v213, 37', load])] == ['v_readlane_b32 s6, v213, 37']
    assert check(['v_readlane_b32 s6, v213, 37', 's_nop 4', load]) == []
    assert check(['v_readlane_b32 s6, v213, 37', 's_nop 3', load]) != []
    assert check(['v_readlane_b32 s7, v213, 37', load]) == []                                   # another register
    assert check(['v_readfirstlane_b32 s70, v3', 's_mov_b32 s1, s2', 's_nop 1', load]) != []    # descriptor word, 3 states only
    assert check(['v_readfirstlane_b32 s70, v3', 's_mov_b32 s1, s2', 's_nop 3', load]) == []
    assert check(['v_cmp_lt_u32_e64 s[6:7], v1, v2', 'global_load_dword v1, v2, s[6:7]']) != []
    assert check(['s_mov_b32 s6, s9', load]) == []                                               # SALU writer: no hazard


@pytest.mark.skipif(not os.path.exists(LLVM_OBJDUMP), reason='llvm-objdump of the ROCm toolchain not available')
def test_no_spill_traffic_inside_the_k_loop_of_production_kernels():
    mod = _report_module()
    rows = {r['kernel']: r for r in mod.report()}
    missing = [k for k in PRODUCTION if k not in rows]
    assert not missing, f'kernels not found in the library: {missing}'
    for name in PRODUCTION:
        r = rows[name]
        assert r['mfma_range_instructions'] > 0, name
        assert r['scratch_in_mfma_range'] == 0, f'{name}: {r["scratch_in_mfma_range"]} scratch instructions between the first and last MFMA'
        assert r.get('vgpr_spill_count', 0) <= 16, f'{name}: {r.get("vgpr_spill_count")} VGPR spills'   # (outside the K loop: the range check above is the strict one)
        # no waterfall loop around the K loop's buffer operations: every descriptor input goes through readfirstlane (round 4: the stream
        # and pipe kernels carried ~12 extra instructions per LDS-DMA piece of the activation tile and of the scales)
        assert r['waterfalls_at_k_loop'] == 0, f'{name}: {r["waterfalls_at_k_loop"]} waterfall loops at the K loop (a descriptor is not provably uniform)'
        # the asm-load rule (HISTORY.md "A latent race"): nothing touches a landing VGPR between its buffer_load and the wait that
        # covers it, anywhere in the kernel; inside the K loop no branch is taken while such a load is in flight
        assert not r['landing_touches'], f'{name}: landing registers touched before their wait: {r["landing_touches"][:3]}'
        # (the skinny kernels' loads are ordinary compiler-tracked loads -- hipcc places their waits itself, also across its own branches)
        if 'skinny' not in name:
            assert not r['landing_branches_in_mfma_range'], f'{name}: branch with VGPR-landing loads in flight: {r["landing_branches_in_mfma_range"][:3]}'
        # a VALU-written SGPR (spill reload, readfirstlane) must not feed a vector-memory instruction within 5 wait states
        assert not r['sgpr_vmem_hazards'], f'{name}: VALU-written SGPR read by a VMEM instruction too early: {r["sgpr_vmem_hazards"][:3]}'
