"""Static guard on what hipcc made of the production kernels (no GPU needed: the library cross-compiles here).

The fast kernels sit at the edge of the register file; a spill that lands inside the K loop costs far more than its
instruction (scratch traffic counts towards vmcnt and tightens every counted wait) and does not show up in any
correctness test -- only in sustained throughput.  It happened once ("Measure sustained, and watch the register
allocator"): this test keeps it from happening silently again."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM_OBJDUMP = '/opt/rocm/lib/llvm/bin/llvm-objdump'

PRODUCTION = [
    'dg_fp8_gemm_duo_kernel<256,256,2,4,0>', 'dg_fp8_gemm_duo_kernel<256,256,2,4,20>', 'dg_fp8_gemm_duo_kernel<128,256,2,4,0>',
    'dg_fp8_gemm_duo_kernel<256,256,2,4,41>', 'dg_fp8_gemm_duo_kernel<128,256,2,4,40>', 'dg_fp8_gemm_duo_kernel<256,256,2,4,40>',
    'dg_fp8_gemm_stream_kernel<64,128,1,4,6,0,1>', 'dg_fp8_gemm_stream_kernel<64,32,4,1,12,0,1>',
    'dg_fp8_gemm_pipe_kernel<128,128,2,2,2,0>', 'dg_fp8_gemm_pipe_kernel<64,256,1,4,1,0>', 'dg_fp8_gemm_pipe_kernel<256,256,2,4,2,0>',
    'dg_fp8_gemm_pipe_pc_kernel<256,256,2,4,2,0,0>', 'dg_fp8_gemm_pipe_pc_kernel<256,256,2,4,2,0,1>',
    'dg_fp8_gemm_duo_e8_kernel<256,256,2,4>',
]


@pytest.mark.skipif(not os.path.exists(LLVM_OBJDUMP), reason='llvm-objdump of the ROCm toolchain not available')
def test_no_spill_traffic_inside_the_k_loop_of_production_kernels():
    spec = importlib.util.spec_from_file_location('codegen_report', os.path.join(ROOT, 'tools', 'codegen_report.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = {r['kernel']: r for r in mod.report()}
    missing = [k for k in PRODUCTION if k not in rows]
    assert not missing, f'kernels not found in the library: {missing}'
    for name in PRODUCTION:
        r = rows[name]
        assert r['mfma_range_instructions'] > 0, name
        assert r['scratch_in_mfma_range'] == 0, f'{name}: {r["scratch_in_mfma_range"]} scratch instructions between the first and last MFMA'
        assert r.get('vgpr_spill_count', 0) <= 8, f'{name}: {r.get("vgpr_spill_count")} VGPR spills'
