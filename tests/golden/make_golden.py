"""Generates the committed golden fixtures under tests/golden/ by running the REFERENCE's own Python code.

Runs only in the build container (needs /root/reference); the GPU box only reads the .npz files.  What comes from the
reference, imported with importlib straight from /root/reference (nothing is copied into this repository):
  * deep_gemm/utils/math.py      -- per_token / per_block / per_channel casts, ceil_to_ue8m0, pack_ue8m0_to_int
  * deep_gemm/testing/numeric.py -- calc_diff
  * the test oracle expression of tests/generators.py:312, ``(a.float() @ b.float().t()).to(out_dtype)``
  * tests/test_layout.py:20-44  -- get_mn_major_tma_aligned_packed_ue8m0_tensor_torch_impl, the reference's torch statement
    of the packed-UE8M0 SF layout (the function's AST node is compiled out of the reference file at generation time; the
    module itself cannot be imported because it imports the compiled extension)
The reference's CUDA kernels cannot run here (no NVIDIA GPU, no nvcc, CUTLASS submodule absent), so there are no
kernel-output goldens; each GEMM fixture instead stores the reference-quantised operands, the reference test result and
this repository's oracle output with its reference-calc_diff, which pins the oracle at the reference's own gate (< 1e-3).

    python tests/golden/make_golden.py [quantisers|gemm|layout|gran32 ...]
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)


def load_ref(rel_path: str, name: str):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel_path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


ref_math = load_ref('deep_gemm/utils/math.py', 'ref_math')
ref_numeric = load_ref('deep_gemm/testing/numeric.py', 'ref_numeric')


def bits(t: torch.Tensor) -> np.ndarray:
    if t.dtype == torch.bfloat16:
        return t.contiguous().view(torch.int16).numpy()
    if t.dtype == torch.float8_e4m3fn:
        return t.contiguous().view(torch.uint8).numpy()
    return t.contiguous().numpy()


def quantiser_fixtures():
    out = {}
    torch.manual_seed(1234)
    cases = {'tok_5x200': (5, 200), 'tok_130x384': (130, 384), 'tok_64x512': (64, 512)}
    for name, shape in cases.items():
        x = torch.randn(shape, dtype=torch.bfloat16) * 3
        x[0, :7] = 0                                     # exercises the 1e-4 amax clamp on a partial block
        out[f'{name}_x'] = bits(x)
        for ue in (False, True):
            q, sf = ref_math.per_token_cast_to_fp8(x, use_ue8m0=ue)
            out[f'{name}_ue{int(ue)}_q'], out[f'{name}_ue{int(ue)}_sf'] = bits(q), bits(sf)
    x = torch.randn((64, 512), dtype=torch.bfloat16)
    out['tok_packed_x'] = bits(x)
    q, sf = ref_math.per_token_cast_to_fp8(x, use_ue8m0=True, use_packed_ue8m0=True)
    out['tok_packed_q'], out['tok_packed_sf'] = bits(q), bits(sf)
    for name, shape in {'blk_200x300': (200, 300), 'blk_256x384': (256, 384)}.items():
        x = torch.randn(shape, dtype=torch.bfloat16) * 0.5
        out[f'{name}_x'] = bits(x)
        for ue in (False, True):
            q, sf = ref_math.per_block_cast_to_fp8(x, use_ue8m0=ue)
            out[f'{name}_ue{int(ue)}_q'], out[f'{name}_ue{int(ue)}_sf'] = bits(q), bits(sf)
    x = torch.randn((256, 96), dtype=torch.bfloat16)
    out['chn_256x96_x'] = bits(x)
    q, sf = ref_math.per_channel_cast_to_fp8(x, use_ue8m0=False)
    out['chn_256x96_q'], out['chn_256x96_sf'] = bits(q), bits(sf)
    x = torch.randn((4, 6, 64), dtype=torch.bfloat16)
    out['cus_4x6x64_x'] = bits(x)
    q, sf = ref_math.per_custom_dims_cast_to_fp8(x, (0, 1), use_ue8m0=False)
    out['cus_4x6x64_q'], out['cus_4x6x64_sf'] = bits(q), bits(sf)
    v = torch.tensor([1.0, 1.5, 2.0, 3e-5, 448.0, 0.3, 1e-38, 6e4, 2.0 ** -126, 0.0])
    out['ue8m0_in'], out['ue8m0_out'] = bits(v), bits(ref_math.ceil_to_ue8m0(v))
    np.savez_compressed(os.path.join(HERE, 'quantisers.npz'), **out)
    print('quantisers.npz', len(out), 'arrays')


def gemm_fixtures():
    import oracle
    out = {}
    # C1 of BASELINE.json: integer-valued operands (exact in e4m3), unit scales -> exact BF16 torch.matmul result.
    torch.manual_seed(0)
    a = torch.randint(-8, 9, (128, 512)).to(torch.bfloat16)
    b = torch.randint(-8, 9, (128, 512)).to(torch.bfloat16)
    out['c1_a_q'], out['c1_b_q'] = bits(a.to(torch.float8_e4m3fn)), bits(b.to(torch.float8_e4m3fn))
    out['c1_ref_d'] = bits((a.float() @ b.float().t()).to(torch.bfloat16))

    cases = {'g64x192x384': (64, 192, 384, torch.bfloat16), 'g33x200x256': (33, 200, 256, torch.bfloat16),
             'g128x128x1024': (128, 128, 1024, torch.bfloat16), 'g96x136x640_f32': (96, 136, 640, torch.float)}
    for name, (m, n, k, out_dtype) in cases.items():
        torch.manual_seed(0)
        a = torch.randn((m, k), dtype=torch.bfloat16)
        b = torch.randn((n, k), dtype=torch.bfloat16)
        ref_d = (a.float() @ b.float().t()).to(out_dtype)                 # tests/generators.py:312
        a_q, sfa = ref_math.per_token_cast_to_fp8(a, use_ue8m0=False)
        b_q, sfb = ref_math.per_block_cast_to_fp8(b, use_ue8m0=False)
        d = torch.empty((m, n), dtype=out_dtype)
        oracle.fp8_gemm_nt(a_q, sfa, b_q, sfb, d)
        diff = float(ref_numeric.calc_diff(d, ref_d))
        assert diff < 1e-3, (name, diff)
        out[f'{name}_a_q'], out[f'{name}_sfa'] = bits(a_q), bits(sfa)
        out[f'{name}_b_q'], out[f'{name}_sfb'] = bits(b_q), bits(sfb)
        out[f'{name}_ref_d'], out[f'{name}_oracle_d'] = bits(ref_d), bits(d)
        out[f'{name}_ref_calc_diff'] = np.float64(diff)
        print(name, 'reference calc_diff(oracle, ref_d) =', diff)
    np.savez_compressed(os.path.join(HERE, 'gemm_cases.npz'), **out)
    print('gemm_cases.npz', len(out), 'arrays')


def gran32_fixtures():
    """Scale granularity 32 along K with UE8M0 scales (the reference's SM100 MX recipe for FP8 x FP8: csrc/apis/gemm.hpp:311-312,
    tests/generators.py:192-194, 230: both operands through per_token_cast_to_fp8(..., gran_k=32)): the reference quantiser's bytes, its packed
    words, and one GEMM case with the reference test expression and this repository's oracle at gran_k = 32."""
    import oracle
    out = {}
    torch.manual_seed(4321)
    for name, shape in {'tok32_5x200': (5, 200), 'tok32_130x384': (130, 384)}.items():
        x = torch.randn(shape, dtype=torch.bfloat16) * 3
        x[0, :7] = 0
        out[f'{name}_x'] = bits(x)
        q, sf = ref_math.per_token_cast_to_fp8(x, use_ue8m0=True, gran_k=32)
        out[f'{name}_q'], out[f'{name}_sf'] = bits(q), bits(sf)
    x = torch.randn((64, 512), dtype=torch.bfloat16)
    out['tok32_packed_x'] = bits(x)
    q, sf = ref_math.per_token_cast_to_fp8(x, use_ue8m0=True, gran_k=32, use_packed_ue8m0=True)
    out['tok32_packed_q'], out['tok32_packed_sf'] = bits(q), bits(sf)
    m, n, k = 96, 136, 640
    torch.manual_seed(0)
    a = torch.randn((m, k), dtype=torch.bfloat16)
    b = torch.randn((n, k), dtype=torch.bfloat16)
    ref_d = (a.float() @ b.float().t()).to(torch.bfloat16)                # tests/generators.py:312
    a_q, sfa = ref_math.per_token_cast_to_fp8(a, use_ue8m0=True, gran_k=32)
    b_q, sfb = ref_math.per_token_cast_to_fp8(b, use_ue8m0=True, gran_k=32)
    d = torch.empty((m, n), dtype=torch.bfloat16)
    oracle.fp8_gemm_nt(a_q, sfa, b_q, sfb, d, gran_n=1, gran_k=32)
    diff = float(ref_numeric.calc_diff(d, ref_d))
    assert diff < 1e-3, diff
    out['g32_a_q'], out['g32_sfa'], out['g32_b_q'], out['g32_sfb'] = bits(a_q), bits(sfa), bits(b_q), bits(sfb)
    out['g32_ref_d'], out['g32_oracle_d'], out['g32_ref_calc_diff'] = bits(ref_d), bits(d), np.float64(diff)
    print('gran_k = 32: reference calc_diff(oracle, ref_d) =', diff)
    np.savez_compressed(os.path.join(HERE, 'gran32.npz'), **out)
    print('gran32.npz', len(out), 'arrays')


def load_ref_function(rel_path: str, name: str, namespace: dict):
    """Compiles ONE top-level function out of a reference file that cannot be imported as a whole."""
    import ast
    with open(os.path.join(REF, rel_path)) as f:
        tree = ast.parse(f.read())
    node = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module(body=[node], type_ignores=[]), os.path.join(REF, rel_path), 'exec')
    exec(code, namespace)
    return namespace[name]


def layout_fixtures():
    # get_tma_aligned_size lives in the compiled extension; its definition (csrc/utils/math.hpp:23-27) is align(x, 16 / elem)
    namespace = {'torch': torch, 'align': ref_math.align,
                 'get_tma_aligned_size': lambda x, elem: ref_math.align(x, 16 // elem)}
    ref_pack = load_ref_function('tests/test_layout.py', 'get_mn_major_tma_aligned_packed_ue8m0_tensor_torch_impl', namespace)
    out = {}
    torch.manual_seed(7)
    for name, shape in {'p33x7': (33, 7), 'p128x56': (128, 56), 'p3x20x9': (3, 20, 9), 'p2x64x4': (2, 64, 4)}.items():
        x = torch.randn(shape[:-1] + (shape[-1] * 128,), dtype=torch.bfloat16).reshape(-1, shape[-1] * 128)
        _, sf = ref_math.per_token_cast_to_fp8(x, use_ue8m0=True)
        sf = sf.view(shape)
        packed = ref_pack(sf)
        out[f'{name}_sf'] = bits(sf)
        out[f'{name}_packed'] = packed.contiguous().numpy()             # logical [.., mn, packed_k] values
        out[f'{name}_strides'] = np.array(packed.stride(), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, 'sf_layout.npz'), **out)
    print('sf_layout.npz', len(out), 'arrays')


if __name__ == '__main__':
    makers = {'quantisers': quantiser_fixtures, 'gemm': gemm_fixtures, 'layout': layout_fixtures, 'gran32': gran32_fixtures}
    for which in sys.argv[1:] or list(makers):               # (python tests/golden/make_golden.py [quantisers|gemm|layout|gran32 ...])
        makers[which]()
