"""Ahead-of-time build of the HIP extension (`libdeepgemm_amd.so`) for gfx950.

There is no JIT: the reference's NVCC/NVRTC runtime (csrc/jit/) is replaced by one `hipcc` invocation whose output is
kept in-tree next to the sources, so that it travels with the repository snapshot to the GPU box.
"""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.path.join(CSRC, 'libdeepgemm_amd.so')
SOURCES = ['dg_api.hip']
HEADERS = ['fp8_gemm_kernels.hpp', 'fp8_gemm_quad.hpp', 'fp8_gemm_moe.hpp', 'fp8_gemm_experiments.hpp', os.path.join('..', '..', 'include', 'deepgemm_amd.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-shared', '-fPIC', '-fno-slp-vectorize']
# DG_EXPERIMENTS=1: also build the timing ablations / rejected kernel variants that HISTORY.md quotes (tools/cycles.py,
# tools/sustained.py, tools/trace*.py take their names); they roughly triple the compile time and are never selected.
if os.environ.get('DG_EXPERIMENTS', '') not in ('', '0'):
    FLAGS.append('-DDG_EXPERIMENTS')


# Tuning aid (tools/ A/B runs of two builds in one GPU session): DG_VARIANT=<tag> builds csrc/libdeepgemm_amd_<tag>.so with the extra
# compiler flags of DG_VARIANT_FLAGS (e.g. -DDG_NT_STORES) and the package loads THAT library.  Unset = the product library.
_VARIANT = os.environ.get('DG_VARIANT', '')
if _VARIANT:
    LIB_PATH = os.path.join(CSRC, f'libdeepgemm_amd_{_VARIANT}.so')
    FLAGS = FLAGS + os.environ.get('DG_VARIANT_FLAGS', '').split()

STAMP_PATH = LIB_PATH + '.flags'      # the flag set the library was built with (DG_EXPERIMENTS toggles must rebuild)

# The fast kernels rely on properties of hipcc's code generation that the language does not promise (registers written by inline-asm
# loads are not touched before the hand-placed wait; no spill inside a K loop).  tests/test_codegen.py checks them on the built
# library; they were validated with this toolchain.  Another version still builds -- with a warning to re-run that test.
VALIDATED_HIP_VERSIONS = ('7.2',)


def _hipcc_version() -> str:
    try:
        out = subprocess.run([HIPCC, '--version'], capture_output=True, text=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        return 'unknown'
    for line in out.splitlines():
        if line.startswith('HIP version:'):
            return line.split(':', 1)[1].strip()
    return 'unknown'


def _stamp() -> str:
    return ' '.join([HIPCC, _hipcc_version(), *FLAGS])


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    if any(os.path.getmtime(os.path.join(CSRC, f)) > built for f in SOURCES + HEADERS):
        return True
    try:
        with open(STAMP_PATH) as f:
            return f.read() != _stamp()
    except OSError:
        # a library without a stamp (built by hand or shipped prebuilt): trust it unless experiments are asked for
        return '-DDG_EXPERIMENTS' in FLAGS


def build_extension(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        tmp = LIB_PATH + f'.{os.getpid()}.tmp'
        cmd = [HIPCC, *FLAGS, *SOURCES, '-o', tmp]
        version = _hipcc_version()
        if not version.startswith(VALIDATED_HIP_VERSIONS):
            print(f'deepgemm_amd: building with HIP {version}; the kernels\' code-generation assumptions were validated with '
                  f'{", ".join(VALIDATED_HIP_VERSIONS)} -- run tests/test_codegen.py on the result', file=sys.stderr)
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.check_call(cmd, cwd=CSRC)
        os.replace(tmp, LIB_PATH)
        with open(STAMP_PATH, 'w') as f:
            f.write(_stamp())
    return LIB_PATH


if __name__ == '__main__':
    print(build_extension(force='--force' in sys.argv, verbose=True))
