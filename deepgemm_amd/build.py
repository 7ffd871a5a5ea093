"""Ahead-of-time build of the HIP extension (`libdeepgemm_amd.so`) for gfx950.

There is no JIT: the reference's NVCC/NVRTC runtime (csrc/jit/) is replaced by an ahead-of-time `hipcc` build whose output is
kept in-tree next to the sources, so that it travels with the repository snapshot to the GPU box.

The build is sharded (round 6): `dg_api.hip` (host code + the plain kernels) and `NUM_SHARDS` instantiation units
(`dg_shard.hip -DDG_SHARD=n`, the template kernels listed in `csrc/kernel_instances.inc`) are compiled in parallel from a private
snapshot of the sources and linked into one library -- one device pass over all instantiations took six minutes on this container.
`DG_MONOLITHIC=1` restores the single translation unit.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB_PATH = os.path.join(CSRC, 'libdeepgemm_amd.so')
SOURCES = ['dg_api.hip', 'dg_shard.hip', 'kernel_instances.inc']
NUM_SHARDS = 10                         # shard ids 0 .. NUM_SHARDS - 1 of kernel_instances.inc
MONOLITHIC = os.environ.get('DG_MONOLITHIC', '') not in ('', '0')
# Per-shard compiler flags.  Shard 9 (the K-grouped quad kernels that read MN-major operands in place): their block body holds ~130 inline-asm
# statements more than the K-major form's, which takes the fully unrolled K-block loops over LLVM's `#pragma unroll` cost limit (16 K units); past
# it the pragma is silently ignored, the accumulator indices stay dynamic and hipcc keeps all 256 accumulators in scratch (agpr_count 16, 1 200
# scratch operations in the loop) -- the mechanism behind every "one more copy of the block body and the accumulators go to memory" note in the
# sources.  The other shards are compiled as they always were.
SHARD_FLAGS = {9: ['-mllvm', '-pragma-unroll-threshold=200000']}
HEADERS = ['fp8_gemm_kernels.hpp', 'fp8_gemm_quad.hpp', 'fp8_gemm_moe.hpp', os.path.join('..', '..', 'include', 'deepgemm_amd.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-slp-vectorize'] + \
        (['-DDG_MONOLITHIC', '-mllvm', '-pragma-unroll-threshold=200000'] if MONOLITHIC else [])      # (one unit: shard 9's flag for all, see SHARD_FLAGS)
# DG_EXPERIMENTS=1: also build the timing ablations / rejected kernel variants that HISTORY.md quotes (tools/cycles.py,
# tools/sustained.py, tools/trace*.py take their names); they roughly triple the compile time and are never selected.
# Their sources live outside the product: tools/experiments/ (no product build reads them).
if os.environ.get('DG_EXPERIMENTS', '') not in ('', '0'):
    _EXPERIMENTS = os.path.join('..', '..', 'tools', 'experiments')
    FLAGS += ['-DDG_EXPERIMENTS', '-I', _EXPERIMENTS, '-I', '.']
    HEADERS += [os.path.join(_EXPERIMENTS, 'fp8_gemm_experiments.hpp'), os.path.join(_EXPERIMENTS, 'experiment_configs.inc')]


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
    return ' '.join([HIPCC, _hipcc_version(), *FLAGS, *(f'shard{n}:' + ','.join(v) for n, v in sorted(SHARD_FLAGS.items()))])


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


def _snapshot(root: str) -> str:
    """A private copy of everything the compiler reads (same relative layout), so that the sources may be edited while a build runs."""
    repo = os.path.dirname(os.path.dirname(CSRC))
    dst_csrc = os.path.join(root, 'deepgemm_amd', 'csrc')
    os.makedirs(dst_csrc)
    for name in os.listdir(CSRC):
        if name.endswith(('.hip', '.hpp', '.inc', '.h')):
            shutil.copy2(os.path.join(CSRC, name), dst_csrc)
    shutil.copytree(os.path.join(repo, 'include'), os.path.join(root, 'include'))
    experiments = os.path.join(repo, 'tools', 'experiments')
    if os.path.isdir(experiments):
        shutil.copytree(experiments, os.path.join(root, 'tools', 'experiments'))
    return dst_csrc


def build_extension(force: bool = False, verbose: bool = False) -> str:
    if force or is_stale():
        version = _hipcc_version()
        if not version.startswith(VALIDATED_HIP_VERSIONS):
            print(f'deepgemm_amd: building with HIP {version}; the kernels\' code-generation assumptions were validated with '
                  f'{", ".join(VALIDATED_HIP_VERSIONS)} -- run tests/test_codegen.py on the result', file=sys.stderr)
        stamp = _stamp()
        with tempfile.TemporaryDirectory(prefix='dg_build_') as root:
            cwd = _snapshot(root)
            units = [('dg_api.o', ['dg_api.hip'])]
            if not MONOLITHIC:
                units += [(f'dg_shard{n}.o', [*SHARD_FLAGS.get(n, []), f'-DDG_SHARD={n}', 'dg_shard.hip']) for n in range(NUM_SHARDS)]

            def compile_unit(unit):
                cmd = [HIPCC, *FLAGS, '-c', *unit[1], '-o', unit[0]]
                if verbose:
                    print(' '.join(cmd), file=sys.stderr)
                subprocess.check_call(cmd, cwd=cwd)
                return unit[0]

            jobs = int(os.environ.get('DG_BUILD_JOBS', '0')) or min(len(units), os.cpu_count() or 1)
            with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
                objects = list(pool.map(compile_unit, units))
            tmp = os.path.join(cwd, 'libdeepgemm_amd.so')
            link = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', *objects, '-o', tmp]
            if verbose:
                print(' '.join(link), file=sys.stderr)
            subprocess.check_call(link, cwd=cwd)
            staged = LIB_PATH + f'.{os.getpid()}.tmp'
            shutil.copy2(tmp, staged)
            os.replace(staged, LIB_PATH)
        with open(STAMP_PATH, 'w') as f:
            f.write(stamp)
    return LIB_PATH


if __name__ == '__main__':
    print(build_extension(force='--force' in sys.argv, verbose=True))
