"""GPU timing helpers with the reference's call signatures (``deep_gemm/testing/bench.py:7,79``).

``bench`` uses events around back-to-back launches after an L2/MALL flush; ``bench_kineto`` takes the
per-kernel device time from ``torch.profiler`` (works on ROCm through roctracer), matching kernels by a
name substring -- the HIP kernels of this library all contain ``gemm_`` or ``transpose_`` in their names.
"""
import os
import sys
from typing import Callable, Optional

import torch

# The MI355X Infinity Cache is 256 MiB: 512 MB of writes evicts L2 + MALL between timed launches.
_FLUSH_BYTES = int(512e6)


def bench(fn, num_warmups: int = 5, num_tests: int = 10, high_precision: bool = False) -> float:
    torch.cuda.synchronize()
    flush = torch.empty(_FLUSH_BYTES // 4, dtype=torch.int, device='cuda')
    for _ in range(num_warmups):
        fn()
    flush.zero_()
    if high_precision:
        # Queue a long kernel so the timed launches are not host-launch bound.
        x = torch.randn((8192, 8192), dtype=torch.float, device='cuda')
        x @ x
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(num_tests):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / num_tests / 1e3


class _Quiet:
    """Redirect the process-level stdout/stderr to /dev/null while the profiler prints."""

    def __init__(self, enabled: bool):
        self.enabled = enabled

    def __enter__(self):
        if self.enabled:
            sys.stdout.flush(), sys.stderr.flush()
            self.null = os.open(os.devnull, os.O_WRONLY)
            self.saved = (os.dup(1), os.dup(2))
            os.dup2(self.null, 1), os.dup2(self.null, 2)
        return self

    def __exit__(self, *_):
        if self.enabled:
            os.dup2(self.saved[0], 1), os.dup2(self.saved[1], 2)
            for fd in (*self.saved, self.null):
                os.close(fd)


def bench_kineto(fn, kernel_names, num_tests: int = 30, suppress_kineto_output: bool = False,
                 trace_path: str = None, flush_l2: bool = True, with_multiple_kernels: bool = False,
                 barrier: Optional[Callable] = None):
    """Average device time (seconds) of the kernel(s) whose name contains ``kernel_names``."""
    assert isinstance(kernel_names, (str, tuple))
    single = isinstance(kernel_names, str)
    names = (kernel_names, ) if single else kernel_names
    if int(os.environ.get('DG_USE_EXTERNAL_PROFILER', os.environ.get('DG_USE_NVIDIA_TOOLS', 0))):
        return 1 if single else (1, ) * len(names)       # rocprofv3 is driving: do not nest profilers

    fn()
    flush = torch.empty(_FLUSH_BYTES // 4, dtype=torch.int, device='cuda') if flush_l2 else None
    with _Quiet(suppress_kineto_output):
        schedule = torch.profiler.schedule(wait=0, warmup=1, active=1, repeat=1)
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA], schedule=schedule,
                                    acc_events=True) as prof:
            for _ in range(2):
                for _ in range(num_tests):
                    if flush is not None:
                        flush.zero_()
                    if barrier is not None:
                        barrier()
                    fn()
                torch.cuda.synchronize()
                prof.step()
    if trace_path is not None:
        prof.export_chrome_trace(trace_path)

    times = []
    events = prof.key_averages()
    for name in names:
        matched = [e for e in events if name in e.key]
        if not with_multiple_kernels:
            assert len(matched) <= 1, f'kernel name {name!r} is ambiguous: {[e.key for e in matched]}'
        total_us = sum(e.device_time_total for e in matched)
        count = sum(e.count for e in matched)
        times.append(total_us / count / 1e6 if count else 0)
    return times[0] if single else tuple(times)
