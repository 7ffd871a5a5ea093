#!/usr/bin/env python3
"""traffic_<config>.json (what bench.py's roofline.traffic / mfma_busy / clock_ghz read) from the rocprofv3 --pmc passes of one workload
(tools/gpu_prof_r06.sh: <workload>_pmc1 = SQ_BUSY_CYCLES + SQ_VALU_MFMA_BUSY_CYCLES, _pmc2 = FETCH_SIZE, _pmc3 = WRITE_SIZE).

    python tools/make_traffic_json.py <dir> <workload> <kernel-name substring> <config name> [git hash of the measured tree]

Per-dispatch means over all launches of the kernel.  traffic = FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md
section HBM) + WRITE_SIZE as reported (KiB).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (shader cycles x 1024 SIMDs), shader cycles =
SQ_BUSY_CYCLES / 32 (it is summed over the chip's 32 shader engines); clock_ghz = shader cycles / the kernel-trace duration of the same pass.
Works on tools/trim_profiles.py's pmc_means.csv (or the raw counter CSV)."""
import csv, glob, json, os, subprocess, sys

root, workload, needle, config = sys.argv[1:5]
git = sys.argv[5] if len(sys.argv) > 5 else subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()


def mean_of(sub, counter, want_trace=False):
    for pat, key_val in (('*means.csv', 'mean_per_dispatch'), ('*counter_collection.csv', 'Counter_Value')):
        for path in glob.glob(os.path.join(root, sub, '**', pat), recursive=True):
            total, n, trace = 0.0, 0, None
            with open(path) as f:
                for row in csv.DictReader(f):
                    if needle in row['Kernel_Name'] and row['Counter_Name'] == counter:
                        w = int(row.get('dispatches', 1) or 1) if key_val == 'mean_per_dispatch' else 1
                        total += float(row[key_val]) * w
                        n += w
                        if row.get('trace_mean_us'):
                            trace = float(row['trace_mean_us'])
            if n:
                return (total / n, n, trace) if want_trace else (total / n, n)
    raise SystemExit(f'no {counter} rows for {needle} under {root}/{sub}')


fetch, n_f = mean_of(f'{workload}_pmc2', 'FETCH_SIZE')
write, n_w = mean_of(f'{workload}_pmc3', 'WRITE_SIZE')
busy, n_b, trace_us = mean_of(f'{workload}_pmc1', 'SQ_BUSY_CYCLES', True)
mfma, _ = mean_of(f'{workload}_pmc1', 'SQ_VALU_MFMA_BUSY_CYCLES')
cycles = busy / 32.0
rec = {'kernel': config, 'workload': workload, 'git': git,
       'source': f'{root}/{workload}_pmc1..3: rocprofv3 --pmc, separate passes, per-dispatch means over all launches of the kernel',
       'correction': 'FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE taken as reported (uncalibrated)',
       'launches_in_pass': [n_b, n_f, n_w], 'fetch_size_counter_kb': fetch, 'write_size_counter_kb': write,
       'fetch_bytes': fetch * 1024 * 2, 'write_bytes': write * 1024, 'traffic_bytes': fetch * 1024 * 2 + write * 1024,
       'shader_cycles': cycles, 'mfma_busy': mfma / (cycles * 1024.0), 'kernel_trace_us_of_the_counter_pass': trace_us,
       'clock_ghz': (cycles / trace_us / 1e3) if trace_us else None}
out = os.path.join(root, f'traffic_{workload}_{config}.json')
with open(out, 'w') as f:
    json.dump(rec, f, indent=1)
print(out, round(rec['traffic_bytes'] / 1e6, 1), 'MB', 'mfma_busy', round(rec['mfma_busy'], 3), 'clock', rec['clock_ghz'])
