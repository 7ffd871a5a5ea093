#!/usr/bin/env python3
"""traffic_<kernel>.json (what bench.py's roofline.traffic reads) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one workload.

    python tools/make_traffic_json.py <dir> <fetch pass subdir> <write pass subdir> <kernel-name substring> <config name>

Per-dispatch means over all launches of the kernel; FETCH_SIZE x 2 (gfx950: 128-byte requests tallied at 64 B, MI355X_MICROARCH.md
section HBM); WRITE_SIZE as reported.  Works on the raw counter CSV or on tools/trim_profiles.py's pmc_means.csv."""
import csv, glob, json, os, sys

root, fetch_dir, write_dir, needle, config = sys.argv[1:6]


def mean_of(sub, counter):
    for pat, key_val in (('*counter_collection.csv', 'Counter_Value'), ('*means.csv', 'mean_per_dispatch')):
        for path in glob.glob(os.path.join(root, sub, '**', pat), recursive=True):
            total, n = 0.0, 0
            with open(path) as f:
                for row in csv.DictReader(f):
                    if needle in row['Kernel_Name'] and row['Counter_Name'] == counter:
                        w = int(row.get('dispatches', 1) or 1) if key_val == 'mean_per_dispatch' else 1
                        total += float(row[key_val]) * w
                        n += w
            if n:
                return total / n, n
    raise SystemExit(f'no {counter} rows for {needle} under {root}/{sub}')


fetch, n_f = mean_of(fetch_dir, 'FETCH_SIZE')
write, n_w = mean_of(write_dir, 'WRITE_SIZE')
rec = {'kernel': config,
       'source': f'{root}/{fetch_dir} (FETCH_SIZE) and {root}/{write_dir} (WRITE_SIZE): rocprofv3 --pmc, separate passes, per-dispatch means over all '
                 f'launches of the kernel',
       'correction': 'FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md section HBM); WRITE_SIZE taken as reported (uncalibrated)',
       'launches_in_pass': [n_f, n_w], 'fetch_size_counter_kb': fetch, 'write_size_counter_kb': write,
       'fetch_bytes': fetch * 1024 * 2, 'write_bytes': write * 1024, 'traffic_bytes': fetch * 1024 * 2 + write * 1024}
out = os.path.join(root, f'traffic_{config}.json')
with open(out, 'w') as f:
    json.dump(rec, f, indent=1)
print(out, round(rec['traffic_bytes'] / 1e6, 1), 'MB')
