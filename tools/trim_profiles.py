#!/usr/bin/env python3
"""Replaces raw rocprofv3 per-dispatch CSVs under profiles/ (pmc_counter_collection.csv, *_kernel_trace.csv of PMC passes) by
their per-kernel means (pmc_means.csv next to where the raw file was): what SUMMARY.txt / the traffic JSONs are computed from.

    python tools/trim_profiles.py [profiles]          # idempotent
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else 'profiles'
for path in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    sums, counts = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            key = (row.get('Kernel_Name', ''), row.get('Counter_Name'))
            sums[key] += float(row.get('Counter_Value', 0) or 0)
            counts[key] += 1
    trace = path.replace('counter_collection', 'kernel_trace')
    durs = defaultdict(list)
    if os.path.exists(trace):
        with open(trace) as f:
            for row in csv.DictReader(f):
                durs[row['Kernel_Name']].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
    out = os.path.join(os.path.dirname(path), os.path.basename(path).replace('counter_collection', 'means'))
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        w.writerow(['Kernel_Name', 'Counter_Name', 'mean_per_dispatch', 'dispatches', 'trace_mean_us', 'trace_min_us', 'trace_max_us'])
        for (name, counter), total in sorted(sums.items()):
            d = durs.get(name)
            w.writerow([name, counter, f'{total / counts[(name, counter)]:.6g}', counts[(name, counter)]] +
                       ([f'{sum(d) / len(d):.3f}', f'{min(d):.3f}', f'{max(d):.3f}'] if d else ['', '', '']))
    os.remove(path)
    if os.path.exists(trace):
        os.remove(trace)
    print('trimmed', path)
