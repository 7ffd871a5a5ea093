#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary (stdout)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof'


def is_gemm(name: str) -> bool:
    return 'gemm' in name.lower() or 'Cijk_' in name      # (this library's kernels; hipBLASLt's Tensile kernels of the comparator runs)


for path in sorted(glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True)):
    print('==', path)
    with open(path) as f:
        for i, row in enumerate(csv.reader(f)):
            if i == 0 or i < 6 or is_gemm(row[0]):
                print(','.join(row)[:240])
for path in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    print('==', path)
    sums, counts = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get('Kernel_Name', '')
            if not is_gemm(name):
                continue
            key = (name[:60], row.get('Counter_Name'))
            sums[key] += float(row.get('Counter_Value', 0))
            counts[key] += 1
    for (name, counter), total in sorted(sums.items()):
        print(f'{name:60s} {counter:28s} per-dispatch={total / counts[(name, counter)]:.6g} (n={counts[(name, counter)]})')
    trace = path.replace('counter_collection', 'kernel_trace')
    if os.path.exists(trace):
        durs = defaultdict(list)
        with open(trace) as f:
            for row in csv.DictReader(f):
                if is_gemm(row['Kernel_Name']):
                    durs[row['Kernel_Name'][:60]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
        for name, d in durs.items():
            print(f'{name:60s} kernel-trace duration us: mean={sum(d) / len(d):.2f} min={min(d):.2f} max={max(d):.2f} (n={len(d)})')
