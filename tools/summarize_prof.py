#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into a short text summary (stdout)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof'
for path in sorted(glob.glob(os.path.join(root, '**', '*kernel_stats.csv'), recursive=True)):
    print('==', path)
    with open(path) as f:
        for i, row in enumerate(csv.reader(f)):
            if i < 8:
                print(','.join(row)[:220])
for path in sorted(glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True)):
    print('==', path)
    sums, counts = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            name = row.get('Kernel_Name', '')
            if 'gemm' not in name:
                continue
            key = (name[:60], row.get('Counter_Name'))
            sums[key] += float(row.get('Counter_Value', 0))
            counts[key] += 1
    for (name, counter), total in sorted(sums.items()):
        print(f'{name:60s} {counter:28s} per-dispatch={total / counts[(name, counter)]:.4g} (n={counts[(name, counter)]})')
