#!/bin/bash
# Extra evidence for the secondary kernels: kernel-trace stats of the contiguous and masked bench workloads, of the
# recipe-(1,1,128) kernel and of the fused quantiser, plus FETCH_SIZE / WRITE_SIZE passes (own runs, counters + kernel
# trace only) of the quantiser.  OUT=profiles-style directory name.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${OUT:-prof_extra}
mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
for W in contiguous masked; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -o bench -- python bench.py --workload $W --steps 50 --warmup 10 --no-cpu-baseline > $OUT/bench_$W.log 2>&1
  echo "stats $W exit $?"
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_wgrad -o wgrad -- python tools/wgrad_bench.py 4096x4096x7168 auto > $OUT/wgrad.log 2>&1
echo "stats wgrad exit $?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_quant -o quant -- python tools/quant_bench.py 4096x7168,16384x7168 > $OUT/quant.log 2>&1
echo "stats quant exit $?"
for PMC in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc_quant_$PMC -o pmc -- python tools/quant_bench.py 16384x7168 > $OUT/pmc_quant_$PMC.log 2>&1
  echo "pmc quant $PMC exit $?"
done
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete
find $OUT -type f -size +2M -delete
for d in $OUT/stats_*; do
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { echo "== $d"; head -6 "$f" | cut -c1-220; }
done > $OUT/SUMMARY.txt
python - >> $OUT/SUMMARY.txt <<'PY'
import csv, glob, os
out = os.environ.get('OUT_DIR', None)
for pmc in ('FETCH_SIZE', 'WRITE_SIZE'):
    for f in glob.glob(f"gpurun_out/*/pmc_quant_{pmc}/**/*counter_collection.csv", recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'per_token_cast' in r.get('Kernel_Name', '')]
        vals = [float(r['Counter_Value']) for r in rows if r.get('Counter_Name') == pmc]
        if vals:
            # median launch; FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1 KB per the guide; gfx950 FETCH needs x2
            vals.sort()
            print(f"quantiser 16384x7168 {pmc}: launches={len(vals)} median={vals[len(vals)//2]:.1f} (raw counter units)")
PY
cat $OUT/SUMMARY.txt
