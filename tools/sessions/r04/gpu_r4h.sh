#!/bin/bash
# Round 4, GPU session H: where do the mid-M stream tiles' bytes come from?  Cache policy of the weight stream + fabric-side traffic counters.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
OUT=gpurun_out/r4h; mkdir -p $OUT
timeout 300 python tools/sweep.py --rounds 5 --iters 20 --configs stream_l8_64x32,stream_nt_l8_64x32,stream_sc_l8_64x32,stream_ntsc_l8_64x32 \
    --shapes 128x4096x7168,64x4096x7168,256x4096x7168,128x7168x2048 2>&1 | grep -v amdgpu.ids | python -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print(r['shape'], r['config'], r.get('us_median'), r.get('us_min'), r.get('ok', r.get('error')))
" | tee $OUT/policy.log
for cfg in stream_l8_64x32 stream_nt_l8_64x32; do
  i=0
  for PMC in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/${cfg}_pmc$i -o pmc -- \
        python tools/sweep.py --rounds 1 --iters 10 --configs $cfg --shapes 128x4096x7168 > $OUT/${cfg}_pmc$i.log 2>&1
    echo "$cfg pmc$i ($PMC) exit $?"
  done
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r4h/*_pmc*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'stream' in row.get('Kernel_Name', ''):
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            print(d.split('/')[-2], k, 'n=%d' % len(v), 'mean=%.4g' % (sum(v) / len(v)))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +2M -delete
