#!/bin/bash
# Evidence of the second half of round 6 (K-grouped GEMM with UE8M0 scales, single-body shifted-scale loop): logs under gpurun_out/r06_kg/
OUT=gpurun_out/r06_kg; mkdir -p $OUT
git rev-parse HEAD > $OUT/git_hash.txt 2>/dev/null
timeout 600 python tools/fuzz_k_grouped_ue8m0.py 80 0 > $OUT/fuzz_k_grouped_ue8m0.log 2>&1; tail -1 $OUT/fuzz_k_grouped_ue8m0.log
timeout 300 python tools/probes/kgrouped_ue8m0_probe.py > $OUT/kgrouped_ue8m0_ab.log 2>&1
timeout 300 python tools/probes/kgrouped_fit_probe.py > $OUT/kgrouped_fit_probe.log 2>&1
(timeout 200 python tools/probes/kgrouped_phase_stamps.py 4096; timeout 200 python tools/probes/kgrouped_phase_stamps.py 1024) > $OUT/kgrouped_phase_stamps.log 2>&1
timeout 300 python tools/probes/wgrad_ue8m0_probe.py > $OUT/wgrad_ue8m0_probe.log 2>&1
timeout 600 tools/probes/g32_loop_ab.sh > $OUT/g32_single_body_loop_ab.log 2>&1
for f in $OUT/*.log; do echo "== $f"; tail -n 3 $f; done
