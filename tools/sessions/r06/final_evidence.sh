#!/bin/bash
# Final evidence of round 6 on the final tree -> gpurun_out/r06_final (copied to profiles/r06_final): the whole GPU suite, the default bench line,
# randomised parity runs, the K-grouped UE8M0 logs, then tools/gpu_prof_r06.sh (kernel-trace stats + PMC passes per workload).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_final; mkdir -p $OUT
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
git rev-parse HEAD > $OUT/git_hash.txt 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_suite_final_tree.txt 2>&1; echo "gpu suite exit $?"; tail -n 2 $OUT/gpu_suite_final_tree.txt
timeout 600 python bench.py > $OUT/bench_default.log 2>&1; echo "bench exit $?"
grep '^{' $OUT/bench_default.log | tail -n 1 > $OUT/bench_default_run_final_tree.json
grep '^secondary_detail' $OUT/bench_default.log > $OUT/bench_default_run_secondary_detail.txt
(timeout 300 python tools/fuzz_k_grouped_ue8m0.py 120 1000; timeout 400 python tools/fuzz_dense.py 5000 40 2>&1 | tail -n 3; timeout 400 python tools/fuzz_grouped.py 5000 40 2>&1 | tail -n 3) > $OUT/fuzz_final_tree.log 2>&1; tail -n 1 $OUT/fuzz_final_tree.log
timeout 300 python tools/probes/kgrouped_ue8m0_probe.py > $OUT/kgrouped_ue8m0_ab.log 2>&1
(KG_MN=1 timeout 200 python tools/probes/kgrouped_fit_probe.py; timeout 200 python tools/probes/kgrouped_fit_probe.py) > $OUT/kgrouped_fit_probe.log 2>&1
(KG_MN=1 timeout 200 python tools/probes/kgrouped_phase_stamps.py 4096) > $OUT/kgrouped_phase_stamps_in_place.log 2>&1
SKIP_DEFAULT= OUT=r06_final WORKLOADS="dense c3_nt contiguous dense_ue8m0 kgrouped_ue8m0" bash tools/gpu_prof_r06.sh
