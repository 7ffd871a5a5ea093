#!/bin/bash
# A/B of two builds on ONE box: $AB_CMD with the in-tree library ("new"), then with libdeepgemm_amd.base.so swapped in ("base"),
# alternating $AB_ROUNDS times (box-to-box variance is 2-3 %, larger than most deltas worth keeping).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONUNBUFFERED=1
L=deepgemm_amd/csrc/libdeepgemm_amd.so
cp $L /tmp/new.so
cp deepgemm_amd/csrc/libdeepgemm_amd.base.so /tmp/base.so
mkdir -p gpurun_out/ab
: > gpurun_out/ab/log.txt
for r in $(seq 1 ${AB_ROUNDS:-2}); do
  for which in new base; do
    cp /tmp/$which.so $L
    echo "== $which (round $r)" | tee -a gpurun_out/ab/log.txt
    bash -c "$AB_CMD" 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/ab/log.txt
  done
done
cp /tmp/new.so $L
