#!/bin/bash
# FP32 reduce-add epilogue of the quad kernels: cache policy of the old values' loads / of the stores (DG_VARIANT=nt1 | nt2 | nt3 built with
# -DDG_RMW_NT=1 | 2 | 3) against the product, K-grouped in-place call and dense weight gradient, alternating on one box
for i in 1 2; do
  for v in "" nt1 nt2 nt3; do
    echo "== variant '${v:-product}'"
    DG_VARIANT=$v KG_MN=1 python tools/probes/kgrouped_fit_probe.py 2>&1 | grep -v amdgpu | tail -n 3
    DG_VARIANT=$v python tools/probes/wgrad_ue8m0_probe.py 2>&1 | grep "packed " | head -n 1
  done
done
