#!/bin/bash
mkdir -p gpurun_out/r2p
for s in 128x4096x7168 1x7168x16384; do timeout 100 python tools/variant_check.py stream_pf4_64x128,stream_ntpf8_64x128,stream_pf2_64x32 $s stream_64x128 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2p/bitcheck.log
timeout 300 python tools/masked_bench.py stream_64x128,stream_nt_64x128,stream_pf4_64x128,stream_pf8_64x128,stream_ntpf4_64x128,stream_ntpf8_64x128 32x20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2p/masked32.log
timeout 300 python tools/masked_bench.py stream_64x128,stream_nt_64x128,stream_pf4_64x128,stream_pf8_64x128,stream_ntpf4_64x128 6x20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2p/masked6.log
timeout 300 python tools/sweep.py --shapes 128x4096x7168,1x7168x16384,128x7168x16384,1x4096x7168,128x2112x7168 --configs stream_64x32,stream_pf1_64x32,stream_pf2_64x32,stream_64x128,stream_pf4_64x128,stream_pf8_64x128 --rounds 3 --iters 20 --sets 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2p/dense.log
