#!/bin/bash
mkdir -p gpurun_out/r2h gpurun_out/r02_prof
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > gpurun_out/r2h/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r2h/pytest.log; tail -3 gpurun_out/r2h/pytest.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof/stats_headline -o bench -- python bench.py --no-secondary > gpurun_out/r02_prof/bench_stats_headline.log 2>&1
find gpurun_out/r02_prof/stats_headline -type f ! -name "*stats.csv" -delete
head -3 gpurun_out/r02_prof/stats_headline/*/bench_kernel_stats.csv 2>/dev/null || head -3 gpurun_out/r02_prof/stats_headline/bench_kernel_stats.csv
grep '^{' gpurun_out/r02_prof/bench_stats_headline.log | cut -c1-400
timeout 300 python tools/sustained.py duo_p_256x256,duo_256x256,duo_128x256,pipe_128x128,pipe_128x256,pipe_256x256,stream_64x128 2048x7168x2048 300 3 > gpurun_out/r2h/c3_tiles.log 2>&1; cat gpurun_out/r2h/c3_tiles.log
timeout 300 python tools/grouped_bench.py --cases 8x512x4096x7168 --configs auto,duo_128x256,pipe_128x128,pipe_128x256,duo_256x256,pipe_64x256 --iters 20 > gpurun_out/r2h/c4_tiles.log 2>&1; cat gpurun_out/r2h/c4_tiles.log
