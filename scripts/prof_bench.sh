#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (the numbers bench.py's roofline is checked against).
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench -o bench --output-format csv -- python3 bench.py > gpurun_out/prof_bench.log 2>&1
grep -h "^{\"metric\"" gpurun_out/prof_bench.log > gpurun_out/prof_bench_line.json
cp gpurun_out/prof_bench/bench_kernel_stats.csv gpurun_out/prof_bench_kernel_stats.csv
rm -rf gpurun_out/prof_bench
head -12 gpurun_out/prof_bench_kernel_stats.csv | cut -c1-160
