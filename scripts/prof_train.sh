#!/bin/bash
# rocprofv3 kernel-trace summary of the adapter training step benchmark (scripts/train_bench.py).
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o train --output-format csv -- python3 scripts/train_bench.py > gpurun_out/prof_train.log 2>&1
grep -h "^{\"metric\"" gpurun_out/prof_train.log > gpurun_out/prof_train_line.json
cp gpurun_out/prof_train/train_kernel_stats.csv gpurun_out/prof_train_kernel_stats.csv
rm -rf gpurun_out/prof_train
head -30 gpurun_out/prof_train_kernel_stats.csv | cut -c1-200
