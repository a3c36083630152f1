#!/bin/bash
# Kernel trace of prefill passes (bench_ops.py --prefill --short).  usage: prof_prefill.sh <tag> [key=value tunings]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag=${1:-pre}
shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python3 scripts/bench_ops.py vicuna-7b --prefill --short "$@" > gpurun_out/prof_$tag.log 2>&1
f=$(ls gpurun_out/prof_$tag/*kernel_stats.csv 2>/dev/null || ls gpurun_out/prof_$tag/*/*kernel_stats.csv)
grep -v "at::native" $f | cut -c1-170 | head -16
cp $f gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
