#!/bin/bash
# Kernel trace of one decode (bench_ops.py --decode): per-kernel totals against the loop's wall time.  usage: prof_decode.sh <tag> [key=value tunings]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag=${1:-dec}
shift
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python3 scripts/bench_ops.py vicuna-7b --decode "$@" > gpurun_out/prof_$tag.log 2>&1
tail -4 gpurun_out/prof_$tag.log
f=$(ls gpurun_out/prof_$tag/*kernel_stats.csv 2>/dev/null || ls gpurun_out/prof_$tag/*/*kernel_stats.csv)
grep -v "at::native" $f | cut -c1-170 | head -24
cp $f gpurun_out/${tag}_kernel_stats.csv
rm -rf gpurun_out/prof_$tag
