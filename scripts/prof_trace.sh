#!/bin/bash
# Kernel trace of one decode and its time budget (scripts/trace_budget.py).  usage: prof_trace.sh <tag> <prompt length> [key=value tunings]
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
tag=${1:-tr}; plen=${2:-600}
shift; shift
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o $tag --output-format csv -- python3 scripts/bench_ops.py vicuna-7b --short --decode --plen $plen "$@" > gpurun_out/prof_$tag.log 2>&1 || exit 1
tail -2 gpurun_out/prof_$tag.log
f=$(ls gpurun_out/prof_$tag/*kernel_trace.csv 2>/dev/null || ls gpurun_out/prof_$tag/*/*kernel_trace.csv)
python3 scripts/trace_budget.py $f > gpurun_out/budget_$tag.txt
cat gpurun_out/budget_$tag.txt
rm -rf gpurun_out/prof_$tag
