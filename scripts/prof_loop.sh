#!/bin/bash
# Kernel trace of the decode loop alone (16-token prompt: one prefill pass).  usage: prof_loop.sh <mode> <tag>
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mode=$1; tag=$2
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python3 scripts/explore.py --recipes 0.08:10 --plen ${PLEN:-16} --new 256 --modes $mode > gpurun_out/prof_$tag.log 2>&1
tail -2 gpurun_out/prof_$tag.log
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv 2>/dev/null || ls gpurun_out/prof_$tag/*kernel_stats.csv)
grep -v "at::native" $f | cut -c1-200 | head -24
