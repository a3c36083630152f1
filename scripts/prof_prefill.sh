#!/bin/bash
# Kernel trace of a long-prompt prefill (no decode rounds).  usage: prof_prefill.sh <plen> <tag>
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
plen=$1; tag=$2
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag --output-format csv -- python3 scripts/explore.py --recipes 0.08:10 --plen $plen --new 1 --modes kangaroo > gpurun_out/prof_$tag.log 2>&1
grep "T0=" gpurun_out/prof_$tag.log | cut -c1-160
f=$(ls gpurun_out/prof_$tag/*kernel_stats.csv 2>/dev/null || ls gpurun_out/prof_$tag/*/*kernel_stats.csv)
grep -v "at::native" $f | cut -c1-170 | head -14
