#!/bin/bash
# PMC passes over the gate/up projection of the verify pass (separate runs per counter group).
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$tag -o pmc -- python3 scripts/pmc_op.py gateup 7 32 > gpurun_out/pmc_$tag.log 2>&1
  f=$(ls gpurun_out/pmc_$tag/*counter_collection.csv 2>/dev/null || ls gpurun_out/pmc_$tag/*/*counter_collection.csv)
  head -1 $f > gpurun_out/pmc_${tag}_linear.csv
  grep "kg_linear_kernel" $f >> gpurun_out/pmc_${tag}_linear.csv || true
  rm -rf gpurun_out/pmc_$tag
  tail -1 gpurun_out/pmc_$tag.log
  wc -l gpurun_out/pmc_${tag}_linear.csv
done
