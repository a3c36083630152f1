#!/bin/bash
# SQ counters of the prefill GEMM kernels.  usage: pmc_prefill.sh "<counters>" <tag>
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $1 --output-format csv -d gpurun_out/pmcpf_$2 -o pmc -- python3 scripts/explore.py --recipes 0.08:10 --plen 128 --new 1 --modes kangaroo > gpurun_out/pmcpf_$2.log 2>&1
f=$(ls gpurun_out/pmcpf_$2/*counter_collection.csv 2>/dev/null || ls gpurun_out/pmcpf_$2/*/*counter_collection.csv)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "kg_gemm" in k or "kg_attn" in k:
        acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n", len(next(iter(d.values()))))
PY
rm -rf gpurun_out/pmcpf_$2
