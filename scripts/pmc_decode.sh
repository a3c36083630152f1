#!/bin/bash
# Hardware counters per kernel over a short 7B decode (one rocprofv3 run per counter group).
# usage: pmc_decode.sh <out.txt> "<counters of pass 1>" "<counters of pass 2>" ...
# PMC_ARGS: further arguments of scripts/bench_ops.py (e.g. "--plen 1000 attn_prefetch=1")
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
out=$1
shift
: > "$out"
i=0
for c in "$@"; do
  i=$((i + 1))
  d=gpurun_out/pmc_pass$i
  timeout -s KILL 240 rocprofv3 --pmc $c -d $d -o p --output-format csv -- python3 scripts/bench_ops.py vicuna-7b --decode --short $PMC_ARGS > $d.log 2>&1 || { echo "pass '$c' failed" >> "$out"; tail -3 $d.log >> "$out"; exit 1; }
  f=$(ls $d/*counter_collection.csv 2>/dev/null || ls $d/*/*counter_collection.csv)
  echo "== pass: $c" >> "$out"
  python3 - "$f" >> "$out" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:64]
    if "at::native" in k or "rocclr" in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -max(n[k].values())):
    print(f"{k:64s} " + "  ".join(f"{c} mean {v / n[k][c]:14.1f} (x{n[k][c]})" for c, v in sorted(agg[k].items())))
PY
  rm -rf $d
done
