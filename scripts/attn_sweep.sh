#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace -d gpurun_out/attn_sweep -o sw --output-format csv -- python3 scripts/attn_sweep.py $1 > gpurun_out/attn_sweep.log 2>&1
python3 - <<'PY'
import csv
rows = [r for r in csv.DictReader(open("gpurun_out/attn_sweep/sw_kernel_trace.csv")) if "kg_attn" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000 for r in rows]
cfgs = [(rows_, L) for rows_ in (1, 7) for L in (64, 128, 256, 512, 1024, 1536, 2000)]
for i, c in enumerate(cfgs):
    seg = sorted(d[20 * i + 5:20 * i + 20])
    print(c, "median %.1f us  min %.1f" % (seg[len(seg) // 2], seg[0]))
PY
rm -rf gpurun_out/attn_sweep
