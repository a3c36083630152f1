"""Where a decode's device time goes, from a rocprofv3 kernel trace (csv): per kernel name the launches that did work
and the ones that exited early (under 2 us), their summed durations, and the idle time in front of them (start minus
the previous kernel's end, on the same queue).  usage: trace_budget.py <kernel_trace.csv> [first_kernel_substring]"""
import csv, sys, collections

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the decode loop: everything after the last prefill GEMM
last_gemm = max((i for i, r in enumerate(rows) if "kg_gemm_kernel" in r[2]), default=-1)
rows = rows[last_gemm + 1:]
rows = [r for r in rows if r[2].startswith("void kg_") or r[2].startswith("kg_")]
agg = collections.OrderedDict()
prev_end = None
for s, e, n in rows:
    n = n.replace("void ", "").split("(")[0]
    d = (e - s) / 1e3
    gap = 0.0 if prev_end is None else max(0.0, (s - prev_end) / 1e3)
    if gap > 50: gap = 0.0        # host stalls between graph launches are not the kernels' doing
    prev_end = e
    a = agg.setdefault(n, [0, 0.0, 0.0, 0, 0.0, 0.0])
    if d < 2.0: a[3] += 1; a[4] += d; a[5] += gap
    else: a[0] += 1; a[1] += d; a[2] += gap
tot = sum(a[1] + a[2] + a[4] + a[5] for a in agg.values())
print(f"{'kernel':44s} {'work':>6s} {'us':>8s} {'gap':>6s} {'share':>6s} | {'early':>6s} {'us':>6s} {'gap':>6s} {'share':>6s}")
for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print(f"{n[:44]:44s} {a[0]:6d} {a[1] / max(a[0], 1):8.2f} {a[2] / max(a[0], 1):6.2f} {100 * (a[1] + a[2]) / tot:5.1f}% | {a[3]:6d} {a[4] / max(a[3], 1):6.2f} {a[5] / max(a[3], 1):6.2f} {100 * (a[4] + a[5]) / tot:5.1f}%")
print(f"total {tot / 1e3:.2f} ms over {len(rows)} launches; work {sum(a[1] for a in agg.values()) / tot:.3f}, gaps before work {sum(a[2] for a in agg.values()) / tot:.3f}, "
      f"early exits {sum(a[4] for a in agg.values()) / tot:.3f}, gaps before early exits {sum(a[5] for a in agg.values()) / tot:.3f}")
