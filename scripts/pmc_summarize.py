"""Per-launch HBM traffic of the gate/up kernel from the counter passes of scripts/pmc_run.sh: reads
gpurun_out/pmc_<counter>_linear.csv, applies the gfx950 correction (FETCH_SIZE tallies a wide coalesced streaming read at
half its bytes) and writes the JSON bench.py reads.  usage: python3 scripts/pmc_summarize.py <out.json> <commit>"""
import csv, json, sys

def mean(path, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and "kg_linear_kernel<0, 1, 1, 3>" in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)

fetch, n = mean("gpurun_out/pmc_FETCH_SIZE_linear.csv", "FETCH_SIZE")
write, _ = mean("gpurun_out/pmc_WRITE_SIZE_linear.csv", "WRITE_SIZE")
hit, _ = mean("gpurun_out/pmc_TCC_HIT_sum_TCC_MISS_sum_linear.csv", "TCC_HIT_sum")
miss, _ = mean("gpurun_out/pmc_TCC_HIT_sum_TCC_MISS_sum_linear.csv", "TCC_MISS_sum")
alg = 2 * 11008 * 4096 * 2
hbm = int(2 * fetch * 1024 + write * 1024)
json.dump({"kernel": "kg_linear_kernel<f16, MFMA, RMSNorm, SiLU*mul> (gate/up projection of the verify pass, 7 rows, 7B shape)",
           "command": "rocprofv3 --pmc <counter> --output-format csv -- python3 scripts/pmc_op.py gateup 7 32   (one run per counter group; scripts/pmc_run.sh)",
           "built_from_commit": sys.argv[2], "dispatches": n, "FETCH_SIZE_KB_mean": round(fetch, 2), "WRITE_SIZE_KB_mean": round(write, 2),
           "TCC_HIT_sum_mean": round(hit, 1), "TCC_MISS_sum_mean": round(miss, 1),
           "correction": "gfx950 FETCH_SIZE counts a wide coalesced streaming read at half its bytes: read bytes = 2 * FETCH_SIZE * 1024",
           "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round(hbm / alg, 4)}, open(sys.argv[1], "w"), indent=1)
print(open(sys.argv[1]).read())
