"""Spec-Bench-shaped evaluation through the harness (counterpart of evaluation/eval.py + speed.py):
N synthetic prompts in the six task groups, Kangaroo and plain greedy decoding on the same kernels,
per-group and overall tokens/s (mean over prompts of new_tokens / wall_time), speed-up = ratio of
means, mean accept length.  Writes one JSON report."""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape, DEFAULT_EXIT_LAYER
from kangaroo_amd.harness import run_eval, speed_report, spec_bench_prompts
from kangaroo_amd.inference import kangaroo_forward, vanilla_forward
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vicuna-7b")
ap.add_argument("--prompts", type=int, default=32)
ap.add_argument("--max-new-tokens", type=int, default=1024)
ap.add_argument("--out", default="gpurun_out/specbench.json")
a = ap.parse_args()
shape = get_shape(a.model)
L = DEFAULT_EXIT_LAYER.get(a.model, 2)
base, adapter = synth_state_dicts(shape, L, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L, release_source=True)
prompts = spec_bench_prompts(shape.vocab_size, a.prompts)
fk = lambda ids, m, eos, n, **kw: kangaroo_forward(ids, m, eos, n, EARLY_STOP_LAYER=L, **kw)
fv = lambda ids, m, eos, n, **kw: vanilla_forward(ids, m, eos, n)
def run_chunks(f, tag):
    out = []
    for i in range(0, len(prompts), 8):          # a progress line per 8 prompts (a silent run counts as hung)
        out += run_eval(model, f, prompts[i:i + 8], a.max_new_tokens, -1, warmup=1 if i == 0 else 0, keep_ids=True)
        print(f"{tag}: {len(out)}/{len(prompts)} prompts", flush=True)
    return out
recs = run_chunks(fk, "kangaroo")
base_recs = run_chunks(fv, "vanilla")
identical = sum(int(r.output_ids[:min(len(r.output_ids), len(b.output_ids))] == b.output_ids[:min(len(r.output_ids), len(b.output_ids))]) for r, b in zip(recs, base_recs))
report = {"model": a.model, "exit_layer": L, "prompts": len(prompts), "max_new_tokens": a.max_new_tokens,
          "identical_to_vanilla": f"{identical}/{len(prompts)}", "overall": speed_report(recs, base_recs), "groups": {}}
for cat in sorted({p.category for p in prompts}):
    rk = [r for r in recs if r.category == cat]
    rb = [r for r in base_recs if r.category == cat]
    rep = speed_report(rk, rb)
    rep["prompts"] = len(rk)
    rep["mean_prompt_len"] = sum(p.input_ids.shape[1] for p in prompts if p.category == cat) / max(1, len(rk))
    report["groups"][cat] = rep
os.makedirs(os.path.dirname(a.out), exist_ok=True)
with open(a.out, "w") as f:
    json.dump(report, f, indent=1)
print(json.dumps(report["overall"]))
for k, v in report["groups"].items():
    print(f"{k:15s} n={v['prompts']:2d} len={v['mean_prompt_len']:6.0f} tok/s={v['tokens_per_second']:6.1f} vanilla={v['tokens_per_second_baseline']:6.1f} x{v['speedup']:.3f} accept={v['mean_accept']:.2f}")
