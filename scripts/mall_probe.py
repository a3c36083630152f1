"""Warm-cache probe: per-op time when kg_bench_op cycles through 1, 2, 4, 8 or all layers' weights.
One layer's o_proj (33.5 MB at 7B) fits the L2s and the memory-side cache, 8 layers' gate/up does not."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts

shape = get_shape("vicuna-7b")
base, adapter = synth_state_dicts(shape, 2, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2, release_source=True)
e = model.engine
for nl in (1, 2, 4, 8, 32):
    e.set_tuning(bench_layers=nl)
    row = []
    for op in ("qkv", "o", "gateup", "down"):
        for rows in (1, 7):
            us, nb = e.bench_op(op, rows, iters=128)
            row.append(f"{op}{rows}:{us:6.2f}us {nb/us/1e6:5.2f}TB/s")
    print(nl, " | ".join(row), flush=True)
