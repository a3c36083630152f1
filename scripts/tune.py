"""Sweep launch geometry of the session's real kernels (real prologues / epilogues) with kg_bench_op."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts

model_name = sys.argv[1] if len(sys.argv) > 1 else "vicuna-7b"
shape = get_shape(model_name)
base, adapter = synth_state_dicts(shape, 2, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2, release_source=True)
eng = model.engine
for rows in (1, 7):
    mode = "vec" if rows == 1 else "mfma"
    for op in ("qkv", "o", "gateup", "down", "head"):
        key_b, key_w = f"{mode}.{op}.blocks_per_cu", f"{mode}.{op}.waves"
        res = []
        for bpc, w in itertools.product((1, 2, 3, 4, 6, 8), (2, 4, 8)):
            eng.set_tuning(**{key_b: bpc, key_w: w})
            try:
                us, nb = eng.bench_op(op, rows, iters=64)
            except Exception as e:
                continue
            res.append((us, bpc, w))
        res.sort()
        print(f"{op:7s} rows {rows}: " + "  ".join(f"{us:5.1f}us(b{b},w{w})" for us, b, w in res[:5]), f" | {nb / res[0][0] / 1e3:.0f} GB/s", flush=True)
