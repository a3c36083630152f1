"""Sweep absolute block counts (balanced divisors of the row-group count among them) and waves per op."""
import os, sys, itertools
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts

model_name = sys.argv[1] if len(sys.argv) > 1 else "vicuna-7b"
ops = sys.argv[2].split(",") if len(sys.argv) > 2 else ["qkv", "o", "gateup", "down", "head"]
shape = get_shape(model_name)
base, adapter = synth_state_dicts(shape, 2, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2, release_source=True)
eng = model.engine
d, F, V = shape.hidden_size, shape.intermediate_size, shape.vocab_size
qkv = (shape.num_attention_heads + 2 * shape.num_key_value_heads) * shape.head_dim
G = {"qkv": qkv // 16, "o": d // 16, "gateup": 2 * F // 16, "down": d // 16, "head": (V + 15) // 16}
for rows in (1, 7):
    mode = "vec" if rows == 1 else "mfma"
    for op in ops:
        g = G[op]
        cands = sorted({256, 512} | {g // k for k in range(1, 17) if g % k == 0 and 128 <= g // k <= 2048} | {-(-g // k) for k in range(1, 12) if 128 <= -(-g // k) <= 2048})
        res = []
        for nb, w in itertools.product(cands, (2, 4, 8)):
            eng.set_tuning(**{f"{mode}.{op}.blocks": nb, f"{mode}.{op}.waves": w})
            try:
                us, nbytes = eng.bench_op(op, rows, iters=96)
            except Exception:
                continue
            res.append((us, nb, w))
        res.sort()
        print(f"{op:7s} rows {rows} G={g}: " + "  ".join(f"{us:5.2f}us(n{b},w{w})" for us, b, w in res[:7]), f"| best {nbytes / res[0][0] / 1e3:.0f} GB/s", flush=True)
