"""Persistent passes vs kernel-by-kernel passes on the same prompts: first differing token, per mode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kangaroo_amd.config import get_shape
from kangaroo_amd.inference import kangaroo_forward, vanilla_forward
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts
shape = get_shape("tinyllama-1.1b")
shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 4})
base, adapter = synth_state_dicts(shape, 2, seed=31, dtype=torch.float16, recipe=Recipe.agreeable(0.3, 8.0))
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2)
eng = model.engine
eng.set_tuning(persist=2)
for op in ("qkv", "o", "gateup", "down", "head"):
    eng.set_tuning(**{f"persist.{op}.waves": eng.get_tuning(f"mfma.{op}.waves")})
extra = [a.split("=") for a in sys.argv[1:]]
g = torch.Generator().manual_seed(5)
def first_diff(a, b):
    n = min(len(a), len(b))
    return next((i for i in range(n) if a[i] != b[i]), None)
for plen in (5, 33, 150):
    ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g).cuda()
    res = {}
    for name, kw in [("persist", dict(persist=2)), ("kernels", dict(persist=1))]:
        eng.set_tuning(**kw)
        for k, v in extra:
            eng.set_tuning(**{k: int(v)})
        out = kangaroo_forward(ids, model, -1, 96, return_stats=True)
        van = vanilla_forward(ids, model, -1, 96)[0][0]
        res[name] = (out[0][0], van, out[3], out[4].draft_lengths)
    ref = res["kernels"]
    for name, r in res.items():
        print(plen, name, "kang vs own vanilla:", first_diff(r[0], r[1]), "| kang vs kernels-kang:", first_diff(r[0], ref[0]), "| vanilla vs kernels-vanilla:", first_diff(r[1], ref[1]), "| accept same:", r[2] == ref[2], flush=True)
