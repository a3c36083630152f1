"""Decode-loop time per round against the share of every o_proj row group that idle attention blocks pull into L2."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts
name = sys.argv[1] if len(sys.argv) > 1 else "vicuna-7b"
shape = get_shape(name)
L = 3 if "13b" in name else 2
base, adapter = synth_state_dicts(shape, L, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L, release_source=True)
eng = model.engine
g = torch.Generator().manual_seed(1)
prompts = [torch.randint(3, shape.vocab_size, (n,), generator=g) for n in (64, 200, 300, 450, 600, 900, 1200)]
ref = None
cfgs = [(1, 6, 17), (6, 6, 17), (6, 2, 129), (6, 3, 129), (6, 3, 65), (6, 4, 65), (6, 4, 17), (4, 3, 129), (1, 6, 17)]
if len(sys.argv) > 2:
    cfgs = [tuple(int(x) for x in c.split(":")) for c in sys.argv[2].split(",")]
for pf, mn, dl in cfgs:
    eng.set_tuning(attn_prefetch=pf, attn_prefetch_min=mn, attn_prefetch_delay=dl)
    line = f"prefetch {pf - 1}/8 from {mn} chunks, delay {dl - 1} x 64 cycles: "
    outs = []
    for ids in prompts:
        eng.generate(ids, 16, mode="kangaroo")
        out, st = eng.generate(ids, 256, mode="kangaroo")
        outs.append(out)
        line += f" T0={ids.numel():4d} {st.decode_ms * 1e3 / st.rounds:7.1f} us/round"
    if ref is None:
        ref = outs
    print(line, "identical" if outs == ref else "DIFFERENT", flush=True)
