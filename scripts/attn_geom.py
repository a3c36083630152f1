"""Decode-loop time per round against the attention launch geometry (blocks per head x teams), three context lengths."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts
shape = get_shape("vicuna-7b")
base, adapter = synth_state_dicts(shape, 2, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2, release_source=True)
eng = model.engine
g = torch.Generator().manual_seed(1)
prompts = [torch.randint(3, shape.vocab_size, (n,), generator=g) for n in (64, 300, 600, 1200)]
for blocks, teams in ((16, 1), (8, 1), (8, 2), (4, 1), (4, 2), (2, 2), (1, 2)):
    eng.set_tuning(attn_blocks=blocks, attn_teams=teams)
    line = f"blocks {blocks:2d} teams {teams}: "
    for ids in prompts:
        eng.generate(ids, 16, mode="kangaroo")
        out, st = eng.generate(ids, 256, mode="kangaroo")
        line += f" T0={ids.numel():4d} {st.decode_ms * 1e3 / st.rounds:7.1f} us/round"
    print(line, flush=True)
