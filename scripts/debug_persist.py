"""Compare one multi-row pass: persistent kernel vs kernel-by-kernel (GPU)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts

name = sys.argv[1] if len(sys.argv) > 1 else "small"
shape = get_shape(name)
if name == "tinyllama-1.1b":
    shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 4})
base, adapter = synth_state_dicts(shape, 2, seed=31, dtype=torch.float16, recipe=Recipe.agreeable(0.3, 8.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2)
eng = model.engine
N, d = shape.num_hidden_layers, shape.hidden_size
# the K split across waves fixes the summation order: compare like with like
for op in ("qkv", "o", "gateup", "down", "head"):
    eng.set_tuning(**{f"persist.{op}.waves": eng.get_tuning(f"mfma.{op}.waves")})
g = torch.Generator().manual_seed(1)
NL = int(sys.argv[2]) if len(sys.argv) > 2 else N
pos = 0
for it, rows in enumerate([7, 16, 3, 7, 2, 5, 7, 16, 16, 16, 7, 7, 7, 16, 16, 16, 16, 4]):
    x = (torch.randn(rows, d, generator=g) * 0.5).half().cuda()
    res = {}
    for mode in (1, 2):
        eng.set_tuning(persist=mode)
        res[mode] = eng.forward_layers(0, NL, x, pos, rows).float().cpu()
    diff = (res[1] - res[2]).abs().max().item()
    bad = (res[1] != res[2]).nonzero()
    print(f"it {it} pos {pos} rows {rows}: nan-off {int(res[1].isnan().sum())} nan-on {int(res[2].isnan().sum())} max|diff| {diff:.4g}  mismatching elements {len(bad)}" + (f" first {bad[0].tolist()} rows {sorted(set(bad[:,0].tolist()))} cols {sorted(set(bad[:,1].tolist()))[:12]}" if len(bad) else ""), flush=True)
    pos += rows
