"""One linear op of the 7B verify pass, launched back to back, for rocprofv3 --pmc passes.
usage: python3 scripts/pmc_op.py [op=gateup] [rows=7] [iters=32]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts

op = sys.argv[1] if len(sys.argv) > 1 else "gateup"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 7
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 32
shape = get_shape("vicuna-7b")
base, adapter = synth_state_dicts(shape, 2, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2, release_source=True)
us, nbytes = model.engine.bench_op(op, rows, iters=iters)
print(f"{op} rows {rows}: {us:.2f} us per launch, {nbytes} weight bytes per launch")
