"""Per-kernel times of the session's linear kernels (kg_bench_op, HIP events on the launch stream) and, optionally,
a short decode: python scripts/bench_ops.py [model] [--decode]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape, DEFAULT_EXIT_LAYER
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts

name = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "vicuna-7b"
shape = get_shape(name)
L = DEFAULT_EXIT_LAYER.get(name, 2)
base, adapter = synth_state_dicts(shape, L, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L, release_source=True)
eng = model.engine
for kv in [a for a in sys.argv[1:] if "=" in a]:
    k, v = kv.split("=")
    if "." in v:      # float keys
        import ctypes
        eng.lib.kg_session_setf(eng._handle, k.encode(), ctypes.c_float(float(v)))
    else:
        eng.set_tuning(**{k: int(v)})
for op in (() if "--short" in sys.argv else ("qkv", "o", "gateup", "down", "head")):
    r = []
    for rows in (1, 7):
        us, nb = eng.bench_op(op, rows, iters=2 * shape.num_hidden_layers)
        r.append(f"rows{rows} {us:6.2f} us {nb / us / 1e3:6.0f} GB/s")
    print(f"{op:7s} " + "   ".join(r), flush=True)
short = "--short" in sys.argv        # one prompt, few tokens, no per-kernel loop: for counter passes
if "--prefill" in sys.argv:
    g = torch.Generator().manual_seed(2)
    plens = tuple(int(x) for x in sys.argv[sys.argv.index("--plens") + 1].split(",")) if "--plens" in sys.argv else (128, 640, 1500)
    for plen in plens:
        ids = torch.randint(3, shape.vocab_size, (plen,), generator=g)
        eng.generate(ids, 2, mode="kangaroo")
        ms = [eng.generate(ids, 2, mode="kangaroo")[1].prefill_ms for _ in range(3)]
        print(f"prefill T0={plen:5d}: {min(ms):7.2f} ms  ({2 * shape.param_count() * plen / min(ms) / 1e9:6.0f} TFLOP/s)" if hasattr(shape, "param_count")
              else f"prefill T0={plen:5d}: {min(ms):7.2f} ms", flush=True)
if "--decode" in sys.argv:
    g = torch.Generator().manual_seed(1)
    plens = (int(sys.argv[sys.argv.index("--plen") + 1]),) if "--plen" in sys.argv else ((64,) if short else (64, 600, 1200))
    for plen in plens:
        ids = torch.randint(3, shape.vocab_size, (plen,), generator=g)
        eng.generate(ids, 16, mode="kangaroo")
        out, st = eng.generate(ids, 64 if short else 512, mode="kangaroo")
        nb = sum((d + 1) * L * shape.layer_weight_bytes() + d * (shape.attn_only_weight_bytes() + shape.head_weight_bytes())
                 + (shape.num_hidden_layers - L) * shape.layer_weight_bytes() + shape.head_weight_bytes() for d in st.draft_lengths)
        print(f"T0={plen:5d} new={st.new_tokens} decode={st.decode_ms:8.1f} ms  {(st.new_tokens - 1) / st.decode_ms * 1e3:7.1f} tok/s  accept {sum(st.accept_lengths) / len(st.accept_lengths):.2f} "
              f"draft {sum(st.draft_lengths) / len(st.draft_lengths):.2f}  us/round {st.decode_ms * 1e3 / st.rounds:7.1f}  loop {nb / st.decode_ms / 1e6:6.0f} GB/s = {nb / st.decode_ms / 1e6 / 8000:.4f}  ids {int(torch.as_tensor(out).long().sum())}", flush=True)
