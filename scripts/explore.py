"""GPU exploration: synthetic-weight calibration and first-order throughput numbers.

usage: python scripts/explore.py [--model vicuna-7b] [--recipes ds:hs,ds:hs,...] [--new 256]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape, DEFAULT_EXIT_LAYER
from kangaroo_amd.harness import spec_bench_prompts
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vicuna-7b")
    ap.add_argument("--recipes", default="0.3:10")
    ap.add_argument("--new", type=int, default=256)
    ap.add_argument("--prompts", type=int, default=3)
    ap.add_argument("--tune", default="")
    ap.add_argument("--plen", type=int, default=0, help="use one random prompt of this many tokens instead of the Spec-Bench-shaped set")
    ap.add_argument("--modes", default="vanilla,kangaroo")
    a = ap.parse_args()
    shape = get_shape(a.model)
    L = DEFAULT_EXIT_LAYER.get(a.model, 2)
    prompts = spec_bench_prompts(shape.vocab_size, a.prompts)
    if a.plen:
        g = torch.Generator().manual_seed(1)
        prompts = [type("P", (), {"input_ids": torch.randint(3, shape.vocab_size, (1, a.plen), generator=g)})()]
    for rc in a.recipes.split(","):
        ds, hs = [float(x) for x in rc.split(":")]
        t0 = time.time()
        base, adapter = synth_state_dicts(shape, L, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(ds, hs), device="cuda")
        model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L, release_source=True)
        del base, adapter
        torch.cuda.synchronize()
        print(f"[{rc}] built in {time.time() - t0:.1f}s, mem {torch.cuda.memory_allocated() / 2**30:.1f} GiB", flush=True)
        if a.tune:
            model.engine.set_tuning(**{k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(","))})
        eng = model.engine
        wb = shape.layer_weight_bytes() * shape.num_hidden_layers + shape.head_weight_bytes()
        for p in prompts:
            ids = p.input_ids[0]
            van = None
            for mode in a.modes.split(","):
                eng.generate(ids, 16, mode=mode)     # warm-up + graph capture
                torch.cuda.synchronize()
                t0 = time.time()
                out, st = eng.generate(ids, a.new, mode=mode)
                dt = time.time() - t0
                line = f"  T0={ids.numel():5d} {mode:8s} new={st.new_tokens:4d} wall={dt * 1e3:8.1f}ms prefill={st.prefill_ms:7.1f}ms decode={st.decode_ms:8.1f}ms " \
                       f"tok/s={st.new_tokens / dt:7.1f} decode-only tok/s={(st.new_tokens - 1) / (st.decode_ms / 1e3):7.1f}"
                if mode == "vanilla":
                    line += f" weights-GB/s={(st.new_tokens - 1) * wb / (st.decode_ms / 1e3) / 1e9:7.0f}"
                    van = out
                else:
                    acc = st.accept_lengths
                    d = st.draft_lengths
                    line += f" rounds={st.rounds} mean_accept={sum(acc) / max(1, len(acc)):.2f} mean_draft={sum(d) / max(1, len(d)):.2f} lossless={van is not None and out == van[:len(out)]}"
                print(line, flush=True)
        del model, eng
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
