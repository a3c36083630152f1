"""Where a tensor-parallel decode spends its time on one GPU: prefill / decode-loop milliseconds of the sharded session
(all ranks rehearsed on this GPU) beside the unsplit session on the same prompts.
python scripts/tp_breakdown.py [model] [tp] [dtype]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape, DEFAULT_EXIT_LAYER
from kangaroo_amd.harness import spec_bench_prompts
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.tp import NativeTPKangarooModel, TPPlan
from kangaroo_amd.weights import Recipe, synth_state_dicts

name = sys.argv[1] if len(sys.argv) > 1 else "llama-33b"
tp = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dt = torch.bfloat16 if (len(sys.argv) <= 3 or sys.argv[3] == "bf16") else torch.float16
shape = get_shape(name)
L = DEFAULT_EXIT_LAYER.get(name, 2)
base, adapter = synth_state_dicts(shape, L, seed=0, dtype=dt, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
prompts = [p for p in spec_bench_prompts(shape.vocab_size, 24) if p.input_ids.shape[1] <= 256][:3]
for label, mk in (("tp%d-local" % tp, lambda: NativeTPKangarooModel(shape, base, adapter, dt, L, TPPlan.local(tp))),
                  ("unsplit", lambda: KangarooModel.from_state_dicts(shape, base, adapter, dt, L))):
    model = mk()
    eng = model.engine
    eng.generate(prompts[0].input_ids[0], 16, mode="kangaroo", eos_token_id=-1)
    eng.generate(prompts[0].input_ids[0], 16, mode="vanilla", eos_token_id=-1)
    for p in prompts:
        out, st = eng.generate(p.input_ids[0], 128, mode="kangaroo", eos_token_id=-1)
        vout, vst = eng.generate(p.input_ids[0], st.new_tokens, mode="vanilla", eos_token_id=-1)
        print(f"{label:10s} T0={p.input_ids.shape[1]:4d} new={st.new_tokens:4d} prefill {st.prefill_ms:7.2f} ms  decode {st.decode_ms:7.2f} ms  rounds {st.rounds:3d} "
              f"({st.decode_ms * 1e3 / st.rounds:7.1f} us/round)  {st.new_tokens / (st.prefill_ms + st.decode_ms) * 1e3:6.1f} tok/s | vanilla decode {vst.decode_ms:7.2f} ms "
              f"({vst.decode_ms * 1e3 / max(1, vst.new_tokens - 1):7.1f} us/token) identical {vout == out}", flush=True)
    del model, eng
    torch.cuda.empty_cache()
