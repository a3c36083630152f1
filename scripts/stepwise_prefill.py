import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from kangaroo_amd.config import get_shape
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts
shape = get_shape("vicuna-7b")
base, adapter = synth_state_dicts(shape, 2, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2, release_source=True)
eng = model.engine
g = torch.Generator().manual_seed(2)
for plen in (200, 640, 1500):
    ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g).cuda()
    for mode, key in (("whole-prompt pass", 129), ("16-row calls", 1)):
        eng.set_tuning(rows_prefill_min=key)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t = time.perf_counter()
            o = model.base_model(ids, output_hidden_states=True)
            a = model.adapter_model.forward_early_stop(inputs_embeds=o.hidden_states[2][:, :-1], use_cache=True)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
        print(f"mirror-class prefill + adapter prompt pass, {plen:5d} tokens, {mode:18s}: {best * 1e3:8.2f} ms", flush=True)
