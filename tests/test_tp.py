"""Tensor-parallel verify (kangaroo_amd/tp.py).

CPU (world size 2, gloo): the partition maths — every rank's shard of a layer, its partial products, the
one-shot exchange (all_gather + sum in rank order) — against the unsplit layer and against the single-process
rehearsal (LocalGroup), bit for bit.
GPU: all ranks' shards of the remaining layers run one after the other on one device; the device's rank-order sum
is compared bit for bit with a NumPy rank-order sum of the very partials the kernels produced, the hidden rows
with the oracle (<= 1e-2 of max-abs, fp16 and bf16 alike) and with the unsplit kernels, and the whole decode
loop with the oracle's ids."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from kangaroo_amd.config import LlamaShape, get_shape
from kangaroo_amd.tp import LocalGroup, add_residual, ordered_sum, padded_heads, shard_layer
from kangaroo_amd.weights import Recipe, synth_state_dicts
from oracle import kangaroo_oracle as ko


def _layer_partials(sh, h, shape, rank=0):
    """fp32 torch restatement of what one shard's kernels compute up to the partial sums (no cache, no RoPE:
    partition maths only): attention output columns are taken as given (`h` stands in for them)."""
    w = sh["wo"].shape[1]
    a = h[:, rank * w:(rank + 1) * w] @ sh["wo"].float().T
    g = torch.nn.functional.silu(h @ sh["wg"].float().T) * (h @ sh["wu"].float().T)
    return a, g @ sh["wd"].float().T


def test_shards_reassemble_and_pad_heads():
    shape = LlamaShape(96, 2, 6, 6, 256, 50)           # 6 heads over 4 ranks: padded to 8, two of them zero
    base, _ = synth_state_dicts(shape, 1, seed=1, dtype=torch.float32)
    p = "model.layers.1."
    for tp in (1, 2, 4):
        sh = [shard_layer(base, p, shape, tp, r) for r in range(tp)]
        hp = padded_heads(6, tp)
        hd = shape.head_dim
        for name, key in (("wq", "self_attn.q_proj.weight"), ("wk", "self_attn.k_proj.weight"), ("wv", "self_attn.v_proj.weight")):
            full = torch.cat([s[name] for s in sh], 0)
            assert full.shape[0] == hp * hd
            assert torch.equal(full[:6 * hd], base[p + key]) and float(full[6 * hd:].abs().max() if hp > 6 else 0) == 0
        wo = torch.cat([s["wo"] for s in sh], 1)
        assert torch.equal(wo[:, :6 * hd], base[p + "self_attn.o_proj.weight"]) and (hp == 6 or float(wo[:, 6 * hd:].abs().max()) == 0)
        assert torch.equal(torch.cat([s["wg"] for s in sh], 0), base[p + "mlp.gate_proj.weight"])
        assert torch.equal(torch.cat([s["wd"] for s in sh], 1), base[p + "mlp.down_proj.weight"])
    with pytest.raises(ValueError):
        shard_layer(base, p, LlamaShape(96, 2, 6, 6, 250, 50), 4, 0)          # 250 / 4 is not whole
    with pytest.raises(ValueError):
        shard_layer(base, p, LlamaShape(96, 2, 6, 3, 256, 50), 2, 0)          # grouped-query heads that tp does not divide


_WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from kangaroo_amd.config import LlamaShape
from kangaroo_amd.tp import DistGroup, LocalGroup, add_residual, ordered_sum, shard_layer
from kangaroo_amd.weights import synth_state_dicts
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from test_tp import _layer_partials
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
shape = LlamaShape(128, 2, 4, 4, 256, 50)
base, _ = synth_state_dicts(shape, 1, seed=3, dtype=torch.float32)
h = torch.randn(5, 128, generator=torch.Generator().manual_seed(9))
p = "model.layers.1."
grp = DistGroup()
mine = shard_layer(base, p, shape, world, rank)
a, m = _layer_partials(mine, h, shape, rank)
got_a = ordered_sum(grp.all_partials([a]))
got_m = ordered_sum(grp.all_partials([m]))
# the single-process rehearsal of the same ranks
loc = LocalGroup(world)
parts = [_layer_partials(shard_layer(base, p, shape, world, r), h, shape, r) for r in range(world)]
ref_a = ordered_sum(loc.all_partials([x[0] for x in parts]))
ref_m = ordered_sum(loc.all_partials([x[1] for x in parts]))
full_a = h @ base[p + "self_attn.o_proj.weight"].T
full_m = (torch.nn.functional.silu(h @ base[p + "mlp.gate_proj.weight"].T) * (h @ base[p + "mlp.up_proj.weight"].T)) @ base[p + "mlp.down_proj.weight"].T
res = {"rank": rank, "bit_equal_local": bool(torch.equal(got_a, ref_a) and torch.equal(got_m, ref_m)),
       "err_a": float((got_a - full_a).abs().max() / full_a.abs().max()), "err_m": float((got_m - full_m).abs().max() / full_m.abs().max())}
# every rank holds the same bits
sig = torch.tensor([float(got_a.double().sum()), float(got_m.double().sum())], dtype=torch.float64)
all_sig = [torch.zeros_like(sig) for _ in range(world)]
dist.all_gather(all_sig, sig)
res["same_on_all_ranks"] = all(bool(torch.equal(s, all_sig[0])) for s in all_sig)
out = [None] * world
dist.all_gather_object(out, res)
if rank == 0:
    print(json.dumps(out))
dist.barrier()
dist.destroy_process_group()
"""


def test_partition_maths_two_ranks_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29621")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29621", str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert len(res) == 2
    for r in res:
        assert r["bit_equal_local"] and r["same_on_all_ranks"]
        assert r["err_a"] < 1e-4 and r["err_m"] < 1e-4          # fp32 sums in another order


# ---------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu


def _cut(shape, n):
    return type(shape)(**{**shape.to_dict(), "num_hidden_layers": n})


@gpu
@pytest.mark.parametrize("name,n_layers,tp,dtype,tol", [
    ("tp-small", 4, 2, torch.float16, 1e-2), ("tp-small", 4, 4, torch.float16, 1e-2),
    ("llama-33b", 4, 4, torch.bfloat16, 1e-2), ("llama-33b", 3, 8, torch.bfloat16, 1e-2),     # 52 heads over 8 ranks: 7 each, 4 zero heads
], ids=["small-tp2", "small-tp4", "33b-tp4-bf16", "33b-tp8-bf16"])
def test_tp_verify_pass_on_one_device(name, n_layers, tp, dtype, tol):
    from kangaroo_amd.kangaroo_model import KangarooModel
    from kangaroo_amd.tp import TPKangarooModel, tp_forward
    shape = LlamaShape(256, 4, 8, 8, 768, 1000) if name == "tp-small" else _cut(get_shape(name), n_layers)
    L = 2
    base, adapter = synth_state_dicts(shape, L, seed=71, dtype=dtype, recipe=Recipe.agreeable(0.3, 8.0), device="cuda")
    model = TPKangarooModel(shape, base, adapter, dtype, L, LocalGroup(tp))
    plain = KangarooModel.from_state_dicts(shape, base, adapter, dtype, L)
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), {k: v.cpu() for k, v in base.items()}, {k: v.cpu() for k, v in adapter.items()}, L)
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, shape.vocab_size, (1, 13), generator=g)
    # prefill through both stacks and the oracle: exit-layer rows, final rows, last-row logits
    o_tp = model.base_model(ids.cuda())
    o_pl = plain.base_model(ids.cuda())
    o_ref = om.base_model(ids)
    # Logits: the stated 1e-2 of the row's max-abs for both types.  The post-norm hidden rows are STORED 16-bit tensors:
    # in bf16 (8 significand bits) one spacing of the largest element is 2^-8 .. 2^-7 = 0.4 .. 0.8 % of it, so 1e-2 of
    # max-abs is under two spacings and the rows are held to four spacings instead (measured at this shape: 1.06e-2, i.e.
    # 1.7 spacings, against an oracle that keeps fp32 between the ops where the reference in bf16 rounds as these kernels do)
    hid_tol = tol if dtype == torch.float16 else 4 * 2.0 ** -7
    for a, b, bound in ((o_tp.hidden_states[-1], o_ref.hidden_states[-1], hid_tol), (o_tp.logits, o_ref.logits, tol), (o_pl.logits, o_ref.logits, tol)):
        err = float((a.float().cpu() - b).abs().max()) / float(b.abs().max())
        print(f"[tp {name} tp{tp}] error {err:.2e} of max-abs (bound {bound:.1e})")
        assert err <= bound
    assert torch.equal(o_tp.hidden_states[L], o_pl.hidden_states[L])                  # the replicated draft stack is the same code
    # one 7-row verify pass: the rank-order sum on the device against NumPy on the traced partials, bit for bit
    x = o_tp.hidden_states[L][0, :7].contiguous()
    trace = []
    for s in model.shards:
        s.k_cache.zero_(); s.v_cache.zero_()
    h = tp_forward(model.shards, LocalGroup(tp), x, 0, trace=trace)
    assert len(trace) == 2 * (shape.num_hidden_layers - L) and trace[0].shape == (tp, 7, shape.hidden_size)
    for parts in trace:
        p = parts.float().cpu().numpy()
        acc = p[0].copy()
        for r in range(1, tp):
            acc = acc + p[r]
        assert np.array_equal(ordered_sum(parts).cpu().numpy(), acc)
    h2 = tp_forward(model.shards, LocalGroup(tp), x, 0)
    assert torch.equal(h, h2)                                                            # deterministic
    # whole loop: the restated reference loop over the TP model against the oracle, through the round-walking comparison
    # of the single-GPU tests (a near-tie escape must be on the committed allow-list)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_model_gpu import check_against_reference
    tr = ko.Trace()
    ref = ko.kangaroo_forward(ids, om, 2, 24, EARLY_STOP_LAYER=L, trace=tr, keep_logits=True)
    trd = ko.Trace()
    out = ko.kangaroo_forward(ids.cuda(), model, 2, 24, EARLY_STOP_LAYER=L, trace=trd)
    check_against_reference(f"tp:host:{name}:tp{tp}", "plen13", (out[0][0], out[3], trd.draft_lengths, out[1], out[2]),
                            (ref[0][0], ref[3], tr.draft_lengths, ref[1], ref[2]), 13, tr.p_max, tr.draft_logits, torch.cat(tr.verify_logits, 0), 0.6,
                            lambda: om, tol=tol)
    # plain greedy decoding over the same TP kernels (no draft steps): the loop is lossless against it
    model.head_model.single_rows_in_verify_arithmetic = True
    van = ko.kangaroo_forward(ids.cuda(), model, 2, 24, EARLY_STOP_LAYER=L, SPECULATIVE_DECODING_STEPS=0)
    m = min(len(van[0][0]), len(out[0][0]))
    assert van[0][0][:m] == out[0][0][:m]
    del model, plain, om
    torch.cuda.empty_cache()


@gpu
def test_captured_tp_pass_equals_the_launch_by_launch_pass():
    """TPGraphs replays one captured pass per row count at changing positions (the first position is read on the
    device): hidden rows, K/V caches and the decode loop's tokens equal the launch-by-launch path bit for bit."""
    from kangaroo_amd.tp import TPKangarooModel
    shape = LlamaShape(256, 4, 8, 8, 768, 1000)
    L, tp, dtype = 2, 2, torch.float16
    base, adapter = synth_state_dicts(shape, L, seed=73, dtype=dtype, recipe=Recipe.agreeable(0.3, 8.0), device="cuda")
    mg = TPKangarooModel(shape, base, adapter, dtype, L, LocalGroup(tp), graphs=True)
    me = TPKangarooModel(shape, base, adapter, dtype, L, LocalGroup(tp), graphs=False)
    assert mg.base_model.graphs is not None and me.base_model.graphs is None
    g = torch.Generator().manual_seed(11)
    pos = 0
    for step in range(60):                             # every row count but the first use of each is a pure replay; the
        n = (16, 5, 1, 7)[step % 4]                    # context grows past two key chunks (128 keys each at this cache size),
        x = torch.randn(n, shape.hidden_size, generator=g).to(dtype).cuda()     # where attention blocks meet at counters
        hg = mg.base_model._deep(x, pos)
        he = me.base_model._deep(x, pos)
        assert torch.equal(hg, he), (step, n, pos)
        pos += n
    assert pos > 400 and sorted(mg.base_model.graphs.graphs) == [1, 5, 7, 16]
    for sg, se in zip(mg.shards, me.shards):
        assert torch.equal(sg.k_cache, se.k_cache) and torch.equal(sg.v_cache, se.v_cache)
    with pytest.raises(ValueError):
        mg.base_model._deep(torch.zeros(5, shape.hidden_size, dtype=dtype, device="cuda"), mg.shards[0].t_max - 2)
    ids = torch.randint(3, shape.vocab_size, (1, 21), generator=g)          # 21 rows: the prompt pass runs 16 + 5
    og = ko.kangaroo_forward(ids.cuda(), mg, 2, 40, EARLY_STOP_LAYER=L)
    oe = ko.kangaroo_forward(ids.cuda(), me, 2, 40, EARLY_STOP_LAYER=L)
    assert og[0] == oe[0] and og[3] == oe[3]


# ---------------------------------------------------------------------------------------------------
# the library-held shards (kg_session_set_tp): fused device-side loop, GEMM prefill, exchange kernel
# ---------------------------------------------------------------------------------------------------
def _r16(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).float().numpy()


def _rank_order_sum(p):
    acc = p[0].copy()
    for r in range(1, p.shape[0]):
        acc = acc + p[r]
    return acc


@gpu
@pytest.mark.parametrize("name,n_layers,tp,dtype,tol", [
    ("tp-small", 4, 2, torch.float16, 1e-2), ("tp-small", 4, 4, torch.float16, 1e-2),
    ("llama-33b", 4, 4, torch.bfloat16, 1e-2), ("llama-33b", 3, 8, torch.bfloat16, 1e-2),
], ids=["small-tp2", "small-tp4", "33b-tp4-bf16", "33b-tp8-bf16"])
def test_native_tp_session_on_one_device(name, n_layers, tp, dtype, tol):
    """All ranks' shards inside ONE session (TPPlan.local): the exchange kernel's result against a NumPy rank-order
    sum of the very partials the shards left in the exchange buffer, bit for bit; prefill rows and logits against the
    oracle; the fused device-side loop against the oracle through the round-walking comparison of the single-GPU
    tests (near-tie escapes must be on the committed allow-list), against the pass-by-pass loop over the same session,
    and against plain greedy decoding over the same sharded kernels (losslessness)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_model_gpu import check_against_reference
    from kangaroo_amd.inference import kangaroo_forward, kangaroo_forward_stepwise, vanilla_forward
    from kangaroo_amd.kangaroo_model import KangarooModel
    from kangaroo_amd.tp import NativeTPKangarooModel, TPPlan
    shape = LlamaShape(256, 4, 8, 8, 768, 1000) if name == "tp-small" else _cut(get_shape(name), n_layers)
    L = 2
    base, adapter = synth_state_dicts(shape, L, seed=71, dtype=dtype, recipe=Recipe.agreeable(0.3, 8.0), device="cuda")
    model = NativeTPKangarooModel(shape, base, adapter, dtype, L, TPPlan.local(tp))
    plain = KangarooModel.from_state_dicts(shape, base, adapter, dtype, L)
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), {k: v.cpu() for k, v in base.items()}, {k: v.cpu() for k, v in adapter.items()}, L)
    eng = model.engine
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(3, shape.vocab_size, (1, 13), generator=g)
    o_tp = model.base_model(ids.cuda())
    o_pl = plain.base_model(ids.cuda())
    o_ref = om.base_model(ids)
    hid_tol = tol if dtype == torch.float16 else 4 * 2.0 ** -7           # stored bf16 rows: see test_tp_verify_pass_on_one_device
    for a, b, bound in ((o_tp.hidden_states[-1], o_ref.hidden_states[-1], hid_tol), (o_tp.logits, o_ref.logits, tol), (o_pl.logits, o_ref.logits, tol)):
        err = float((a.float().cpu() - b).abs().max()) / float(b.abs().max())
        print(f"[native tp {name} tp{tp}] error {err:.2e} of max-abs (bound {bound:.1e})")
        assert err <= bound
    assert torch.equal(o_tp.hidden_states[L], o_pl.hidden_states[L])          # the replicated draft stack is the same code
    # one sharded layer on 7 rows: round16(h + round16(sum_r partial_r)) twice, from the partials in the exchange buffer
    x = o_tp.hidden_states[L][0, :7].contiguous()
    h = eng.forward_layers(L, L + 1, x, 0, 7)
    po = eng.tp_partials(0)[:, :7].cpu().numpy()
    pd = eng.tp_partials(1)[:, :7].cpu().numpy()
    assert float(np.abs(po).max()) > 0 and float(np.abs(pd).max()) > 0
    h_mid = _r16(x.float().cpu().numpy() + _r16(_rank_order_sum(po), dtype), dtype)
    h_out = _r16(h_mid + _r16(_rank_order_sum(pd), dtype), dtype)
    assert np.array_equal(h.float().cpu().numpy(), h_out)
    assert torch.equal(h, eng.forward_layers(L, L + 1, x, 0, 7))                # deterministic
    # the loops
    for plen, new in ((13, 24), (40, 16)):                                    # 40 rows: the prompt goes through the GEMM passes
        ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g)
        tr = ko.Trace()
        ref = ko.kangaroo_forward(ids, om, 2, new, EARLY_STOP_LAYER=L, trace=tr, keep_logits=True)
        out, new_tok, idx, accept, st = kangaroo_forward(ids.cuda(), model, 2, new, EARLY_STOP_LAYER=L, return_stats=True)
        check_against_reference(f"tp:{name}:tp{tp}", f"plen{plen}", (out[0], accept, st.draft_lengths, new_tok, idx),
                                (ref[0][0], ref[3], tr.draft_lengths, ref[1], ref[2]), plen, tr.p_max, tr.draft_logits, torch.cat(tr.verify_logits, 0), 0.6,
                                lambda: om, tol=tol)
        van = vanilla_forward(ids.cuda(), model, 2, new_tok)[0][0]
        assert van == out[0][:len(van)]
        if plen <= eng.get_tuning("prefill_rows"):
            # same prompt pass (decode kernels), same rounds: the pass-by-pass loop gives the same tokens
            sw = kangaroo_forward_stepwise(ids.cuda(), model, 2, new, EARLY_STOP_LAYER=L)
            assert sw[0][0] == out[0] and sw[3] == accept
    del model, plain, om
    torch.cuda.empty_cache()


_IPC_WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from kangaroo_amd.config import LlamaShape
from kangaroo_amd.inference import kangaroo_forward, vanilla_forward
from kangaroo_amd.tp import NativeTPKangarooModel, TPPlan
from kangaroo_amd.weights import Recipe, synth_state_dicts
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)                       # both ranks share the box's one GPU
shape = LlamaShape(256, 4, 8, 8, 768, 1000)
L, dtype = 2, torch.float16
base, adapter = synth_state_dicts(shape, L, seed=71, dtype=dtype, recipe=Recipe.agreeable(0.3, 8.0), device="cuda")
model = NativeTPKangarooModel(shape, base, adapter, dtype, L, TPPlan.distributed())
g = torch.Generator().manual_seed(5)
res = []
for plen, new in ((13, 24), (40, 16), (9, 30)):
    ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g)
    dist.barrier()
    out, new_tok, idx, accept = kangaroo_forward(ids.cuda(), model, 2, new, EARLY_STOP_LAYER=L)
    dist.barrier()
    van = vanilla_forward(ids.cuda(), model, 2, new_tok)[0][0]
    res.append({"out": out[0], "accept": accept, "van": van})
allr = [None] * world
dist.all_gather_object(allr, res)
if rank == 0:
    print(json.dumps(allr))
dist.barrier()
del model
dist.destroy_process_group()
"""


@gpu
def test_native_tp_two_processes_share_one_gpu_through_ipc(tmp_path):
    """One rank per PROCESS, as in a multi-GPU job, with both processes on the box's one GPU: each maps the other's
    exchange buffer through its inter-process handle, the exchange kernels push partials into the peer's buffer,
    raise its flags and wait for their own.  Both ranks must produce the tokens of the single-process rehearsal of
    the same two shards (which runs the same kernels on the same data, shard after shard), for the fused loop and
    for plain decoding."""
    from kangaroo_amd.inference import kangaroo_forward, vanilla_forward
    from kangaroo_amd.tp import NativeTPKangarooModel, TPPlan
    script = tmp_path / "w.py"
    script.write_text(_IPC_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29631", str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("[")][-1])
    assert len(res) == 2 and res[0] == res[1]                                  # the ranks agree on every token
    shape = LlamaShape(256, 4, 8, 8, 768, 1000)
    L, dtype = 2, torch.float16
    base, adapter = synth_state_dicts(shape, L, seed=71, dtype=dtype, recipe=Recipe.agreeable(0.3, 8.0), device="cuda")
    model = NativeTPKangarooModel(shape, base, adapter, dtype, L, TPPlan.local(2))
    g = torch.Generator().manual_seed(5)
    for (plen, new), r in zip(((13, 24), (40, 16), (9, 30)), res[0]):
        ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g)
        o, new_tok, idx, accept = kangaroo_forward(ids.cuda(), model, 2, new, EARLY_STOP_LAYER=L)
        assert o[0] == r["out"] and accept == r["accept"]
        assert r["van"] == r["out"][:len(r["van"])]


@gpu
def test_zz_tp_explained_cases_are_on_the_allow_list():
    """The near-tie escapes the tensor-parallel comparisons took are entries of the committed allow-list (a new one
    already failed where it occurred); entries of the list for these cases that no longer occur must be struck."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_model_gpu as M
    if os.environ.get("KG_EXPLAINED_UPDATE") or any("::" in a or a == "-k" for a in sys.argv):
        pytest.skip("partial run or list being regenerated")
    ran = {(e["case"], e["prompt"], e["kind"], e["position"]) for e in M._SEEN if e["case"].startswith("tp:")}
    listed = {k for k in M._ALLOW_KEYS if k[0].startswith("tp:")}
    assert ran == listed, f"not on the list: {sorted(ran - listed)}; listed but did not occur: {sorted(listed - ran)}"
