"""End-to-end parity (GPU): the HIP path through the C ABI against
 (a) the committed golden vectors written by the reference's own code, and
 (b) the CPU oracle run live on the same seeded weights at sizes it finishes in seconds.

Bar: output ids, accept lengths and per-round draft lengths bit-exact; logits within
1e-2 of the row's max-abs.  Where a sequence diverges, the divergence must be an argmax
near-tie: `explain_divergence` re-scores the device's sequence with the oracle
(teacher forcing) and requires the device's token to be within tolerance of the
oracle's best logit at the first differing position."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kangaroo_amd.config import LlamaShape, get_shape
from kangaroo_amd.inference import kangaroo_forward, kangaroo_forward_stepwise, vanilla_forward
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts
from oracle import kangaroo_oracle as ko

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "*.npz")) if not p.endswith("layers.npz") and not os.path.basename(p).startswith("train_"))
LOGIT_TOL = 1e-2
KG_STEPS_MAX = 7           # a round writes at most steps + 1 = 7 tokens


def load_case(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    base = {k[5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("base/")}
    adapter = {k[8:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("adapter/")}
    return z, meta, base, adapter


def oracle_model(shape_name, base, adapter, L, round_to=None):
    return ko.OracleKangarooModel(ko.Shape(**get_shape(shape_name).to_dict()), base, adapter, L, round_to)


# ---- near-tie escapes ------------------------------------------------------------------------------
# A 16-bit device path and an fp32 reference cannot agree bit for bit when two candidates are closer than the
# device's rounding noise.  Every place where a test lets such a difference pass is RECORDED (case, prompt, kind,
# position) and must be listed in tests/golden/EXPLAINED.json; a difference that is not on that list fails the
# test, and test_zz_explained_cases_equal_the_allow_list fails when the list names a case that did not occur.
# KG_EXPLAINED_UPDATE=1 skips the membership check so that a new list can be written from
# gpurun_out/EXPLAINED_run.json after a kernel change (every entry still has to pass its near-tie test).
EXPLAINED_PATH = os.path.join(GOLD, "EXPLAINED.json")
_ALLOW = json.load(open(EXPLAINED_PATH)) if os.path.exists(EXPLAINED_PATH) else []
_ALLOW_KEYS = {(e["case"], e["prompt"], e["kind"], e["position"]) for e in _ALLOW}
_SEEN = []


def record_explained(case, prompt, kind, position, detail):
    key = (case, str(prompt), kind, int(position))
    _SEEN.append({"case": key[0], "prompt": key[1], "kind": key[2], "position": key[3], "detail": detail})
    out_dir = os.path.join(os.path.dirname(os.path.dirname(__file__)), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "EXPLAINED_run.json"), "w") as f:
            json.dump(_SEEN, f, indent=1)
    except OSError:
        pass
    if not os.environ.get("KG_EXPLAINED_UPDATE"):
        assert key in _ALLOW_KEYS, f"near-tie escape not on the allow-list: {key}: {detail}"


def p_tol(p, scale, tol=None):
    """How far a softmax maximum p can move under the rounding noise of stored 16-bit logits.  The noise is
    taken as a fifth of the parity tolerance times the row's max-abs (2e-3 x max-abs for fp16: the kernel tests
    bound a stored logit's error by 4e-3 of max-abs, typical errors are a quarter of that; at max-abs 34 that
    is two fp16 spacings of the top logit), and d p / d l = p (1 - p) for the top logit, the opposite sign for
    the rest, hence the factor 2."""
    return 2.0 * p * (1.0 - p) * (0.2 * (tol or LOGIT_TOL)) * scale


def explain_divergence(dev_ids, ref_ids, T0, omodel, ref_is_oracle=True, tol=LOGIT_TOL):
    """First differing position must be a near-tie for the oracle under teacher forcing.  Returns
    (position, detail) or None when the sequences agree on their common length.  ref_is_oracle: ref_ids came
    from the fp32 reference / oracle, whose token there must then be the oracle's argmax."""
    n = min(len(dev_ids), len(ref_ids))
    first = next((i for i in range(n) if dev_ids[i] != ref_ids[i]), None)
    if first is None:
        return None
    assert first >= T0
    out = omodel.base_model(torch.tensor([dev_ids[:first]]))
    row = out.logits[0, -1]
    best = float(row.max())
    gap = best - float(row[dev_ids[first]])
    scale = float(row.abs().max())
    if ref_is_oracle:
        assert int(row.argmax()) == ref_ids[first], "the reference's own token is not the oracle's argmax on that prefix"
    else:
        assert best - float(row[ref_ids[first]]) <= tol * scale
    assert gap <= tol * scale, f"divergence at {first} is not a near-tie: gap {gap} vs scale {scale}"
    return first, f"oracle top-1 minus device token's logit {gap:.4g} of max-abs {scale:.4g}"


def explain_round_divergence(dev_ids, dev_accept, dev_draft, ref_ids, ref_accept, ref_draft, T0, ref_pmax, ref_draft_logits,
                             ref_verify_logits, thr, top_k=1, top_p=None, temperature=1.0, tol=LOGIT_TOL):
    """Rounds are walked in step; up to the first round whose (accept, draft, tokens) differ both loops were in
    the same state, so the reference's logits of THAT round are the fp32 values of what the device computed in
    16 bits.  The round must hold a decision that sits within rounding noise: a draft confidence next to the
    threshold, a draft or verify argmax with a top-2 gap inside the tolerance, or (non-greedy rules) a drafted
    token next to the rule's boundary.  Returns (position of the round's first token, detail)."""
    da, ra = list(dev_accept[1:]), list(ref_accept[1:])
    s, i_d, i_v = T0, 0, 0
    for r in range(min(len(da), len(ra))):
        q = int(ref_draft[r]) + 1
        if (da[r], dev_draft[r]) == (ra[r], ref_draft[r]) and dev_ids[s + 1:s + 1 + ra[r]] == ref_ids[s + 1:s + 1 + ra[r]]:
            s += ra[r]; i_d += q - 1; i_v += q
            continue
        for j in range(q - 1):
            row = torch.as_tensor(ref_draft_logits[i_d + j]).float()
            scale = float(row.abs().max())
            p = float(ref_pmax[i_d + j])
            if abs(p - thr) <= p_tol(p, scale, tol):
                return s, f"round {r} step {j}: p_max {p:.4f} vs threshold {thr} (tolerance {p_tol(p, scale, tol):.4f})"
            top2 = row.topk(2).values
            if float(top2[0] - top2[1]) <= tol * scale:
                return s, f"round {r} step {j}: draft top-2 gap {float(top2[0] - top2[1]):.4g} of {scale:.4g}"
        for i in range(q):
            row = torch.as_tensor(ref_verify_logits[i_v + i]).float()
            scale = float(row.abs().max())
            srt = row.sort(descending=True).values
            if float(srt[0] - srt[1]) <= tol * scale:
                return s, f"round {r} verify row {i}: top-2 gap {float(srt[0] - srt[1]):.4g} of {scale:.4g}"
            if top_k > 1 and float(srt[top_k - 1] - srt[top_k]) <= tol * scale:
                return s, f"round {r} verify row {i}: gap at the top-{top_k} boundary {float(srt[top_k - 1] - srt[top_k]):.4g} of {scale:.4g}"
            if top_p is not None:
                pr = torch.softmax(row / temperature, -1).sort(descending=True).values
                cum = pr.cumsum(-1)
                near = (cum - top_p).abs().min()
                if float(near) <= tol * scale / temperature:       # a probability mass moves by at most the logit error
                    return s, f"round {r} verify row {i}: cumulative mass {float(near):.4g} from top_p {top_p}"
        raise AssertionError(f"round {r} (start {s}): accept/draft {da[r]}/{dev_draft[r]} vs reference {ra[r]}/{ref_draft[r]} with no near-tie in that round")
    raise AssertionError("lists differ only in length")


def check_against_reference(case, prompt, dev, ref, T0, ref_pmax, ref_draft_logits, ref_verify_logits, thr, omodel_fn, rule=None, tol=LOGIT_TOL):
    """dev / ref = (ids, accept_list, draft_lengths, new_token, idx).  Bit-exact, or an allow-listed near-tie;
    whatever precedes the first difference is still compared exactly."""
    d_ids, d_acc, d_dr, d_new, d_idx = dev
    r_ids, r_acc, r_dr, r_new, r_idx = ref
    rule = rule or {}
    if d_ids == r_ids and d_acc == r_acc and d_dr == r_dr:
        assert (d_new, d_idx) == (r_new, r_idx)
        return
    # first round whose (accept, draft, tokens) differ: both loops entered it in the same state
    s0 = T0
    for a, c, b, d in zip(d_acc[1:], d_dr, r_acc[1:], r_dr):
        if (a, c) != (b, d) or d_ids[s0 + 1:s0 + 1 + b] != r_ids[s0 + 1:s0 + 1 + b]:
            break
        s0 += b
    assert d_ids[:s0 + 1] == r_ids[:s0 + 1]          # everything up to that round's start agrees exactly
    n = min(len(d_ids), len(r_ids))
    first = next((i for i in range(n) if d_ids[i] != r_ids[i]), None)
    if first is not None and not rule and first <= s0 + 1 + max(KG_STEPS_MAX, 0):
        # a token of that very round differs: under greedy acceptance both tokens are verify argmaxes of the same
        # prefix, so the teacher-forced oracle row decides whether it was a near-tie
        pos, why = explain_divergence(d_ids, r_ids, T0, omodel_fn(), tol=tol)
        record_explained(case, prompt, "ids", pos, why)
        return
    pos, why = explain_round_divergence(d_ids, d_acc, d_dr, r_ids, r_acc, r_dr, T0, ref_pmax, ref_draft_logits, ref_verify_logits, thr,
                                        top_k=rule.get("hyper_k", 1) if rule.get("do_sample") == "top_k" else 1,
                                        top_p=rule.get("hyper_p") if rule.get("do_sample") == "top_p" else None,
                                        temperature=rule.get("temperature", 1.0), tol=tol)
    assert pos == s0
    record_explained(case, prompt, "round", pos, why)
    if first is not None and not rule:
        # the rounds parted on a near-tie first (same tokens, other boundaries) and a token differs later on
        pos, why = explain_divergence(d_ids, r_ids, T0, omodel_fn(), tol=tol)
        record_explained(case, prompt, "ids", pos, why)


def _golden_ref(z, pi):
    return (z[f"p{pi}/output_ids"].tolist(), z[f"p{pi}/accept_length_list"].tolist(), z[f"p{pi}/draft_lengths"].tolist(),
            int(z[f"p{pi}/new_token"]), int(z[f"p{pi}/idx"]))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_fused_loop_matches_reference_golden(path):
    z, meta, base, adapter = load_case(path)
    case = os.path.basename(path)[:-4]
    shape = get_shape(meta["shape"])
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, meta["exit_layer"])
    top_k = meta.get("accept_top_k", 1)          # > 1: top-k verification (inference_kangaroo_MTBench.py:377-389)
    rule = dict(do_sample="top_k", hyper_k=top_k) if top_k > 1 else {}
    if meta.get("accept_top_p"):                 # top-p verification (:356-374)
        rule = dict(do_sample="top_p", hyper_p=meta["accept_top_p"], temperature=meta["temperature"])
    if meta.get("accept_typical"):               # typical verification (:293-314)
        rule = dict(do_sample="typical", temperature=meta["temperature"], epsilon=meta["accept_typical"][0], delta=meta["accept_typical"][1])
    for pi in range(meta["n_prompts"]):
        ids = torch.from_numpy(z[f"p{pi}/input_ids"]).cuda()
        out, new_tok, idx, accept, st = kangaroo_forward(ids, model, meta["eos"], meta["max_new_tokens"], EARLY_STOP_LAYER=meta["exit_layer"],
                                                         SPECULATIVE_DECODING_STEPS=meta["steps"], threshold=meta["threshold"], return_stats=True, **rule)
        # verify logits are stored per round as one [q, V] block per row
        check_against_reference(case, f"p{pi}", (out[0], accept, st.draft_lengths, new_tok, idx), _golden_ref(z, pi), ids.shape[1],
                                z[f"p{pi}/p_max"], z[f"p{pi}/draft_logits"], z[f"p{pi}/verify_logits"], meta["threshold"],
                                lambda: oracle_model(meta["shape"], base, adapter, meta["exit_layer"]), rule)
        if not rule:
            # lossless on the device whatever the comparison with the reference said: plain greedy decoding on
            # the same kernels gives the same ids
            van = vanilla_forward(ids, model, meta["eos"], new_tok)[0][0]
            assert van == out[0][:len(van)]
            assert len(van) == len(out[0]) or van[-1] == meta["eos"]


@pytest.mark.parametrize("path", CASES[:3], ids=[os.path.basename(p)[:-4] for p in CASES[:3]])
def test_stepwise_interface_matches_golden(path):
    """The reference-shaped methods (forward_draft_or_large_model / forward_early_stop /
    head_model with view-based caches) driven pass by pass."""
    z, meta, base, adapter = load_case(path)
    shape = get_shape(meta["shape"])
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, meta["exit_layer"])
    for pi in range(meta["n_prompts"]):
        ids = torch.from_numpy(z[f"p{pi}/input_ids"]).cuda()
        tr = {}
        out, new_tok, idx, accept = kangaroo_forward_stepwise(ids, model, meta["eos"], meta["max_new_tokens"], EARLY_STOP_LAYER=meta["exit_layer"],
                                                              SPECULATIVE_DECODING_STEPS=meta["steps"], threshold=meta["threshold"], trace=tr)
        check_against_reference("stepwise:" + os.path.basename(path)[:-4], f"p{pi}", (out[0], accept, tr["draft_lengths"], new_tok, idx), _golden_ref(z, pi),
                                ids.shape[1], z[f"p{pi}/p_max"], z[f"p{pi}/draft_logits"], z[f"p{pi}/verify_logits"], meta["threshold"],
                                lambda: oracle_model(meta["shape"], base, adapter, meta["exit_layer"]))
        if (out[0], accept, tr["draft_lengths"]) == _golden_ref(z, pi)[:3]:
            ref_p = z[f"p{pi}/p_max"]
            scales = np.abs(z[f"p{pi}/draft_logits"]).max(axis=-1) if len(ref_p) else np.zeros(0)
            tol = np.maximum(1e-3, 2 * ref_p * (1 - ref_p) * 0.5 * LOGIT_TOL * scales)
            assert np.all(np.abs(np.asarray(tr.get("p_max", [])) - ref_p) <= tol)


def test_logits_against_golden():
    """Prefill + draft + verify logits of the device path vs the reference's."""
    z, meta, base, adapter = load_case(CASES[1])
    shape = get_shape(meta["shape"])
    L = meta["exit_layer"]
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L)
    om = oracle_model(meta["shape"], base, adapter, L)
    ids = torch.from_numpy(z["p0/input_ids"])
    o_ref = om.base_model(ids)
    o_dev = model.base_model(ids.cuda())
    scale = float(o_ref.logits.abs().max())
    assert float((o_dev.logits.float().cpu() - o_ref.logits).abs().max()) <= LOGIT_TOL * scale
    h_ref, h_dev = o_ref.hidden_states[L], o_dev.hidden_states[L].float().cpu()
    assert float((h_dev - h_ref).abs().max()) <= LOGIT_TOL * float(h_ref.abs().max())
    # adapter prefill + one incremental step
    a_ref, kv_ref = om.adapter_model.forward_early_stop(inputs_embeds=o_ref.hidden_states[L])
    a_dev, kv_dev = model.adapter_model.forward_early_stop(inputs_embeds=o_dev.hidden_states[L], use_cache=True)
    assert float((a_dev.float().cpu() - a_ref).abs().max()) <= LOGIT_TOL * float(a_ref.abs().max())
    assert kv_dev[0][0].shape == kv_ref[0][0].shape
    assert float((kv_dev[0][0].float().cpu() - kv_ref[0][0]).abs().max()) <= LOGIT_TOL * float(kv_ref[0][0].abs().max())


@pytest.mark.parametrize("shape_name,L,recipe,seed", [
    ("small", 2, Recipe.agreeable(0.3, 8.0), 11),
    ("small", 3, Recipe(), 12),
    ("tinyllama-1.1b", 2, Recipe.agreeable(0.2, 10.0), 13),
    ("tinyllama-1.1b-full", 2, Recipe.agreeable(0.2, 10.0), 14),
])
def test_against_live_oracle(shape_name, L, recipe, seed):
    """Larger shapes than the fixtures: oracle (fp32 CPU) and device on the same weights."""
    shape = get_shape(shape_name.replace("-full", ""))
    if shape_name == "tinyllama-1.1b":
        shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 4})     # same layer shape, 4 layers: seconds on the CPU
    # "-full": all 22 layers, the reference's own CPU-runnable configuration (about half a minute of fp32 CPU work)
    base, adapter = synth_state_dicts(shape, L, seed=seed, dtype=torch.float16, recipe=recipe)
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L)
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), base, adapter, L)
    g = torch.Generator().manual_seed(seed)
    for plen in (9, 40, 150):                       # 150: the prompt goes through the layers as a whole (one product per projection)
        ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g)
        tr = ko.Trace()
        ref = ko.kangaroo_forward(ids, om, 2, 48, EARLY_STOP_LAYER=L, trace=tr, keep_logits=True)
        out, new_tok, idx, accept, st = kangaroo_forward(ids.cuda(), model, 2, 48, EARLY_STOP_LAYER=L, return_stats=True)
        check_against_reference(f"live:{shape_name}:L{L}:seed{seed}", f"plen{plen}", (out[0], accept, st.draft_lengths, new_tok, idx),
                                (ref[0][0], ref[3], tr.draft_lengths, ref[1], ref[2]), plen, tr.p_max, tr.draft_logits, torch.cat(tr.verify_logits, 0), 0.6,
                                lambda: ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), base, adapter, L))
        van = vanilla_forward(ids.cuda(), model, 2, new_tok)[0][0]
        assert van == out[0][:len(van)]


@pytest.mark.parametrize("rule", [dict(do_sample="top_k", hyper_k=2), dict(do_sample="top_p", hyper_p=0.8, temperature=0.7),
                                  dict(do_sample="top_p", hyper_p=0.3, temperature=1.5)],
                         ids=["top_k2", "top_p80_t70", "top_p30_t150"])
def test_non_greedy_rules_against_live_oracle(rule):
    """Top-k / top-p verification (inference_kangaroo_MTBench.py:356-389) at the full vocabulary size
    (TinyLlama layer shape, 4 layers) against the oracle on the same weights.  These rules are not
    lossless; ids, accept lengths and draft lengths must match the oracle's exactly (seeds were kept
    whose decisions sit clear of 16-bit noise)."""
    shape = get_shape("tinyllama-1.1b")
    shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 4})
    base, adapter = synth_state_dicts(shape, 2, seed=31, dtype=torch.float16, recipe=Recipe.agreeable(0.5, 8.0))
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2)
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), base, adapter, 2)
    okw = dict(accept_top_k=rule["hyper_k"]) if rule["do_sample"] == "top_k" else dict(accept_top_p=rule["hyper_p"], temperature=rule["temperature"])
    ids = torch.randint(3, shape.vocab_size, (1, 12), generator=torch.Generator().manual_seed(31))
    tr = ko.Trace()
    ref = ko.kangaroo_forward(ids, om, 2, 40, EARLY_STOP_LAYER=2, trace=tr, keep_logits=True, **okw)
    out, new_tok, idx, accept, st = kangaroo_forward(ids.cuda(), model, 2, 40, EARLY_STOP_LAYER=2, return_stats=True, **rule)
    check_against_reference(f"live-rule:{rule['do_sample']}", "seed31", (out[0], accept, st.draft_lengths, new_tok, idx),
                            (ref[0][0], ref[3], tr.draft_lengths, ref[1], ref[2]), 12, tr.p_max, tr.draft_logits, torch.cat(tr.verify_logits, 0), 0.6,
                            None, rule)
    # the rule was applied, and is more permissive than greedy matching
    greedy = kangaroo_forward(ids.cuda(), model, 2, 40, EARLY_STOP_LAYER=2)
    assert sum(accept) / len(accept) >= sum(greedy[3]) / len(greedy[3])


def test_top_p_one_keeps_every_draft():
    """hyper_p = 1: every token's predecessors weigh <= 1, all nuclei are the whole vocabulary, every
    drafted token survives and each round accepts draft length + 1 tokens."""
    shape = get_shape("small")
    base, adapter = synth_state_dicts(shape, 2, seed=33, dtype=torch.float16, recipe=Recipe.agreeable(0.3, 8.0))
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2)
    ids = torch.randint(3, shape.vocab_size, (1, 10), generator=torch.Generator().manual_seed(33)).cuda()
    out, new_tok, idx, accept, st = kangaroo_forward(ids, model, -1, 48, do_sample="top_p", hyper_p=1.0, temperature=1.0, return_stats=True)
    assert accept[1:] == [d + 1 for d in st.draft_lengths]
    # temperature 0 selects the greedy rule (:282-290)
    a = kangaroo_forward(ids, model, -1, 48, do_sample="top_p", hyper_p=0.9, temperature=0.0)
    b = kangaroo_forward(ids, model, -1, 48)
    assert a == b


def test_eos_stop_max_length_and_shortest_prompt_against_live_oracle():
    """Loop exits of inference_kangaroo.py: an eos among the verified tokens ends the call with the eos
    as last id (:97-100,123-124); max_length caps the budget (:43); a one-token prompt."""
    z, meta, base, adapter = load_case(CASES[1])
    L = meta["exit_layer"]
    shape = get_shape(meta["shape"])
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L)
    om = lambda: oracle_model(meta["shape"], base, adapter, L)
    ids = torch.from_numpy(z["p0/input_ids"])
    T0 = ids.shape[1]
    gold = z["p0/output_ids"].tolist()
    checked = 0
    # every generated token takes a turn as eos: accepted-draft, correction and bonus positions all occur
    for k in range(T0, min(len(gold), T0 + 24)):
        eos = gold[k]
        ref = ko.kangaroo_forward(ids, om(), eos, 64, EARLY_STOP_LAYER=L)
        out = kangaroo_forward(ids.cuda(), model, eos, 64, EARLY_STOP_LAYER=L)
        assert ref[0][0][-1] == eos or len(ref[0][0]) > k + 1      # the reference stops only on a verify-side eos
        assert out == ref, f"eos {eos} (position {k})"
        checked += int(ref[0][0][-1] == eos)
    assert checked >= 12
    for max_length in (T0 + 9, T0 + 16, 48):
        ref = ko.kangaroo_forward(ids, om(), -1, 64, max_length=max_length, EARLY_STOP_LAYER=L)
        out = kangaroo_forward(ids.cuda(), model, -1, 64, max_length=max_length, EARLY_STOP_LAYER=L)
        assert out == ref and len(out[0][0]) <= max_length
    one = torch.tensor([[gold[3]]])
    tr = ko.Trace()
    ref = ko.kangaroo_forward(one, om(), -1, 24, EARLY_STOP_LAYER=L, trace=tr, keep_logits=True)
    out = kangaroo_forward(one.cuda(), model, -1, 24, EARLY_STOP_LAYER=L, return_stats=True)
    check_against_reference("one-token-prompt", "p0", (out[0][0], out[3], out[4].draft_lengths, out[1], out[2]),
                            (ref[0][0], ref[3], tr.draft_lengths, ref[1], ref[2]), 1, tr.p_max, tr.draft_logits, torch.cat(tr.verify_logits, 0), 0.6, om)


def test_constructor_from_checkpoint_dirs_and_eval_harness(tmp_path):
    """KangarooModel(base_dir, adapter_dir, args, EARLY_STOP_LAYER) reads the on-disk formats of
    kangaroo/kangaroo_model.py:22-44 (sharded HF checkpoint + adapter directory); run_eval writes the
    answer-file fields of evaluation/eval.py:232-233 and speed_report applies evaluation/speed.py's formulas."""
    import types
    from kangaroo_amd.harness import run_eval, speed_report, spec_bench_prompts
    from kangaroo_amd.weights import save_synthetic_checkpoint
    z, meta, base, adapter = load_case(CASES[1])
    L = meta["exit_layer"]
    shape = get_shape(meta["shape"])
    save_synthetic_checkpoint(str(tmp_path / "base"), str(tmp_path / "adapter"), shape, base, adapter, n_shards=2)
    model = KangarooModel(str(tmp_path / "base"), str(tmp_path / "adapter"), types.SimpleNamespace(dtype="float16"), EARLY_STOP_LAYER=L)
    ids = torch.from_numpy(z["p0/input_ids"]).cuda()
    out = kangaroo_forward(ids, model, meta["eos"], meta["max_new_tokens"], EARLY_STOP_LAYER=L, SPECULATIVE_DECODING_STEPS=meta["steps"], threshold=meta["threshold"])
    assert out[0][0] == z["p0/output_ids"].tolist()
    with pytest.raises(ValueError):
        KangarooModel(str(tmp_path / "base"), str(tmp_path / "adapter"), types.SimpleNamespace(dtype="float32"), EARLY_STOP_LAYER=L)
    # harness: kangaroo and vanilla over the same prompts
    prompts = [p for p in spec_bench_prompts(shape.vocab_size, 12) if p.input_ids.shape[1] <= 200][:4]
    fk = lambda ids, m, eos, n, **kw: kangaroo_forward(ids, m, eos, n, EARLY_STOP_LAYER=L, **kw)
    fv = lambda ids, m, eos, n, **kw: vanilla_forward(ids, m, eos, n)
    af = tmp_path / "answers.jsonl"
    recs = run_eval(model, fk, prompts, 48, -1, answer_file=str(af), keep_ids=True)
    base_recs = run_eval(model, fv, prompts, 48, -1, keep_ids=True)
    for r, b in zip(recs, base_recs):
        n = min(len(r.output_ids), len(b.output_ids))
        assert r.output_ids[:n] == b.output_ids[:n]
        assert sum(r.accept_lengths) == r.new_tokens and r.idxs == len(r.accept_lengths) - 1
    rep = speed_report(recs, base_recs)
    assert rep["speedup"] > 0 and 1.0 <= rep["mean_accept"] <= meta["steps"] + 1
    lines = [json.loads(l) for l in open(af)]
    assert len(lines) == len(prompts) and set(lines[0]["choices"][0]) >= {"new_tokens", "wall_time", "idxs", "accept_lengths"}


def test_vanilla_gemv_mode_agrees_up_to_near_ties():
    """Plain decoding with every layer in the 1-row GEMV kernels (tuning key vanilla_gemv = 2): same model,
    other summation order in the remaining layers, so the ids equal the standard comparator's or part on
    an argmax near-tie."""
    shape = get_shape("small")
    base, adapter = synth_state_dicts(shape, 2, seed=41, dtype=torch.float16, recipe=Recipe.agreeable(0.3, 8.0))
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2)
    ids = torch.randint(3, shape.vocab_size, (1, 14), generator=torch.Generator().manual_seed(41)).cuda()
    std = vanilla_forward(ids, model, -1, 64)[0][0]
    model.engine.set_tuning(vanilla_gemv=2)
    assert model.engine.get_tuning("vanilla_gemv") == 1
    gemv = vanilla_forward(ids, model, -1, 64)[0][0]
    model.engine.set_tuning(vanilla_gemv=1)
    assert len(gemv) == len(std)
    if gemv != std:
        pos, why = explain_divergence(gemv, std, 14, oracle_model("small", base, adapter, 2), ref_is_oracle=False)
        record_explained("vanilla-gemv-vs-verify-arithmetic", "seed41", "ids", pos, why)
    assert vanilla_forward(ids, model, -1, 64)[0][0] == std


class _StubTokenizer:
    """Hugging Face call surface over a toy vocabulary: a character maps to 3 + ord % (V - 3)."""
    eos_token_id = 2
    special_tokens_map = {"eos_token": "</s>"}

    def __init__(self, vocab):
        self.vocab = vocab

    def __call__(self, texts, return_tensors="pt"):
        import types
        ids = torch.tensor([[3 + ord(ch) % (self.vocab - 3) for ch in texts[0][-40:]]])
        return types.SimpleNamespace(input_ids=ids, to=lambda dev: types.SimpleNamespace(input_ids=ids.to(dev)))

    def decode(self, ids, spaces_between_special_tokens=False):
        return " ".join(str(int(i)) for i in ids)


def test_question_file_eval_loop(tmp_path):
    """Question JSONL in, answer JSONL out with the reference's fields (evaluation/eval.py:221-246), two turns
    per question, Kangaroo and plain decoding giving the same answers."""
    from kangaroo_amd.eval import run_eval as run_questions
    z, meta, base, adapter = load_case(CASES[1])
    L = meta["exit_layer"]
    shape = get_shape(meta["shape"])
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L)
    tok = _StubTokenizer(shape.vocab_size)
    qf = tmp_path / "question.jsonl"
    qf.write_text("".join(json.dumps({"question_id": 80 + i, "category": "mt_bench", "turns": [f"first {i}", f"second {i}"]}) + "\n" for i in range(3)))
    outs = {}
    for name, fwd in (("kangaroo", lambda inp, m, t, n, **kw: kangaroo_forward(inp, m, t, n, EARLY_STOP_LAYER=L, **kw)),
                      ("vanilla", lambda inp, m, t, n, **kw: vanilla_forward(inp, m, t, n))):
        af = tmp_path / f"{name}.jsonl"
        mean_accept = run_questions(model, tok, fwd, name, str(qf), None, None, str(af), 24)
        rows = [json.loads(l) for l in open(af)]
        assert [r["question_id"] for r in rows] == [80, 81, 82] and all(r["model_id"] == name for r in rows)
        c = rows[0]["choices"][0]
        assert set(c) == {"index", "turns", "idxs", "new_tokens", "wall_time", "accept_lengths"}
        assert len(c["turns"]) == 2 and len(c["new_tokens"]) == 2 and sum(c["accept_lengths"]) == sum(c["new_tokens"])
        assert mean_accept >= 1.0
        outs[name] = [r["choices"][0]["turns"] for r in rows]
    for a, b in zip(outs["kangaroo"], outs["vanilla"]):
        # first turns see the same prompt: same ids up to the shorter, budget-bounded answer (the second
        # turn's prompt contains the first answer, whose length differs between the two loops)
        ta, tb = a[0].split(), b[0].split()
        n = min(len(ta), len(tb))
        assert n >= 24 - 1 - 6 and ta[:n] == tb[:n]


def test_bf16_runs_and_is_lossless():
    shape = get_shape("small")
    base, adapter = synth_state_dicts(shape, 2, seed=21, dtype=torch.bfloat16, recipe=Recipe.agreeable(0.3, 8.0))
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.bfloat16, 2)
    ids = torch.randint(3, shape.vocab_size, (1, 17), generator=torch.Generator().manual_seed(3)).cuda()
    out, new_tok, idx, accept = kangaroo_forward(ids, model, 2, 64)
    van = vanilla_forward(ids, model, 2, new_tok)[0][0]
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), base, adapter, 2)
    if van != out[0][:len(van)]:
        pos, why = explain_divergence(out[0], van, 17, om, ref_is_oracle=False)
        record_explained("bf16-kangaroo-vs-vanilla", "seed3", "ids", pos, why)
    assert max(accept) <= 7 and min(accept) >= 1 and sum(accept[1:]) == new_tok - 1


def test_error_conventions_and_bounds():
    z, meta, base, adapter = load_case(CASES[0])
    shape = get_shape(meta["shape"])
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2)
    with pytest.raises(AssertionError):
        model.base_model.forward_draft_or_large_model(in_tokens_small=torch.tensor([[3]]).cuda())
    model.base_model.past_key_values = []
    with pytest.raises(ValueError):
        model.base_model.forward_draft_or_large_model()
    with pytest.raises(ValueError):
        model.base_model.forward_draft_or_large_model(in_tokens_small=torch.tensor([[3]]).cuda(), in_features_large=torch.zeros(1, 1, 64).cuda())
    with pytest.raises(NotImplementedError):
        model.forward()
    # the loop stops 1 + steps short of the budget (inference_kangaroo.py:46)
    ids = torch.from_numpy(z["p0/input_ids"]).cuda()
    out, new_tok, idx, accept = kangaroo_forward(ids, model, -1, 40)
    assert new_tok <= 40 and new_tok >= 40 - 1 - 6
    out1, new1, _, acc1 = kangaroo_forward(ids, model, -1, 1)            # no round fits: only the prefill token
    assert new1 == 1 and acc1 == [1] and len(out1[0]) == ids.shape[1] + 1


@pytest.mark.parametrize("shape_name,dtype", [("small", torch.float16), ("tinyllama-1.1b", torch.float16), ("small", torch.bfloat16)])
def test_persistent_pass_is_bit_identical_to_kernel_by_kernel(shape_name, dtype):
    """The verify pass / deep layers can run as one persistent kernel (tuning key "persist" = 2,
    shapes with head_dim >= 32) instead of one kernel per op.  Same arithmetic, same summation
    order: tokens, accept lengths and draft lengths must agree exactly."""
    shape = get_shape(shape_name)
    if shape_name == "tinyllama-1.1b":
        shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 4})
    base, adapter = synth_state_dicts(shape, 2, seed=31, dtype=dtype, recipe=Recipe.agreeable(0.3, 8.0))
    model = KangarooModel.from_state_dicts(shape, base, adapter, dtype, 2)
    eng = model.engine
    eng.set_tuning(persist=2)
    assert eng.get_tuning("persist") == 1, "persistent passes should be available for this shape"
    # the K split across a block's waves fixes the summation order: compare like with like
    for op in ("qkv", "o", "gateup", "down", "head"):
        eng.set_tuning(**{f"persist.{op}.waves": eng.get_tuning(f"mfma.{op}.waves")})
    g = torch.Generator().manual_seed(5)
    for plen in (5, 33, 150):
        ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g).cuda()
        eng.set_tuning(persist=2)
        on = kangaroo_forward(ids, model, -1, 96, return_stats=True)
        van_on = vanilla_forward(ids, model, -1, 40)[0][0]
        eng.set_tuning(persist=1)
        assert eng.get_tuning("persist") == 0
        off = kangaroo_forward(ids, model, -1, 96, return_stats=True)
        van_off = vanilla_forward(ids, model, -1, 40)[0][0]
        assert on[0][0] == off[0][0] and on[3] == off[3] and on[4].draft_lengths == off[4].draft_lengths
        assert van_on == van_off
        assert van_on == on[0][0][:len(van_on)]
    eng.set_tuning(persist=2)


@pytest.mark.parametrize("shape_name,dtype", [("small", torch.float16), ("tinyllama-1.1b", torch.float16), ("vicuna-7b", torch.float16),
                                              ("vicuna-13b", torch.float16), ("llama-33b", torch.bfloat16), ("gqa-hd128", torch.float16)],
                         ids=["small", "tinyllama-1.1b", "vicuna-7b", "vicuna-13b", "llama-33b-bf16", "gqa-hd128"])
def test_gemm_prefill_matches_chunked_prefill(shape_name, dtype):
    """Prompts longer than 16 tokens are prefilled in one 128-row GEMM pass, from 129 tokens on layer by layer over the
    whole prompt (one matrix product per projection, csrc/mm.hip; the d-column projections of a short prompt in K
    ranges added by a row kernel; at head_dim 128 the attention that shares key tiles between four waves); switched off,
    in 16-row passes of the decode kernels.  Same math, different K summation order: caches and the first token's
    hidden state agree to 16-bit rounding noise.  The real layer shapes (d 4096 / 5120 / 6656, 32 / 40 / 52 heads) run
    with three layers."""
    # grouped query heads at head_dim 128 (8 query heads on 2 key / value heads): no configuration of the reference has both
    shape = LlamaShape(1024, 3, 8, 2, 2816, 4000) if shape_name == "gqa-hd128" else get_shape(shape_name)
    if shape_name == "tinyllama-1.1b":
        shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 4})
    if shape_name in ("vicuna-7b", "vicuna-13b", "llama-33b"):
        shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 3})
    base, adapter = synth_state_dicts(shape, 2, seed=41, dtype=dtype, recipe=Recipe.agreeable(0.3, 8.0))
    model = KangarooModel.from_state_dicts(shape, base, adapter, dtype, 2)
    eng = model.engine
    assert eng.get_tuning("gemm_prefill") == 1
    g = torch.Generator().manual_seed(9)
    assert eng.get_tuning("rows_prefill_min") == 129           # from here on the prompt goes through the layers in whole-prompt passes
    # a stored bf16 value is one of 256 per binade: two summation orders can leave the largest element one spacing (2^-8 .. 2^-7
    # of it) apart, and a value next to a rounding boundary two; fp16 has 8 times finer spacing
    tol = 1e-2 if dtype == torch.float16 else 2 * 2.0 ** -7
    worst = 0.0
    for plen in (17, 64, 65, 129, 200, 257, 333, 511):          # 257: one row past two 128-row tiles; 511: the last row block short by one
        ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g).cuda()
        caches = {}
        for mode in (2, 1):
            eng.set_tuning(gemm_prefill=mode, rows_prefill_min=129 if mode == 2 else 1)
            eng.k_cache.zero_(); eng.v_cache.zero_(); eng.ak_cache.zero_(); eng.av_cache.zero_()
            out, st = eng.generate(ids[0], 1, mode="kangaroo")
            torch.cuda.synchronize()
            caches[mode] = (eng.k_cache.float().cpu().clone(), eng.v_cache.float().cpu().clone(),
                            eng.ak_cache.float().cpu().clone(), eng.av_cache.float().cpu().clone(), out)
        for a, b in zip(caches[2][:4], caches[1][:4]):
            assert float(b.abs().max()) > 0
            err = float((a - b).abs().max()) / float(b.abs().max())
            worst = max(worst, err)
            assert err <= tol, (plen, err)
    print(f"[prefill {shape_name}] caches of the two prompt passes differ by at most {worst:.2e} of max-abs (bound {tol:.1e})")
    eng.set_tuning(gemm_prefill=2, rows_prefill_min=129)
    del model
    torch.cuda.empty_cache()


def test_full_size_7b_losslessness_and_determinism():
    """At the benchmark's size (Vicuna-7B shape, exit layer 2, 6 draft steps, threshold 0.6) the fp32 CPU
    oracle takes minutes per prompt; beside the short oracle comparison of
    test_7b_shape_parity_with_the_oracle_is_a_near_tie_or_exact, the size-independent properties are checked:
    Kangaroo's output equals plain greedy decoding on the same kernels token for token, accept lengths add up
    to the new tokens, every round accepts between 1 and steps+1 tokens and never more than it drafted + 1,
    and a second run reproduces the first bit for bit."""
    _losslessness_and_determinism("vicuna-7b", 2, torch.float16, ((23, 160), (300, 96)))


@pytest.mark.parametrize("path", [CASES[1], CASES[2], next(p for p in CASES if "L3" in p)], ids=lambda p: os.path.basename(p)[:-4])
def test_reference_loop_drives_the_mirror_classes(path):
    """The reference's own loop, statement for statement (oracle.kangaroo_forward restates
    evaluation/inference_kangaroo.py:15-129), pointed at the HIP classes instead of the oracle's model: device
    token buffer and device position_ids on every call (:20-21,29,57,60,63,77,86,89), prefill logits for ALL rows
    and hidden_states[-1] / [EARLY_STOP_LAYER] (:30-35), `[:, -1:]` slices, `.float()` on the head output (:79,91),
    cache views read through `.shape[2]` and truncated with `[:, :, :n, :]` (:53,55,87,108-119)."""
    z, meta, base, adapter = load_case(path)
    case = "refloop:" + os.path.basename(path)[:-4]
    shape = get_shape(meta["shape"])
    L = meta["exit_layer"]
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L)
    for pi in range(meta["n_prompts"]):
        ids = torch.from_numpy(z[f"p{pi}/input_ids"]).cuda()
        tr = ko.Trace()
        out, new_tok, idx, accept = ko.kangaroo_forward(ids, model, meta["eos"], meta["max_new_tokens"], EARLY_STOP_LAYER=L,
                                                        SPECULATIVE_DECODING_STEPS=meta["steps"], threshold=meta["threshold"], trace=tr)
        check_against_reference(case, f"p{pi}", (out[0], accept, tr.draft_lengths, new_tok, idx), _golden_ref(z, pi), ids.shape[1],
                                z[f"p{pi}/p_max"], z[f"p{pi}/draft_logits"], z[f"p{pi}/verify_logits"], meta["threshold"],
                                lambda: oracle_model(meta["shape"], base, adapter, L))
        # same answer as the fused loop on the same kernels
        fused = kangaroo_forward(ids, model, meta["eos"], meta["max_new_tokens"], EARLY_STOP_LAYER=L, SPECULATIVE_DECODING_STEPS=meta["steps"], threshold=meta["threshold"])
        assert fused[0][0] == out[0]
    # position ids that do not continue the cache are refused, on the device as on the host
    model.base_model.past_key_values = list(model.base_model(ids, position_ids=torch.arange(ids.shape[1], device="cuda").view(1, -1)).past_key_values)
    T = ids.shape[1]
    with pytest.raises(ValueError):
        model.base_model.forward_draft_or_large_model(in_tokens_small=ids[:, -1:], position_ids=torch.tensor([[T + 1]], device="cuda"))
    with pytest.raises(ValueError):
        model.base_model.forward_draft_or_large_model(in_tokens_small=ids[:, -1:], position_ids=torch.tensor([[T - 1]]))
    with pytest.raises(ValueError):
        model.adapter_model.forward_early_stop(inputs_embeds=torch.zeros(1, 2, shape.hidden_size, device="cuda", dtype=torch.float16),
                                               position_ids=torch.tensor([[3, 5]], device="cuda"))
    with pytest.raises(ValueError):
        model.base_model(torch.tensor([[3, shape.vocab_size]]).cuda())          # token id beyond the embedding table


@pytest.mark.parametrize("shape_name", ["small", "tinyllama-1.1b"])
def test_mirror_classes_prefill_a_long_prompt_as_a_whole(shape_name):
    """`model.base_model(input_ids, output_hidden_states=True)` and the adapter's prompt pass of the reference's loop
    (evaluation/inference_kangaroo.py:30-38) on a 200-token prompt: one pass over the whole prompt (kg_prefill_rows,
    kg_adapter_forward_rows) instead of thirteen 16-row calls.  Logits of ALL rows and both hidden-state tensors against the
    fp32 oracle, the caches against the ones the fused loop's prefill leaves (the same pass: equal bits), and the reference's
    loop driven over the classes against the oracle's loop."""
    shape = get_shape(shape_name)
    if shape_name == "tinyllama-1.1b":
        shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 4})
    L = 2
    base, adapter = synth_state_dicts(shape, L, seed=91, dtype=torch.float16, recipe=Recipe.agreeable(0.3, 8.0))
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L)
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), base, adapter, L)
    eng = model.engine
    assert eng.whole_prompt_rows() == 129
    g = torch.Generator().manual_seed(91)
    ids = torch.randint(3, shape.vocab_size, (1, 200), generator=g)
    dev = model.base_model(ids.cuda(), output_hidden_states=True)
    ref = om.base_model(ids)
    assert dev.logits.shape == (1, 200, shape.vocab_size)
    for name, a, b in (("logits", dev.logits, ref.logits), ("hidden_states[L]", dev.hidden_states[L], ref.hidden_states[L]),
                       ("hidden_states[-1]", dev.hidden_states[-1], ref.hidden_states[-1])):
        err = float((a.float().cpu() - b).abs().max()) / float(b.abs().max())
        print(f"[whole-prompt prefill {shape_name}] {name}: {err:.2e} of max-abs")
        assert err <= LOGIT_TOL, name
    assert dev.past_key_values[0][0].shape[2] == 200
    k_step, v_step = eng.k_cache.clone(), eng.v_cache.clone()
    # the adapter's prompt pass (rows 0 .. 198 of hidden_states[L]) against the oracle's adapter and against 16-row passes
    x = dev.hidden_states[L][:, :-1]
    a_dev, _ = model.adapter_model.forward_early_stop(inputs_embeds=x, position_ids=torch.arange(199, device="cuda").view(1, -1), use_cache=True)
    ak_rows = eng.ak_cache.clone()
    a_ref, _ = om.adapter_model.forward_early_stop(inputs_embeds=x.float().cpu(), position_ids=torch.arange(199).view(1, -1), use_cache=True)
    assert float((a_dev.float().cpu() - a_ref).abs().max()) <= LOGIT_TOL * float(a_ref.abs().max())
    eng.set_tuning(rows_prefill_min=1)                   # off: 16-row passes of the decode kernels
    a_16, _ = model.adapter_model.forward_early_stop(inputs_embeds=x, position_ids=torch.arange(199, device="cuda").view(1, -1), use_cache=True)
    eng.set_tuning(rows_prefill_min=129)
    assert float((a_dev.float() - a_16.float()).abs().max()) <= LOGIT_TOL * float(a_16.float().abs().max())
    assert float((ak_rows.float() - eng.ak_cache.float()).abs().max()) <= LOGIT_TOL * float(ak_rows.float().abs().max())
    # the fused loop's prefill is the same pass over the same rows: the same cache bits
    eng.generate(ids[0], 1, mode="kangaroo")
    assert torch.equal(eng.k_cache[..., :200, :], k_step[..., :200, :])
    if eng.get_tuning("v_transposed"):
        assert torch.equal(eng.v_cache[..., :200], v_step[..., :200])
    else:
        assert torch.equal(eng.v_cache[..., :200, :], v_step[..., :200, :])
    # the reference's loop over the classes
    tr = ko.Trace()
    rf = ko.kangaroo_forward(ids, om, 2, 40, EARLY_STOP_LAYER=L, trace=tr, keep_logits=True)
    trd = ko.Trace()
    out = ko.kangaroo_forward(ids.cuda(), model, 2, 40, EARLY_STOP_LAYER=L, trace=trd)
    check_against_reference(f"refloop-long:{shape_name}", "plen200", (out[0][0], out[3], trd.draft_lengths, out[1], out[2]),
                            (rf[0][0], rf[3], tr.draft_lengths, rf[1], rf[2]), 200, tr.p_max, tr.draft_logits, torch.cat(tr.verify_logits, 0), 0.6,
                            lambda: om)


def test_full_depth_error_against_fp32_and_against_the_reference_arithmetic_in_16_bit():
    """The 1e-2 bar against the fp32 oracle is checked at the real layer shapes with the layer count cut.  At the real DEPTH
    (32 layers, narrower so that the CPU oracle takes seconds) fp16 still meets it; bf16 cannot, whoever computes it: the
    oracle run with every stored tensor rounded to bf16 -- the reference's own arithmetic in that type, op boundary by op
    boundary -- sits 1.5 .. 1.9e-2 of max-abs from its fp32 self.  The device is held to that: no further from fp32 than
    1.5 x the reference arithmetic's own distance (measured on MI355X: see the printed lines)."""
    shape = LlamaShape(1024, 32, 8, 8, 2816, 32000)
    L = 2
    g = torch.Generator().manual_seed(7)
    ids = torch.randint(3, shape.vocab_size, (1, 32), generator=g)
    for dtype in (torch.float16, torch.bfloat16):
        base, adapter = synth_state_dicts(shape, L, seed=0, dtype=dtype, recipe=Recipe.agreeable(0.08, 10.0))
        model = KangarooModel.from_state_dicts(shape, base, adapter, dtype, L)
        ref = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), base, adapter, L).base_model(ids).logits[0]
        rnd = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), base, adapter, L, round_to=dtype).base_model(ids).logits[0]
        dev = model.base_model(ids.cuda()).logits[0].float().cpu()
        scale = ref.abs().amax(-1)
        e_dev = float(((dev - ref).abs().amax(-1) / scale).max())
        e_ref = float(((rnd - ref).abs().amax(-1) / scale).max())
        e_dr = float(((dev - rnd).abs().amax(-1) / scale).max())
        print(f"[depth 32, {str(dtype).split('.')[-1]}] logits, max over 32 rows, of the row's max-abs: device vs fp32 oracle {e_dev:.2e}; "
              f"oracle with {str(dtype).split('.')[-1]} op boundaries vs fp32 oracle {e_ref:.2e}; device vs that oracle {e_dr:.2e}")
        if dtype == torch.float16:
            assert e_dev <= LOGIT_TOL
        assert e_dev <= 1.5 * e_ref
        del model
        torch.cuda.empty_cache()


def _cut(shape, n_layers):
    return type(shape)(**{**shape.to_dict(), "num_hidden_layers": n_layers})


@pytest.mark.parametrize("name,L,n_layers,dtype,tol,recipe", [
    ("vicuna-13b", 3, 5, torch.float16, LOGIT_TOL, (0.3, 10.0)),       # data/vicuna_13B_config.json:8-21, exit layer 3 (data/ge_data_all_vicuna.py:149-152)
    ("llama-33b", 2, 4, torch.bfloat16, LOGIT_TOL, (0.3, 10.0)),       # d 6656 / F 17920 / 52 heads in bf16: the 2-wave o_proj split and bf16 MFMA at real size
    ("vicuna-7b", 2, 4, torch.float16, LOGIT_TOL, (0.3, 10.0)),
], ids=["13b-L3-5layers-f16", "33b-4layers-bf16", "7b-4layers-f16"])
def test_real_layer_shapes_against_live_oracle(name, L, n_layers, dtype, tol, recipe):
    """The layer shapes of the benchmark configurations with the layer count cut so that the fp32 CPU oracle
    finishes in seconds: every kernel runs at its real K / N (two-chunk X prologue at d > 4096, the large-K
    down_proj mode, the o_proj wave splits).  The bf16 case is held to the same 1e-2 of the row's max-abs as the
    fp16 ones (measured on MI355X: 7.0e-3 and 7.9e-3 in bf16 at the 33B shape, 1.0e-3 .. 1.3e-3 in fp16); the measured
    error of every case is printed (pytest -s)."""
    shape = _cut(get_shape(name), n_layers)
    base, adapter = synth_state_dicts(shape, L, seed=61, dtype=dtype, recipe=Recipe.agreeable(*recipe), device="cuda")
    model = KangarooModel.from_state_dicts(shape, base, adapter, dtype, L)
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), {k: v.cpu() for k, v in base.items()}, {k: v.cpu() for k, v in adapter.items()}, L)
    del base, adapter
    g = torch.Generator().manual_seed(61)
    for plen in (11, 37, 160):                      # 160: the whole-prompt pass (matrix products of csrc/mm.hip, K ranges, shared-key attention)
        ids = torch.randint(3, shape.vocab_size, (1, plen), generator=g)
        tr = ko.Trace()
        ref = ko.kangaroo_forward(ids, om, 2, 40, EARLY_STOP_LAYER=L, trace=tr, keep_logits=True)
        out, new_tok, idx, accept, st = kangaroo_forward(ids.cuda(), model, 2, 40, EARLY_STOP_LAYER=L, return_stats=True)
        check_against_reference(f"shape:{name}:{n_layers}", f"plen{plen}", (out[0], accept, st.draft_lengths, new_tok, idx),
                                (ref[0][0], ref[3], tr.draft_lengths, ref[1], ref[2]), plen, tr.p_max, tr.draft_logits, torch.cat(tr.verify_logits, 0), 0.6,
                                lambda: om, tol=tol)
        # logits of the prompt's last row: device vs oracle, within the stated tolerance of the row's max-abs
        dev_row = model.base_model(ids.cuda(), logits_rows="last").logits[0, -1].float().cpu()
        ref_row = om.base_model(ids).logits[0, -1]
        err = float((dev_row - ref_row).abs().max()) / float(ref_row.abs().max())
        print(f"[logits] {name} {n_layers} layers {str(dtype).split('.')[-1]} plen {plen}: last-row logits max-abs error {err:.2e} of the row's max-abs (bound {tol:.0e})")
        assert err <= tol
        van = vanilla_forward(ids.cuda(), model, 2, new_tok)[0][0]
        assert van == out[0][:len(van)]
    del model, om
    torch.cuda.empty_cache()


def _losslessness_and_determinism(name, L, dtype, prompts):
    shape = get_shape(name)
    base, adapter = synth_state_dicts(shape, L, seed=0, dtype=dtype, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
    model = KangarooModel.from_state_dicts(shape, base, adapter, dtype, L, release_source=True)
    del base, adapter
    eng = model.engine
    g = torch.Generator().manual_seed(77)
    for plen, new in prompts:
        ids = torch.randint(3, shape.vocab_size, (plen,), generator=g)
        out, st = eng.generate(ids, new, mode="kangaroo")
        out2, st2 = eng.generate(ids, new, mode="kangaroo")
        van, _ = eng.generate(ids, st.new_tokens, mode="vanilla")
        assert out == out2 and st.accept_lengths == st2.accept_lengths and st.draft_lengths == st2.draft_lengths
        assert out == van
        assert out[:plen] == ids.tolist()
        assert 1 + sum(st.accept_lengths) == st.new_tokens
        assert len(st.accept_lengths) == len(st.draft_lengths) == st.rounds
        assert all(1 <= a <= 7 and a <= d + 1 and 1 <= d <= 6 for a, d in zip(st.accept_lengths, st.draft_lengths))
        assert new - 1 - 6 <= st.new_tokens <= new
    del model, eng
    torch.cuda.empty_cache()


def test_full_size_13b_losslessness_and_determinism():
    """Vicuna-13B shape, all 40 layers, exit layer 3 (the third benchmark configuration): the size-independent
    properties of test_full_size_7b_losslessness_and_determinism."""
    _losslessness_and_determinism("vicuna-13b", 3, torch.float16, ((23, 120), (300, 72)))


def test_7b_shape_parity_with_the_oracle_is_a_near_tie_or_exact():
    """The benchmark's own model (Vicuna-7B shape, 32 layers, bench.py's weight recipe) against the fp32 CPU
    oracle on the same weights: 2 prompts x 24 new tokens.  Whatever differs first (a token, or an accept /
    draft length) is traced by oracle/parity.py and has to be a near-tie within 1e-2 of the row's max-abs,
    with the device's logits row for the same prefix within that tolerance of the oracle's."""
    from oracle import parity
    shape = get_shape("vicuna-7b")
    L = 2
    base, adapter = synth_state_dicts(shape, L, seed=0, dtype=torch.float16, recipe=Recipe.agreeable(0.08, 10.0), device="cuda")
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, L)
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), {k: v.cpu() for k, v in base.items()}, {k: v.cpu() for k, v in adapter.items()}, L)
    del base, adapter
    for seed, plen in ((7, 32), (8, 20)):
        ids = torch.randint(3, shape.vocab_size, (1, plen), generator=torch.Generator().manual_seed(seed))
        tr = ko.Trace()
        ref = ko.kangaroo_forward(ids, om, -1, 24, EARLY_STOP_LAYER=L, trace=tr, keep_logits=True)
        out, new_tok, idx, accept, st = kangaroo_forward(ids.cuda(), model, -1, 24, EARLY_STOP_LAYER=L, return_stats=True)
        rec = parity.first_difference(out[0], accept, st.draft_lengths, ref[0][0], ref[3], tr.draft_lengths, plen, om, tr.p_max, tr.draft_logits, 0.6,
                                      dev_logits_row=lambda pre: model.base_model(torch.tensor([pre]).cuda(), logits_rows="last").logits[0, -1])
        assert rec["kind"] == "none" or rec["within_tol"], rec
        if rec["kind"] != "none":
            record_explained("7b-bench-weights", f"seed{seed}", rec["kind"], rec.get("position", rec.get("round", -1)), json.dumps(rec))
    del model, om
    torch.cuda.empty_cache()



def test_attention_geometry_does_not_change_results():
    """Blocks per head and teams per block only decide which block/team works on which key chunk;
    the chunk partials and their merge order are fixed, so outputs are bit-identical."""
    shape = get_shape("tinyllama-1.1b")
    shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": 4})
    base, adapter = synth_state_dicts(shape, 2, seed=51, dtype=torch.float16, recipe=Recipe.agreeable(0.3, 8.0))
    model = KangarooModel.from_state_dicts(shape, base, adapter, torch.float16, 2)
    eng = model.engine
    g = torch.Generator().manual_seed(6)
    ids = torch.randint(3, shape.vocab_size, (700,), generator=g)
    ref = None
    for blocks, teams in ((16, 1), (2, 2), (4, 2), (2, 1), (1, 2)):
        eng.set_tuning(attn_blocks=blocks, attn_teams=teams)
        out, st = eng.generate(ids, 80, mode="kangaroo")
        if ref is None:
            ref = (out, st.accept_lengths, st.draft_lengths)
        else:
            assert (out, st.accept_lengths, st.draft_lengths) == ref, (blocks, teams)
    eng.set_tuning(attn_blocks=16, attn_teams=1)
    # blocks without keys fetch the head of the o_proj weight stream and discard it: off, whole row groups from the
    # first chunk on, and the default share must all give the same tokens
    keep = (eng.get_tuning("attn_prefetch"), eng.get_tuning("attn_prefetch_min"))
    for share, first in ((1, 1), (9, 1), (4, 3)):
        eng.set_tuning(attn_prefetch=share, attn_prefetch_min=first)
        out, st = eng.generate(ids, 80, mode="kangaroo")
        assert (out, st.accept_lengths, st.draft_lengths) == ref, (share, first)
    eng.set_tuning(attn_prefetch=keep[0], attn_prefetch_min=keep[1])


def test_zz_explained_cases_equal_the_allow_list():
    """Runs last in this file: the near-tie escapes taken in this process are exactly the committed list (a new
    one already failed where it occurred; a listed one that no longer occurs must be struck from the list)."""
    ran = {(e["case"], e["prompt"], e["kind"], e["position"]) for e in _SEEN if not e["case"].startswith("tp:")}
    if os.environ.get("KG_EXPLAINED_UPDATE"):
        pytest.skip("allow-list is being regenerated")
    if not ran and not _ALLOW_KEYS:
        return
    # only meaningful when the whole file ran in this process
    import sys
    if any(a.startswith("tests/test_model_gpu.py::") or "-k" == a for a in sys.argv):
        pytest.skip("partial run")
    listed = {k for k in _ALLOW_KEYS if not k[0].startswith("tp:")}          # the tensor-parallel cases are checked where they run (tests/test_tp.py)
    assert ran == listed, f"not on the list: {sorted(ran - listed)}; listed but did not occur: {sorted(listed - ran)}"
