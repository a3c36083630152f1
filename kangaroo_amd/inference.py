"""Decode entry points with the reference's `forward_func` signature.

`kangaroo_forward(inputs, model, tokenizer, max_new_tokens, ...)` has the signature and
the 4-tuple return of the reference's kangaroo_forward
(evaluation/inference_kangaroo.py:15,126-129).  It runs the whole loop on the device
(one captured graph per verify round; early-exit test, accept / reject and cache
rollback in kernels).  `kangaroo_forward_stepwise` walks the same algorithm from Python
through the model classes' reference-shaped methods, one call per pass, the way the
reference's loop drives them; it exists to exercise that surface, not for speed.
`vanilla_forward` is plain greedy decoding on the same kernels (the comparator the
reference gets from HF generate, evaluation/inference_baseline.py:14-25).
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def _eos(tokenizer) -> int:
    return tokenizer if isinstance(tokenizer, int) else tokenizer.eos_token_id


def _ids(inputs) -> torch.Tensor:
    return inputs if isinstance(inputs, torch.Tensor) else inputs.input_ids


def _rule_of(do_sample, hyper_k, hyper_p, temperature, epsilon=0.3, delta=0.09):
    """The verification rule as the reference's loops name it: False / "greedy" (inference_kangaroo.py:95-104),
    "top_k" with hyper_k (inference_kangaroo_MTBench.py:377-389) or "top_p" with hyper_p and temperature
    (:356-374; temperature 0 means greedy there, :282-290), or "typical" with temperature, epsilon and delta
    (:293-314, reproduced as written there: see KangarooEngine.generate).  "typical_sampling" (:316-354) draws
    from torch.multinomial and is not offered.  Returns engine.generate keyword arguments."""
    if do_sample in (False, True, None, "greedy"):      # the loop of inference_kangaroo.py is greedy whatever the flag says (:15-129; its caller passes True, :239)
        return {}
    if do_sample == "top_k":
        if int(hyper_k) < 1:
            raise ValueError("hyper_k must be >= 1")
        return {} if temperature == 0.0 else dict(accept_top_k=int(hyper_k))
    if do_sample == "top_p":
        if not 0.0 < float(hyper_p) <= 1.0:
            raise ValueError("hyper_p must be in (0, 1]")
        return {} if temperature == 0.0 else dict(accept_top_p=float(hyper_p), temperature=float(temperature))
    if do_sample == "typical":
        return {} if temperature == 0.0 else dict(accept_typical=(float(epsilon), float(delta)), temperature=float(temperature))
    raise NotImplementedError(f"verification rule {do_sample!r} is not implemented (greedy, top_k, top_p and typical are)")


def kangaroo_forward(inputs, model, tokenizer, max_new_tokens, do_sample=False, max_length=2048, EARLY_STOP_LAYER=2,
                     SPECULATIVE_DECODING_STEPS=6, threshold=0.6, return_stats=False, hyper_k=2, hyper_p=0.8, temperature=0.7,
                     epsilon=0.3, delta=0.09):
    ids = _ids(inputs)
    if ids.shape[0] != 1:
        raise ValueError("batch size 1 only")
    if EARLY_STOP_LAYER != model.engine.exit_layer:
        raise ValueError("EARLY_STOP_LAYER differs from the model's layer split")
    out, st = model.engine.generate(ids[0], max_new_tokens, mode="kangaroo", steps=SPECULATIVE_DECODING_STEPS,
                                    threshold=threshold, eos_token_id=_eos(tokenizer), max_length=max_length,
                                    **_rule_of(do_sample, hyper_k, hyper_p, temperature, epsilon, delta))
    accept_length_list = [1] + st.accept_lengths
    res = ([out], st.new_tokens, len(accept_length_list) - 1, accept_length_list)
    return res + (st,) if return_stats else res


def vanilla_forward(inputs, model, tokenizer, max_new_tokens, max_length=2048, return_stats=False):
    """Greedy decoding, one token per replayed graph.  Returns ([ids], new_tokens, steps, [1]*steps)."""
    ids = _ids(inputs)
    out, st = model.engine.generate(ids[0], max_new_tokens, mode="vanilla", eos_token_id=_eos(tokenizer), max_length=max_length)
    res = ([out], st.new_tokens, st.new_tokens, [1] * st.new_tokens)
    return res + (st,) if return_stats else res


@torch.no_grad()
def kangaroo_forward_stepwise(inputs, model, tokenizer, max_new_tokens, do_sample=False, max_length=2048,
                              EARLY_STOP_LAYER=2, SPECULATIVE_DECODING_STEPS=6, threshold=0.6, trace=None):
    """Same algorithm, driven pass by pass through base_model / adapter_model / head_model."""
    ids = _ids(inputs)
    dev = ids.device
    eos = _eos(tokenizer)
    T0 = ids.shape[1]
    steps = SPECULATIVE_DECODING_STEPS
    toks = torch.full((1, max_length), eos, dtype=torch.long, device=dev)
    toks[:, :T0] = ids
    base, adapter, head = model.base_model, model.adapter_model, model.head_model

    out = base(ids, position_ids=None, output_hidden_states=True, logits_rows="last")
    base.past_key_values = list(out.past_key_values)
    toks[0, T0] = int(out.logits[0, -1].float().argmax())
    _, akv = adapter.forward_early_stop(inputs_embeds=out.hidden_states[EARLY_STOP_LAYER], use_cache=True)

    accepts: List[int] = [1]
    s = T0
    limit = min(max_length, T0 + max_new_tokens) - 1 - steps
    stop = False
    exited = None
    while s < limit and not stop:
        e = s + 1
        score = 0.0
        prev_exited = exited
        rows = []
        for step in range(steps + 1):
            missing = akv[0][0].shape[2] < e - 1          # adapter cache one row short after a full accept
            h = base.forward_draft_or_large_model(in_tokens_small=toks[:, e - 1:e])
            rows.append(h)
            if step == steps or (step > 0 and score < threshold):
                break
            a_in = torch.cat([prev_exited[:, -1:, :], h], dim=1) if missing else h
            a_out, akv = adapter.forward_early_stop(inputs_embeds=a_in, past_key_values=akv, use_cache=True)
            lg = head(a_out[:, -1:, :]).float()
            toks[0, e] = lg[0, -1].argmax()
            e += 1
            score = float(lg.softmax(dim=-1).max())
            if trace is not None:
                trace.setdefault("p_max", []).append(score)
        exited = torch.cat(rows, dim=1)
        assert base.past_key_values[EARLY_STOP_LAYER][0].shape[2] == s
        _, hn = base.forward_draft_or_large_model(in_features_large=exited)
        target = head(hn).float().argmax(dim=-1)[0].tolist()
        q = e - s
        if trace is not None:
            trace.setdefault("draft_lengths", []).append(q - 1)
        drafted = toks[0, s + 1:s + q].tolist()
        i = 0
        while i < q - 1 and target[i] != eos and target[i] == drafted[i]:
            i += 1
        toks[0, s + 1 + i] = target[i]
        stop = target[i] == eos
        accepts.append(i + 1)
        s = s + 1 + i
        if base.past_key_values[0][0].shape[2] > s:
            base.past_key_values = [(k[:, :, :s, :], v[:, :, :s, :]) for k, v in base.past_key_values]
        if akv[0][0].shape[2] > s:
            akv = tuple((k[:, :, :s, :], v[:, :, :s, :]) for k, v in akv)
    return [toks[0, :s + 1].tolist()], s - T0 + 1, len(accepts) - 1, accepts
