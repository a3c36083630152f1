"""The Kangaroo adapter over the HIP engine.

Mirror of the reference's AdapterModel (kangaroo/adapter.py:317-460): one
attention-only decoder layer (RMSNorm -> MHA with RoPE and KV cache -> residual, no
MLP, :262-314) followed by a final RMSNorm (:456).  `forward_early_stop` keeps the
reference's keyword names and its return convention `(hidden, ((k, v),))`; k and v are
views `[1, Hkv, :len, hd]` of the engine's adapter cache, so the caller's truncation
(evaluation/inference_kangaroo.py:114-118) is a length change, not a copy.
"""
from __future__ import annotations

from typing import Optional

import torch

from .earlyexit import _check_positions
from .engine import KangarooEngine


class AdapterModel:
    def __init__(self, engine: KangarooEngine):
        self.engine = engine
        self.config = engine.shape

    def eval(self):
        return self

    def half(self):
        return self

    def to(self, *a, **k):
        return self

    def _views(self, length: int):
        return (self.engine.adapter_kv_views(length),)

    @torch.no_grad()
    def forward_early_stop(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                           inputs_embeds: Optional[torch.Tensor] = None, use_cache: Optional[bool] = None,
                           output_attentions=None, output_hidden_states=None, return_dict=None, std=None):
        if inputs_embeds is None:
            raise ValueError("inputs_embeds is required")
        if attention_mask is not None:
            raise NotImplementedError("only the all-ones mask of the decode loop is supported")
        e = self.engine
        _, q_len, d = inputs_embeds.shape
        past = past_key_values[0][0].shape[2] if past_key_values is not None else 0
        if past + q_len > e.max_length:
            raise ValueError("sequence exceeds the cache")
        _check_positions(position_ids, past, q_len)
        x = inputs_embeds[0]
        whole = e.whole_prompt_rows()
        if whole and past == 0 and q_len >= whole:
            out = e.adapter_forward_rows(x, q_len)                # the adapter's prompt pass as a whole (inference_kangaroo.py:36-38)
        else:
            out = torch.empty((q_len, d), dtype=e.dtype, device=e.device)
            C = e.get_tuning("prefill_rows")
            for c0 in range(0, q_len, C):
                n = min(C, q_len - c0)
                out[c0:c0 + n] = e.adapter_forward(x[c0:c0 + n], past + c0, n)
        hidden = out.unsqueeze(0)
        if use_cache:
            return hidden, self._views(past + q_len)
        return hidden

    forward = forward_early_stop
    __call__ = forward_early_stop
