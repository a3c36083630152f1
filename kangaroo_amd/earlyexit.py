"""Early-exit base model over the HIP engine.

Mirror of the reference's EarlyExitLlamaForCausalLM (kangaroo/earlyexit.py:6-71): same
attribute names (`past_key_values`, `early_exit_layer`), same method
`forward_draft_or_large_model(in_tokens_small=, in_features_large=, position_ids=)`,
same errors (AssertionError when the cache is unset, ValueError for both / neither
input).  The K/V tensors handed out are VIEWS `[1, Hkv, :len, hd]` of the engine's
pre-allocated caches: the reference loop's `.shape[2]` reads and `[:, :, :n, :]`
truncations keep working, while appends are in place and truncation costs nothing.
Cache slot == position id, so the length of the view is the position of the next row.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .engine import KangarooEngine


def _check_positions(position_ids, past: int, n: int):
    """Cache slot == position id on this path: the rows of a pass sit at positions past .. past+n-1, which is
    what the reference's loop passes (evaluation/inference_kangaroo.py:29,57,60,86) and what
    kangaroo/earlyexit.py:34-39 defaults to.  Anything else is refused (host or device tensor alike; a device
    tensor costs one read-back here, in the pass-by-pass interface only)."""
    if position_ids is None:
        return
    got = position_ids.reshape(-1).tolist()
    if got != list(range(past, past + n)):
        raise ValueError(f"position ids must continue the cache (slot == position): expected {past}..{past + n - 1}, got {got[:4]}{'...' if len(got) > 4 else ''}")


class _HiddenStates:
    """hidden_states of the prefill output: index EARLY_STOP_LAYER (input of the exit
    layer) and -1 / N (after the final norm) are materialised; the loop reads only those
    (evaluation/inference_kangaroo.py:32,35)."""

    def __init__(self, n_layers: int, items: dict):
        self._n = n_layers + 1
        self._items = items

    def __len__(self):
        return self._n

    def __getitem__(self, i: int) -> torch.Tensor:
        if i < 0:
            i += self._n
        if i in self._items:
            return self._items[i]
        raise NotImplementedError(f"hidden_states[{i}] is not kept by the HIP prefill (kept: {sorted(self._items)})")


class BaseModelOutput:
    def __init__(self, logits, hidden_states, past_key_values):
        self.logits = logits
        self.hidden_states = hidden_states
        self.past_key_values = past_key_values


class EarlyExitLlamaForCausalLM:
    def __init__(self, engine: KangarooEngine, EARLY_STOP_LAYER: int):
        if EARLY_STOP_LAYER != engine.exit_layer:
            raise ValueError("EARLY_STOP_LAYER differs from the layer split the engine was built with")
        self.engine = engine
        self.past_key_values: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None
        self.early_exit_layer = EARLY_STOP_LAYER
        self.config = engine.shape

    @property
    def device(self):
        return self.engine.device

    def eval(self):
        return self

    def _views(self, layer: int, length: int):
        return self.engine.kv_views(layer, length)

    def _chunk_rows(self) -> int:
        return self.engine.get_tuning("prefill_rows")

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                 output_hidden_states: bool = True, logits_rows: str = "all") -> BaseModelOutput:
        """Prefill (evaluation/inference_kangaroo.py:30): all layers over the prompt.
        logits_rows='last' computes only the final row's logits (shape [1, 1, V])."""
        e = self.engine
        ids = input_ids.reshape(-1)
        T = ids.numel()
        if T > e.max_length:
            raise ValueError("prompt longer than the cache")
        _check_positions(position_ids, 0, T)
        e.set_tokens(ids, 0)
        L, N, d = self.early_exit_layer, e.shape.num_hidden_layers, e.shape.hidden_size
        whole = e.whole_prompt_rows()
        if whole and T >= whole and L == e.exit_layer:
            # the prompt through the layers as a whole, one matrix product per projection (what the fused loop's prefill runs)
            early, final, lg = e.prefill_rows(T, logits_rows == "all")
            lg = (lg if lg is not None else e.head_logits(final[T - 1:T])).unsqueeze(0)
            pkv = tuple(self._views(l, T) for l in range(N))
            return BaseModelOutput(lg, _HiddenStates(N, {L: early.unsqueeze(0), N: final.unsqueeze(0)}), pkv)
        C = self._chunk_rows()
        early = torch.empty((T, d), dtype=e.dtype, device=e.device)
        final = torch.empty((T, d), dtype=e.dtype, device=e.device)
        logits = [] if logits_rows == "all" else None
        for c0 in range(0, T, C):
            n = min(C, T - c0)
            early[c0:c0 + n] = e.forward_layers(0, L, None, c0, n)
            final[c0:c0 + n] = e.final_norm(e.forward_layers(L, N, early[c0:c0 + n], c0, n))
            if logits is not None:
                logits.append(e.head_logits(final[c0:c0 + n]))
        if logits is None:
            lg = e.head_logits(final[T - 1:T]).unsqueeze(0)
        else:
            lg = torch.cat(logits, 0).unsqueeze(0)
        pkv = tuple(self._views(l, T) for l in range(N))
        hs = _HiddenStates(N, {L: early.unsqueeze(0), N: final.unsqueeze(0)})
        return BaseModelOutput(lg, hs, pkv)

    forward = __call__

    @torch.no_grad()
    def forward_draft_or_large_model(self, in_tokens_small=None, in_features_large=None, position_ids=None):
        assert self.past_key_values is not None
        if in_tokens_small is not None and in_features_large is not None:
            raise ValueError("You cannot specify both in_tokens_small and in_features_large at the same time")
        elif in_tokens_small is not None:
            seq_length = in_tokens_small.shape[1]
        elif in_features_large is not None:
            seq_length = in_features_large.shape[1]
        else:
            raise ValueError("You have to specify either in_tokens_small or in_features_large")
        e = self.engine
        small = in_tokens_small is not None
        focus = 0 if small else self.early_exit_layer
        past = self.past_key_values[focus][0].shape[2]
        _check_positions(position_ids, past, seq_length)
        if small:
            e.set_tokens(in_tokens_small, past)
            h = e.forward_layers(0, self.early_exit_layer, None, past, seq_length)
            layers = range(0, self.early_exit_layer)
        else:
            h = e.forward_layers(self.early_exit_layer, e.shape.num_hidden_layers, in_features_large[0], past, seq_length)
            layers = range(self.early_exit_layer, e.shape.num_hidden_layers)
        for l in layers:
            self.past_key_values[l] = self._views(l, past + seq_length)
        if not small:
            return h.unsqueeze(0), e.final_norm(h).unsqueeze(0)
        return h.unsqueeze(0)
