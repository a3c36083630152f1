"""Training data for the adapter: the per-sample record of the reference's data/ge_data_all_vicuna.py (:139-171).

For one tokenised conversation the reference runs the full model once and keeps, per position, the input of the
early-exit layer and the last hidden state after the final norm (HF `hidden_states[L]` and `hidden_states[-1]`),
together with the ids and the loss mask; `train.py:109-117` reads exactly these keys.  Here the same pass is the
device prefill of the mirror base model (128-row GEMM passes), which materialises just those two hidden states.
Tokenisation, the conversation template and the loss mask (:39-115) stay with the caller's dataset code.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, Tuple

import torch


@torch.no_grad()
def ge(model, input_ids: torch.Tensor, loss_mask: torch.Tensor) -> Dict[str, torch.Tensor]:
    """data/ge_data_all_vicuna.py:139-155 for one sample: model = KangarooModel (or its base_model); input_ids [1, T]
    or [T]; returns CPU tensors under the reference's keys (only the exit layer the model was built with is kept)."""
    base = getattr(model, "base_model", model)
    ids = input_ids.reshape(1, -1)
    out = base(ids.to(base.device), output_hidden_states=True, logits_rows="last")
    L = base.early_exit_layer
    return {"input_ids": ids[0].cpu(), "loss_mask": loss_mask.reshape(-1).cpu(),
            f"hidden_state_layer{L}": out.hidden_states[L][0].cpu(), "hidden_state": out.hidden_states[-1][0].cpu()}


def writedata(name: str, data_point: Dict[str, torch.Tensor]) -> str:
    """:161-166: data_<k>.ckpt with k = files already in the directory."""
    os.makedirs(name, exist_ok=True)
    path = os.path.join(name, f"data_{len(os.listdir(name))}.ckpt")
    torch.save(data_point, path)
    return path


def generate(model, samples: Iterable[Tuple[torch.Tensor, torch.Tensor]], outdir: str) -> int:
    """:169-171 over (input_ids, loss_mask) pairs; returns the number of files written."""
    n = 0
    for ids, mask in samples:
        writedata(outdir, ge(model, ids, mask))
        n += 1
    return n
