"""CPU restatement of the reference's adapter training step (train.py:258-296).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
fp32 torch with autograd; the forward is kangaroo/adapter.py:351-460 for a padded batch (one attention-only
decoder layer, then the final RMSNorm), the loss is the soft-label cross entropy through the frozen lm_head
(train.py:265-278), the update is clip_grad_value_ + AdamW + linear warm-up (train.py:227-236,284-296).
Pinned by tests/golden/train_*.npz, which oracle/gen_golden_train.py writes from the reference's own
AdapterModel, torch.optim.AdamW and transformers' scheduler.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

from .kangaroo_oracle import rope_tables, _rotate_half

PARAM_NAMES = ("input_layernorm", "q_proj", "k_proj", "v_proj", "o_proj", "norm")


def additive_mask(attention_mask: torch.Tensor) -> torch.Tensor:
    """kangaroo/adapter.py:327-348 with :38-67: causal mask plus the expanded padding mask, both built from
    finfo(float32).min and ADDED (a key that is both in the future and padded gets -inf)."""
    B, S = attention_mask.shape
    mn = torch.finfo(torch.float32).min
    causal = torch.full((S, S), mn, dtype=torch.float32)
    causal.masked_fill_(torch.arange(S).unsqueeze(0) <= torch.arange(S).unsqueeze(1), 0.0)
    inv = 1.0 - attention_mask.to(torch.float32)[:, None, None, :].expand(B, 1, S, S)
    pad = inv.masked_fill(inv.to(torch.bool), mn)
    return pad + causal[None, None]


def adapter_forward(x: torch.Tensor, attention_mask: torch.Tensor, p: Dict[str, torch.Tensor], n_heads: int, n_kv_heads: int,
                    eps: float, rope_base: float = 10000.0) -> torch.Tensor:
    """predict = AdapterModel(inputs_embeds=x, attention_mask=mask) (kangaroo/adapter.py:378-460): x [B, S, d]."""
    B, S, d = x.shape
    hd = d // n_heads

    def norm(h, w):        # kangaroo/adapter.py:254-259
        var = h.pow(2).mean(-1, keepdim=True)
        return w * (h * torch.rsqrt(var + eps))

    xn = norm(x, p["input_layernorm"])
    q = (xn @ p["q_proj"].T).view(B, S, n_heads, hd).transpose(1, 2)
    k = (xn @ p["k_proj"].T).view(B, S, n_kv_heads, hd).transpose(1, 2)
    v = (xn @ p["v_proj"].T).view(B, S, n_kv_heads, hd).transpose(1, 2)
    cos, sin = rope_tables(hd, S, rope_base)
    cos, sin = cos[None, None], sin[None, None]
    q = q * cos + _rotate_half(q) * sin          # :87-95, positions 0..S-1
    k = k * cos + _rotate_half(k) * sin
    rep = n_heads // n_kv_heads
    if rep > 1:
        k = k.repeat_interleave(rep, dim=1)
        v = v.repeat_interleave(rep, dim=1)
    scores = q @ k.transpose(2, 3) / math.sqrt(hd) + additive_mask(attention_mask)
    a = torch.softmax(scores, dim=-1, dtype=torch.float32)
    o = (a @ v).transpose(1, 2).reshape(B, S, d)
    h = x + o @ p["o_proj"].T
    return norm(h, p["norm"])


def loss_and_accept(predict: torch.Tensor, target: torch.Tensor, loss_mask: torch.Tensor, head_w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """train.py:265-279: soft-label cross entropy against softmax(head(target)) and the expected acceptance
    sum_v min(p_last, p_exit), both averaged over the positions where loss_mask is 1."""
    with torch.no_grad():
        target_head = target @ head_w.T
        target_p = torch.softmax(target_head, dim=2)
    out_head = predict @ head_w.T
    out_logp = torch.log_softmax(out_head, dim=2)
    lm = loss_mask.to(torch.float32)
    loss = -torch.sum(torch.sum(lm[:, :, None] * (target_p * out_logp), 2)) / lm.sum()
    with torch.no_grad():
        acc = torch.minimum(target_p, torch.softmax(out_head, dim=2)).sum(dim=2)
        prob_acc = torch.sum(lm * acc) / lm.sum()
    return loss, prob_acc


def linear_warmup_lr(base_lr: float, step: int, warmup: int, total: int) -> float:
    """transformers.get_linear_schedule_with_warmup as train.py:235 uses it: the factor of optimizer step `step`
    (0-based; the very first step runs at lr 0)."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    return base_lr * max(0.0, (total - step) / max(1, total - warmup))


class AdamWState:
    """torch.optim.AdamW(lr, betas=(0.9, 0.95)) with its defaults eps=1e-8, weight_decay=1e-2 (train.py:227)."""

    def __init__(self, params: Dict[str, torch.Tensor], betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-2):
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0
        self.betas, self.eps, self.wd = betas, eps, weight_decay

    def step(self, params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], lr: float) -> None:
        self.t += 1
        b1, b2 = self.betas
        for k in params:
            g = grads[k]
            params[k].mul_(1.0 - lr * self.wd)
            self.m[k].mul_(b1).add_(g, alpha=1.0 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
            bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            params[k].addcdiv_(self.m[k], denom, value=-lr / bc1)


def train_step(params: Dict[str, torch.Tensor], opt: AdamWState, head_w, batch, n_heads, n_kv_heads, eps, lr: float, grad_clip: float = 0.5,
               grad_accum: int = 8):
    """One iteration of train.py:258-296 on fp32 tensors; returns (loss, prob_acc, grads before clipping).
    grad_accum: train.py:8,46-47 build Accelerator(gradient_accumulation_steps=8) and accelerator.backward (train.py:284)
    divides the loss by it, although the loop steps the optimizer every iteration: the gradients that are clipped at
    0.5 and fed to AdamW are those of loss / 8.  The returned loss is the undivided one (what train.py:281 prints)."""
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    predict = adapter_forward(batch["hidden_states_early"], batch["attention_mask"], leaves, n_heads, n_kv_heads, eps)
    loss, prob_acc = loss_and_accept(predict, batch["target"], batch["loss_mask"], head_w)
    (loss / grad_accum).backward()
    grads = {k: leaves[k].grad.detach().clone() for k in leaves}
    clipped = {k: g.clamp(-grad_clip, grad_clip) for k, g in grads.items()}       # accelerator.clip_grad_value_ (train.py:285)
    opt.step(params, clipped, lr)
    return float(loss.detach()), float(prob_acc), grads
