"""Golden vectors of the adapter training step, from the REFERENCE's own AdapterModel (build container only).

TEST INFRASTRUCTURE.  Imports kangaroo.adapter.AdapterModel from /root/reference at run time and runs the
statements of train.py:258-296 around it on CPU in fp32 (no accelerate: its fp16 autocast needs a GPU; the
optimizer, scheduler and clipping calls are the ones train.py makes, and the loss is divided by the
gradient_accumulation_steps of train.py:8,46-47 before backward, as accelerator.backward does).  Writes DATA only:
tests/golden/train_<case>.npz = inputs, initial parameters, per-step loss / expected acceptance, the first
step's gradients and the parameters after the last step.

Usage:  python oracle/gen_golden_train.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle.gen_golden import _import_reference          # noqa: E402  (same import stubs as the decode fixtures)

GRAD_ACCUM = 8      # train.py:8,46-47: Accelerator(gradient_accumulation_steps=8); the loop never enters accelerator.accumulate(),
                    # so every iteration steps the optimizer, and accelerator.backward() divides the loss by 8 first

CASES = [
    # name, hidden, heads, kv heads, vocab, batch lengths, steps, lr, warm-up steps
    ("train_tiny_mha", 64, 4, 4, 97, [13, 9], 4, 5e-3, 2),
    ("train_tiny_gqa", 64, 4, 2, 97, [16, 5, 11], 3, 2e-3, 1),
]


def make_batch(g, lens, d):
    """What CustomDataset + DataCollatorWithPadding hand to the loop (train.py:101-185): per sample the
    early-exit hidden states and the final hidden states shifted left by one with a zero row appended,
    loss_mask with its last entry cleared, everything right-padded with zeros to the longest sample."""
    S = max(lens)
    B = len(lens)
    early = torch.zeros(B, S, d)
    target = torch.zeros(B, S, d)
    att = torch.zeros(B, S, dtype=torch.long)
    lm = torch.zeros(B, S, dtype=torch.long)
    for b, n in enumerate(lens):
        he = torch.randn(n, d, generator=g)
        hb = torch.randn(n, d, generator=g) * 1.5
        early[b, :n - 1] = he[1:]
        target[b, :n - 1] = hb[1:]
        att[b, :n] = 1
        m = (torch.rand(n, generator=g) > 0.3).long()
        m[-1] = 0
        lm[b, :n] = m
    return {"hidden_states_early": early, "target": target, "attention_mask": att, "loss_mask": lm}


def main():
    ref_adapter, _ = _import_reference()
    from transformers import LlamaConfig, get_linear_schedule_with_warmup
    out_dir = os.path.join(REPO, "tests", "golden")
    for name, d, H, Hkv, V, lens, steps, lr, warm in CASES:
        torch.manual_seed(0)
        g = torch.Generator().manual_seed(len(name) + d)
        cfg = LlamaConfig(hidden_size=d, intermediate_size=2 * d, num_hidden_layers=1, num_attention_heads=H,
                          num_key_value_heads=Hkv, vocab_size=V, rms_norm_eps=1e-6, max_position_embeddings=64, pad_token_id=0)
        model = ref_adapter.AdapterModel(cfg)
        with torch.no_grad():
            for p in model.parameters():
                if p.dim() == 2:
                    p.copy_(torch.randn(p.shape, generator=g) * (1.0 / d ** 0.5))
                else:
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
        head = torch.nn.Linear(d, V, bias=False)
        with torch.no_grad():
            head.weight.copy_(torch.randn(V, d, generator=g) * 0.3)
        for p in head.parameters():
            p.requires_grad = False
        names = {"input_layernorm": model.layers[0].input_layernorm.weight, "q_proj": model.layers[0].self_attn.q_proj.weight,
                 "k_proj": model.layers[0].self_attn.k_proj.weight, "v_proj": model.layers[0].self_attn.v_proj.weight,
                 "o_proj": model.layers[0].self_attn.o_proj.weight, "norm": model.norm.weight}
        rec = {f"init.{k}": v.detach().numpy().copy() for k, v in names.items()}
        rec["head"] = head.weight.detach().numpy().copy()
        batches = [make_batch(g, lens, d) for _ in range(steps)]
        # train.py:227-236
        optimizer = torch.optim.AdamW(model.parameters(), lr=lr, betas=(0.9, 0.95))
        scheduler = get_linear_schedule_with_warmup(optimizer, num_warmup_steps=warm, num_training_steps=100)
        losses, accs, lrs = [], [], []
        model.train()
        for i, data in enumerate(batches):
            # train.py:259-296 (accelerator.backward / clip_grad_value_ are loss.backward / torch's clip_grad_value_)
            optimizer.zero_grad()
            data = dict(data)
            data["hidden_states_early"] = data["hidden_states_early"].clone().requires_grad_(True)     # train.py:262
            predict = model(inputs_embeds=data["hidden_states_early"], attention_mask=data["attention_mask"])
            with torch.no_grad():
                target_head = head(data["target"])
                target_p = torch.nn.Softmax(dim=2)(target_head).detach()
            out_head = head(predict)
            prob_exit = torch.nn.functional.softmax(out_head, dim=2)
            prob_last = torch.nn.functional.softmax(target_head, dim=2)
            prob_acc = torch.min(prob_last, prob_exit).sum(dim=2)
            out_logp = torch.nn.LogSoftmax(dim=2)(out_head)
            loss_mask = data["loss_mask"][:, :, None]
            plogp = target_p * out_logp
            loss = -torch.sum(torch.sum(loss_mask * plogp, 2)) / loss_mask.sum()
            prob_acc = torch.sum(data["loss_mask"] * prob_acc) / data["loss_mask"].sum()
            lrs.append(optimizer.param_groups[0]["lr"])
            (loss / GRAD_ACCUM).backward()             # accelerator.backward(loss) (train.py:284) with gradient_accumulation_steps=8
            if i == 0:
                for k, v in names.items():
                    rec[f"grad0.{k}"] = v.grad.detach().numpy().copy()
                rec["predict0"] = predict.detach().numpy().copy()
            torch.nn.utils.clip_grad_value_(model.parameters(), 0.5)
            optimizer.step()
            scheduler.step()
            losses.append(float(loss.detach()))
            accs.append(float(prob_acc))
            for k, v in data.items():
                rec[f"batch{i}.{k}"] = v.detach().numpy().copy()
        for k, v in names.items():
            rec[f"final.{k}"] = v.detach().numpy().copy()
        rec["loss"] = np.array(losses, dtype=np.float64)
        rec["prob_acc"] = np.array(accs, dtype=np.float64)
        rec["lr"] = np.array(lrs, dtype=np.float64)
        rec["meta"] = np.array([d, H, Hkv, V, steps, warm], dtype=np.int64)
        rec["grad_accum"] = np.array([GRAD_ACCUM], dtype=np.int64)
        rec["base_lr"] = np.array([lr], dtype=np.float64)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, "loss", losses, "accept", accs, "lr", lrs)


if __name__ == "__main__":
    main()
