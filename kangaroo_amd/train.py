"""Adapter training on MI355X: the loop body of the reference's train.py (:258-296) over the C ABI.

The adapter is one attention-only decoder layer plus the final RMSNorm (kangaroo/adapter.py:317-460), trained so
that the frozen lm_head on its output matches the head on the full model's last hidden state (soft-label cross
entropy, train.py:265-278).  Per step and rank:

    forward   RMSNorm -> q/k/v products -> RoPE -> scores -> masked softmax -> context -> o_proj
              -> residual -> RMSNorm -> two lm_head products (adapter output, target)
    loss      one fused pass over both logit rows: log-softmax, cross entropy, expected acceptance, dlogits
    backward  the transposed products, softmax / RoPE / RMSNorm backward kernels
    update    gradient all-reduce over the data-parallel group (RCCL), clip by value, AdamW, 16-bit weight copy

Every matrix product is kg_matmul_nt (csrc/mm.hip: C = A . B^T on the matrix cores, 256 x 256 tiles), the heads of
[B, S, H, hd] tensors addressed in place (kg_matmul_nt_heads), causal products skipping the empty half; a product
whose contraction index is not contiguous in an operand takes that operand K-major where it is one of the [S, S]
matrices (dV^T = dO^T . P, dK^T = Q^T . dS) and transposes it first otherwise (kg_transpose16: the weights once per
step, the head once for good, activations for the weight gradients).  Everything between the products is a
hand-written kernel of csrc/train.hip.  With 288 GB of HBM the step materialises the attention probabilities
[B, H, S, S] and both logit matrices [B*S, V] in 16 bit instead of recomputing them (4 x 2048 tokens at the 7B shape:
1 GiB + 2 x 0.5 GiB).  torch only allocates.

Mixed precision as the reference runs it (accelerate, fp16 autocast, train.py:46): fp32 master weights and
optimizer state, 16-bit activations and GEMM inputs, fp32 statistics; gradients of activations are carried in 16
bit under a constant loss scale that the optimizer kernel divides out.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .config import LlamaShape
from .packing import rope_tables

PARAM_KEYS = {
    "input_layernorm": "layers.0.input_layernorm.weight",
    "q_proj": "layers.0.self_attn.q_proj.weight",
    "k_proj": "layers.0.self_attn.k_proj.weight",
    "v_proj": "layers.0.self_attn.v_proj.weight",
    "o_proj": "layers.0.self_attn.o_proj.weight",
    "norm": "norm.weight",
}


def prepare_sample(data: Dict[str, torch.Tensor], exit_layer, max_len: int = 2048) -> Dict[str, torch.Tensor]:
    """CustomDataset.__getitem__ (train.py:109-148) on one loaded sample: truncate to max_len, shift the early-exit
    and final hidden states (and input ids) left by one with a zero row appended, clear the last loss_mask entry."""
    hidden = data["hidden_state"][:max_len][None, :]
    ids = data["input_ids"][:max_len][None, :]
    loss_mask = data["loss_mask"][:max_len].tolist()
    early = data[f"hidden_state_layer{exit_layer}"][:max_len][None, :]
    length = hidden.shape[1]
    loss_mask[-1] = 0
    zrow = torch.zeros(1, 1, hidden.shape[2], dtype=hidden.dtype)
    return {
        "attention_mask": [1] * length,
        "loss_mask": loss_mask,
        "target": torch.cat((hidden[:, 1:, :], zrow), dim=1),
        "hidden_state_big": hidden,
        "input_ids": torch.cat((ids[:, 1:], torch.zeros(1, 1, dtype=ids.dtype)), dim=1),
        "hidden_state_early": torch.cat((early[:, 1:, :], zrow.to(early.dtype)), dim=1),
    }


class DataCollatorWithPadding:
    """train.py:151-185: right-pad every field with zeros to the longest sample of the batch."""

    def __call__(self, features: Sequence[Dict]) -> Dict[str, torch.Tensor]:
        n = max(f["hidden_state_big"].shape[1] for f in features)

        def pad3(t):
            return torch.cat((t, torch.zeros(t.shape[0], n - t.shape[1], t.shape[2], dtype=t.dtype)), dim=1)

        def pad2(t):
            return torch.cat((t, torch.zeros(t.shape[0], n - t.shape[1], dtype=t.dtype)), dim=1)

        return {
            "input_ids": torch.cat([pad2(f["input_ids"]) for f in features]),
            "hidden_states": torch.cat([pad3(f["hidden_state_big"]) for f in features]),
            "hidden_states_early": torch.cat([pad3(f["hidden_state_early"]) for f in features]),
            "target": torch.cat([pad3(f["target"]) for f in features]),
            "loss_mask": torch.tensor([f["loss_mask"] + [0] * (n - len(f["loss_mask"])) for f in features]),
            "attention_mask": torch.tensor([f["attention_mask"] + [0] * (n - len(f["attention_mask"])) for f in features]),
        }


def linear_warmup_lr(base_lr: float, step: int, warmup: int, total: int) -> float:
    """The factor transformers.get_linear_schedule_with_warmup gives optimizer step `step` (0-based), times
    base_lr (train.py:235): the first step runs at 0."""
    if step < warmup:
        return base_lr * step / max(1, warmup)
    return base_lr * max(0.0, (total - step) / max(1, total - warmup))


class GradSync:
    """Data-parallel gradient mean (what accelerate's DDP wrapper does for train.py:284): gradients are copied
    into fp32 buckets as the backward pass produces them and each bucket's all-reduce is started at once, so the
    exchange of the o_proj bucket overlaps the attention backward; `finish` waits and divides by the world size."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.pending = []

    def start(self, tensors: List[torch.Tensor]) -> List[torch.Tensor]:
        """tensors -> fp32 views of one flat bucket being reduced."""
        if self.world == 1:                                   # nothing to exchange: the gradients as they are
            return [t if t.dtype == torch.float32 else t.to(torch.float32) for t in tensors]
        flat = torch.cat([t.reshape(-1).to(torch.float32) for t in tensors])
        work = self.dist.all_reduce(flat, group=self.group, async_op=True)
        self.pending.append((flat, work))
        out, off = [], 0
        for t in tensors:
            out.append(flat[off:off + t.numel()].view(t.shape))
            off += t.numel()
        return out

    def finish(self) -> None:
        for flat, work in self.pending:
            if work is not None:
                work.wait()
                flat.div_(self.world)
        self.pending = []


class AdapterTrainer:
    """One optimisation step per `step(batch)` call; `state_dict()` returns the adapter in the reference's key names
    (loadable by KangarooModel / kangaroo/kangaroo_model.py:27-30)."""

    def __init__(self, shape: LlamaShape, adapter_sd: Dict[str, torch.Tensor], head_weight: torch.Tensor, dtype=torch.float16,
                 lr: float = 5e-5, betas=(0.9, 0.95), adam_eps: float = 1e-8, weight_decay: float = 1e-2, grad_clip: float = 0.5,
                 num_warmup_steps: int = 2000, total_steps: int = 800000, loss_scale: Optional[float] = None, device="cuda", group=None,
                 grad_accum: int = 8):
        """grad_accum: the reference builds Accelerator(gradient_accumulation_steps=8) (train.py:8,46-47) and never enters
        accelerator.accumulate(), so the optimizer steps every iteration while accelerator.backward (train.py:284) divides
        the loss by 8: the gradients that meet the 0.5 clip and AdamW's eps are those of loss / 8.  With more than one
        rank accelerate's scheduler wrapper also advances the warm-up schedule once per rank and optimizer step."""
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("dtype must be float16 or bfloat16")
        if not torch.cuda.is_available():
            raise _lib.KangarooHipError("adapter training needs a GPU: there is no CPU path")
        self.lib = _lib.lib()
        self.shape, self.dtype, self.device = shape, dtype, torch.device(device)
        self.dt = _lib.DT_F16 if dtype == torch.float16 else _lib.DT_BF16
        self.lr, self.betas, self.adam_eps, self.wd, self.clip = lr, betas, adam_eps, weight_decay, grad_clip
        self.warmup, self.total = num_warmup_steps, total_steps
        self.loss_scale = float(loss_scale) if loss_scale is not None else (256.0 if dtype == torch.float16 else 1.0)
        self.grad_accum = int(grad_accum)
        self.t = 0
        self.sync = GradSync(group)
        self.p = {k: adapter_sd[name].detach().to(self.device, torch.float32).contiguous().clone() for k, name in PARAM_KEYS.items()}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.w16 = {k: v.to(dtype) for k, v in self.p.items() if v.dim() == 2}
        self.head16 = head_weight.detach().to(self.device, dtype).contiguous()
        if self.head16.shape[0] % 8:                                                  # zero rows up to a multiple of 8: their logits are exact zeros the loss never reads
            self.head16 = torch.nn.functional.pad(self.head16, (0, 0, 0, 8 - self.head16.shape[0] % 8)).contiguous()
        self.head16_t = self._transpose(self.head16)                      # [d, V]: the frozen head for dhn = dlogits . W
        cos, sin = rope_tables(shape.head_dim, shape.max_position_embeddings, shape.rope_base, device=self.device)
        self.cos, self.sin = cos.contiguous(), sin.contiguous()

    # -- kernel wrappers -------------------------------------------------------------
    def _st(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    def _rmsnorm_fwd(self, x, gamma):
        rows, d = x.shape
        y = torch.empty_like(x)
        rstd = torch.empty(rows, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.kg_train_rmsnorm_fwd(self.dt, self._p(x), self._p(gamma), self.shape.rms_norm_eps, rows, d, self._p(y), self._p(rstd), self._st()), "kg_train_rmsnorm_fwd")
        return y, rstd

    def _rmsnorm_bwd(self, dy, x, gamma, rstd, want_dx: bool):
        rows, d = x.shape
        dx = torch.empty_like(x) if want_dx else None
        dg = torch.empty(d, dtype=torch.float32, device=self.device)
        scratch = torch.empty(self.lib.kg_train_rmsnorm_bwd_scratch(rows, d), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.kg_train_rmsnorm_bwd(self.dt, self._p(dy), self._p(x), self._p(gamma), self._p(rstd), rows, d, self._p(dx), self._p(dg), self._p(scratch), self._st()),
                   "kg_train_rmsnorm_bwd")
        return dx, dg

    def _rope(self, x, rows, S, n_heads, inverse: bool):
        _lib.check(self.lib.kg_train_rope(self.dt, self._p(x), rows, S, n_heads, self.shape.head_dim, self._p(self.cos), self._p(self.sin), 1 if inverse else 0, self._st()), "kg_train_rope")

    def _mm(self, A, B, out_f32: bool = False):
        """A [M, K] . B [N, K]^T -> [M, N] (row strides may exceed K: views of wider tensors)."""
        M, K = A.shape
        N = B.shape[0]
        assert B.shape[1] == K and A.stride(1) == 1 and B.stride(1) == 1
        out = torch.empty((M, N), dtype=torch.float32 if out_f32 else self.dtype, device=self.device)
        _lib.check(self.lib.kg_matmul_nt(self.dt, self._p(A), self._p(B), self._p(out), M, N, K, A.stride(0), B.stride(0), N,
                                         1, 0, 0, 0, 0, 1 if out_f32 else 0, 0, 1.0, self._st()), "kg_matmul_nt")
        return out

    def _transpose(self, x, batch: int = 1):
        """[batch, R, C] (or [R, C]) -> [batch, C, R], contiguous."""
        x = x.contiguous()
        R, Cn = x.shape[-2], x.shape[-1]
        out = torch.empty(x.shape[:-2] + (Cn, R), dtype=x.dtype, device=x.device)
        _lib.check(self.lib.kg_transpose16(self._p(x), self._p(out), R, Cn, Cn, R, batch, R * Cn, R * Cn, self._st()), "kg_transpose16")
        return out

    # -- one step --------------------------------------------------------------------
    def step(self, batch: Dict[str, torch.Tensor]) -> Dict[str, float]:
        sh, dt, dev = self.shape, self.dtype, self.device
        H, Hkv, hd, d, V = sh.num_attention_heads, sh.num_key_value_heads, sh.head_dim, sh.hidden_size, sh.vocab_size
        if hd % 8 or d % 8:
            raise ValueError("head_dim and hidden_size must be multiples of 8 (16-byte rows)")
        Vp = self.head16.shape[0]                                                    # vocabulary rounded up to whole 16-byte pieces
        x3 = batch["hidden_states_early"].to(dev, dt)
        tgt3 = batch["target"].to(dev, dt)
        key_mask = batch["attention_mask"].to(dev, torch.int32)
        loss_mask = batch["loss_mask"].to(dev, torch.int32)
        B, S0, _ = x3.shape
        if S0 > sh.max_position_embeddings:
            raise ValueError("sequence longer than max_position_embeddings")
        # The collator pads to the longest sample with zero rows and zero masks (train.py:151-185); padding a little
        # further, to a multiple of 8 positions, changes no position that counts and makes every row of the
        # [S, S] matrices a whole number of 16-byte pieces.
        S = (S0 + 7) // 8 * 8
        if S != S0:
            pad = S - S0
            x3 = torch.nn.functional.pad(x3, (0, 0, 0, pad)); tgt3 = torch.nn.functional.pad(tgt3, (0, 0, 0, pad))
            key_mask = torch.nn.functional.pad(key_mask, (0, pad)); loss_mask = torch.nn.functional.pad(loss_mask, (0, pad))
        R = B * S
        x = x3.reshape(R, d).contiguous()
        tgt = tgt3.reshape(R, d).contiguous()
        key_mask = key_mask.contiguous()
        loss_mask = loss_mask.reshape(R).contiguous()
        n_loss = int(loss_mask.sum().item())
        if n_loss == 0:
            raise ValueError("the batch has no position with loss_mask == 1")
        W = self.w16
        scale = 1.0 / math.sqrt(hd)
        rep = H // Hkv
        qd, kvd = H * hd, Hkv * hd

        def heads(A, lda, sA, sA2, Bm, ldb, sB, sB2, out, ldc, sC, sC2, M, N, K, causal):
            # one product per (b, h): A / out at b * sX + h * sX2, B at b * sB + (h // rep) * sB2 (elements)
            _lib.check(self.lib.kg_matmul_nt_heads(self.dt, self._p(A), self._p(Bm), self._p(out), M, N, K, lda, ldb, ldc, B * H, sA, sB, sC,
                                                   H, sA2, sB2, sC2, rep, 0, 0, causal, 1.0, self._st()), "kg_matmul_nt_heads")

        # ---- forward (kangaroo/adapter.py:378-460) ----
        xn, rstd1 = self._rmsnorm_fwd(x, self.p["input_layernorm"])
        q = self._mm(xn, W["q_proj"])                                                # [R, H hd]: heads are read in place below
        k = self._mm(xn, W["k_proj"])
        v = self._mm(xn, W["v_proj"])
        self._rope(q, R, S, H, False)
        self._rope(k, R, S, Hkv, False)
        # scores[b, h] = q[b, :, h] . k[b, :, h // rep]^T over the causal half, then the probabilities in place
        probs = torch.empty((B, H, S, S), dtype=dt, device=dev)
        heads(q, qd, S * qd, hd, k, kvd, S * kvd, hd, probs, S, H * S * S, S * S, S, S, hd, 1)
        _lib.check(self.lib.kg_train_softmax_fwd(self.dt, self._p(probs), self._p(key_mask), B, H, S, scale, self._st()), "kg_train_softmax_fwd")
        # ctx[b, :, h] = probs[b, h] . v[b, :, h // rep]: the contraction runs over the keys, so V goes in transposed
        vt = self._transpose(v.view(B, S, kvd), B)                                   # [B, Hkv hd, S]
        ctx = torch.empty((R, qd), dtype=dt, device=dev)
        heads(probs, S, H * S * S, S * S, vt, S, kvd * S, hd * S, ctx, qd, S * qd, hd, S, hd, S, 2)
        h = self._mm(ctx, W["o_proj"])
        _lib.check(self.lib.kg_train_add(self.dt, self._p(h), self._p(x), h.numel(), self._st()), "kg_train_add")     # h = x + o_proj(ctx)
        hn, rstd2 = self._rmsnorm_fwd(h, self.p["norm"])

        # ---- loss through the frozen head (train.py:265-279) ----
        logits = self._mm(hn, self.head16)                                           # becomes dlogits
        tlogits = self._mm(tgt, self.head16)
        row_loss = torch.empty(R, dtype=torch.float32, device=dev)
        row_acc = torch.empty(R, dtype=torch.float32, device=dev)
        row_rank = torch.empty(R, dtype=torch.int32, device=dev)
        _lib.check(self.lib.kg_train_softce(self.dt, self._p(logits), self._p(tlogits), self._p(loss_mask), R, V, Vp, self.loss_scale / n_loss,
                                            self._p(row_loss), self._p(row_acc), self._p(row_rank), self._st()), "kg_train_softce")
        del tlogits

        # ---- backward: every product as A . B^T, the operand whose contraction index is strided transposed first ----
        dhn = self._mm(logits, self.head16_t)                                        # [R, d] = dlogits . W
        del logits
        dh, dg2 = self._rmsnorm_bwd(dhn, h, self.p["norm"], rstd2, True)
        dh_t = self._transpose(dh)                                                   # [d, R]
        g_o = self._mm(dh_t, self._transpose(ctx), out_f32=True)                     # [d, H hd] = dh^T . ctx, summed over the rows in fp32
        (g_o32, dg2s) = self.sync.start([g_o, dg2])
        dctx = self._mm(dh, self._transpose(W["o_proj"]))                            # [R, H hd] = dh . Wo
        # dprobs[b, h] = dctx[b, :, h] . v[b, :, h // rep]^T (causal half), then dscores in place
        dprobs = torch.empty((B, H, S, S), dtype=dt, device=dev)
        heads(dctx, qd, S * qd, hd, v, kvd, S * kvd, hd, dprobs, S, H * S * S, S * S, S, S, hd, 1)
        # dv[b, :, h]^T = dctx[b, :, h]^T . probs[b, h]: the contraction runs over the queries (at or after each key), i.e. over
        # the ROWS of the stored probabilities, which go in as they lie (K-major B); the result comes out transposed
        dctx_t = self._transpose(dctx.view(B, S, qd), B)                             # [B, H hd, S]
        dv_t = torch.empty((B, qd, S), dtype=dt, device=dev)
        _lib.check(self.lib.kg_matmul_nt_heads(self.dt, self._p(dctx_t), self._p(probs), self._p(dv_t), hd, S, S, S, S, S, B * H, qd * S, H * S * S, qd * S,
                                               H, hd * S, S * S, hd * S, 1, 2, 0, 4, 1.0, self._st()), "kg_matmul_nt_heads")
        dvh = self._transpose(dv_t, B).view(B, S, H, hd)
        _lib.check(self.lib.kg_train_softmax_bwd(self.dt, self._p(probs), self._p(dprobs), B, H, S, scale, self._st()), "kg_train_softmax_bwd")
        del probs
        # dq[b, :, h] = dscores[b, h] . k[b, :, h // rep] (keys contracted: K transposed); dk[b, :, h] = dscores[b, h]^T . q[b, :, h]
        kt = self._transpose(k.view(B, S, kvd), B)                                   # [B, Hkv hd, S]
        dq = torch.empty((R, qd), dtype=dt, device=dev)
        heads(dprobs, S, H * S * S, S * S, kt, S, kvd * S, hd * S, dq, qd, S * qd, hd, S, hd, S, 2)
        q_t = self._transpose(q.view(B, S, qd), B)                                   # [B, H hd, S]
        dk_t = torch.empty((B, qd, S), dtype=dt, device=dev)
        _lib.check(self.lib.kg_matmul_nt_heads(self.dt, self._p(q_t), self._p(dprobs), self._p(dk_t), hd, S, S, S, S, S, B * H, qd * S, H * S * S, qd * S,
                                               H, hd * S, S * S, hd * S, 1, 2, 0, 4, 1.0, self._st()), "kg_matmul_nt_heads")
        del dprobs
        dkh = self._transpose(dk_t, B).view(B, S, H, hd)
        if rep > 1:                                                                  # query heads that shared a key / value head
            dk = dkh.view(B, S, Hkv, rep, hd).sum(dim=3).reshape(R, kvd).contiguous()
            dv = dvh.view(B, S, Hkv, rep, hd).sum(dim=3).reshape(R, kvd).contiguous()
        else:
            dk, dv = dkh.view(R, kvd), dvh.view(R, kvd)
        self._rope(dq, R, S, H, True)
        self._rope(dk, R, S, Hkv, True)
        xn_t = self._transpose(xn)                                                   # [d, R]
        g_q = self._mm(self._transpose(dq), xn_t, out_f32=True)                      # [H hd, d] = dq^T . xn
        g_k = self._mm(self._transpose(dk), xn_t, out_f32=True)
        g_v = self._mm(self._transpose(dv), xn_t, out_f32=True)
        dxn = self._mm(dq, self._transpose(W["q_proj"]))                             # [R, d] = dq . Wq + dk . Wk + dv . Wv
        for dd, key in ((dk, "k_proj"), (dv, "v_proj")):
            t = self._mm(dd, self._transpose(W[key]))
            _lib.check(self.lib.kg_train_add(self.dt, self._p(dxn), self._p(t), dxn.numel(), self._st()), "kg_train_add")
        _, dg1 = self._rmsnorm_bwd(dxn, x, self.p["input_layernorm"], rstd1, False)
        (g_q32, g_k32, g_v32, dg1s) = self.sync.start([g_q, g_k, g_v, dg1])
        self.sync.finish()
        grads = {"input_layernorm": dg1s, "q_proj": g_q32, "k_proj": g_k32, "v_proj": g_v32, "o_proj": g_o32, "norm": dg2s}

        # ---- update (train.py:285-296) ----
        lr = linear_warmup_lr(self.lr, self.t * self.sync.world, self.warmup, self.total)     # AcceleratedScheduler: world steps per update
        self.t += 1
        inv_scale = 1.0 / (self.loss_scale * self.grad_accum)
        # A gradient that is not finite (an fp16 overflow under the constant loss scale) must not reach the moments: the
        # reference's GradScaler drops such a step and its schedule step.  The gradients are the all-reduced ones, so every
        # rank sees the same flag; the update kernels read it on the device, the host learns of it with the step's sums.
        bad = torch.zeros(1, dtype=torch.int32, device=dev)
        for g in grads.values():
            _lib.check(self.lib.kg_train_nonfinite(self._p(g), g.numel(), self._p(bad), self._st()), "kg_train_nonfinite")
        for key, g in grads.items():
            p16 = self.w16.get(key)
            _lib.check(self.lib.kg_train_adamw(self._p(self.p[key]), self._p(self.m[key]), self._p(self.v[key]), self._p(g), 2, inv_scale, self.clip, lr,
                                               self.betas[0], self.betas[1], self.adam_eps, self.wd, self.t, self.p[key].numel(), self._p(p16), self.dt,
                                               self._p(bad), self._st()), "kg_train_adamw")
        self._last = (grads, inv_scale)
        sums = torch.stack([row_loss.sum(), row_acc.sum()] + [((row_rank >= 0) & (row_rank < k)).sum().float() for k in (1, 2, 3)] + [bad[0].float()]).tolist()
        skipped = sums[5] != 0.0
        if skipped:
            self.t -= 1                                                              # neither the optimizer nor the schedule advanced
        if sums[0] != sums[0]:
            raise FloatingPointError("the loss is NaN (train.py:288-293 stops here too)")
        # top-k counts as train.py:298-309 accumulates them (the target token = argmax of the target head)
        return {"loss": sums[0] / n_loss, "prob_acc": sums[1] / n_loss, "lr": lr, "tokens": B * S0, "loss_positions": n_loss,
                "top1_correct": int(sums[2]), "top2_correct": int(sums[3]), "top3_correct": int(sums[4]), "skipped": skipped}

    @property
    def last_grads(self) -> Dict[str, torch.Tensor]:
        """The last step's gradients, unscaled, before clipping (tests, logging); computed when asked for."""
        grads, inv_scale = self._last
        return {k: g * inv_scale for k, g in grads.items()}

    def state_dict(self, dtype=None) -> Dict[str, torch.Tensor]:
        return {PARAM_KEYS[k]: (v.to(dtype) if dtype is not None else v.clone()) for k, v in self.p.items()}

    def save_state(self, path: str) -> None:
        """Everything a later call needs to continue (the role of accelerator.save_state, train.py:335): weights,
        both AdamW moments and the step count, as one torch.save file of plain tensors."""
        torch.save({"step": self.t, "p": {k: v.cpu() for k, v in self.p.items()}, "m": {k: v.cpu() for k, v in self.m.items()},
                    "v": {k: v.cpu() for k, v in self.v.items()}}, path)

    def load_state(self, path: str) -> None:
        """accelerator.load_state (train.py:245-246) for a file written by save_state."""
        st = torch.load(path, map_location="cpu", weights_only=True)
        self.t = int(st["step"])
        for k in self.p:
            self.p[k].copy_(st["p"][k]); self.m[k].copy_(st["m"][k]); self.v[k].copy_(st["v"][k])
            if k in self.w16:
                self.w16[k].copy_(self.p[k])


def list_files(path: str) -> List[str]:
    """train.py:92-98."""
    import os
    out = []
    for root, _, files in os.walk(path):
        out += [os.path.join(root, f) for f in files]
    return out


def epoch_batches(n_files: int, bs: int, seed: int, rank: int, world: int) -> List[List[int]]:
    """File indices of every batch rank `rank` trains on in one epoch: the epoch's shuffled order is cut into batches
    of `bs` (the last one may be short: DataLoader(drop_last=False), train.py:217), batch i goes to rank i % world, and
    when the batch count is not a multiple of `world` the order wraps round to its start until it is, as accelerate's
    sharded loader does (even_batches): EVERY rank runs the same number of steps, which the gradient all-reduce of
    every step needs.  One rank gets every batch unchanged."""
    if n_files < 1 or bs < 1:
        return []
    g = torch.Generator().manual_seed(seed)
    order = torch.randperm(n_files, generator=g).tolist()
    batches = [order[i:i + bs] for i in range(0, n_files, bs)]
    if world > 1:
        j = bs - len(batches[-1])                          # fill the short batch, then add whole batches, from the start of the order
        batches[-1] = batches[-1] + [order[k % n_files] for k in range(j)]
        while len(batches) % world:
            batches.append([order[(j + k) % n_files] for k in range(bs)])
            j += bs
    return batches[rank::world]


def train_epoch(trainer: AdapterTrainer, files: Sequence[str], exit_layer, bs: int, max_len: int = 2048, seed: int = 0,
                rank: int = 0, world: int = 1, log_steps: int = 20, log=print) -> Dict[str, float]:
    """One epoch of train.py:250-333 over hidden-state files written by the reference's data generation
    (each a torch.save of hidden_state / hidden_state_layer<L> / input_ids / loss_mask): shuffled batches of `bs`
    files, every rank taking its own share of each epoch's order (DataLoader(shuffle=True) + accelerate's
    sharding, see epoch_batches), the totals the reference prints at the end of the epoch summed over the ranks
    (accelerator.gather_for_metrics, train.py:322-329)."""
    collate = DataCollatorWithPadding()
    tot = {"loss": 0.0, "batches": 0, "positions": 0, "top1": 0, "top2": 0, "top3": 0}
    for idx in epoch_batches(len(files), bs, seed, rank, world):
        feats = [prepare_sample(torch.load(files[i], map_location="cpu", weights_only=True), exit_layer, max_len) for i in idx]
        out = trainer.step(collate(feats))
        tot["loss"] += out["loss"]; tot["batches"] += 1; tot["positions"] += out["loss_positions"]
        for k in (1, 2, 3):
            tot[f"top{k}"] += out[f"top{k}_correct"]
        if rank == 0 and tot["batches"] % log_steps == 1:
            log(f"Step: {tot['batches'] - 1}\tLR: {out['lr']}\tAccept: {out['prob_acc']}\tLoss: {out['loss']}")
    if world > 1:
        keys = ("loss", "batches", "positions", "top1", "top2", "top3")
        t = torch.tensor([float(tot[k]) for k in keys], dtype=torch.float64, device=trainer.device)
        trainer.sync.dist.all_reduce(t, group=trainer.sync.group)
        tot = {k: float(v) for k, v in zip(keys, t.tolist())}
    n = max(1, tot["positions"])
    return {"loss": tot["loss"] / max(1, tot["batches"]), "accuracy": tot["top1"] / n, "top2": tot["top2"] / n, "top3": tot["top3"] / n, "batches": tot["batches"]}


def main(argv=None) -> None:
    """The command line of train.py (:2-13) for one epoch (`--start` = epoch index, as start_train.py:29-41 calls it
    once per epoch); run under torchrun for one process per GPU."""
    import argparse, json, os
    from .weights import load_base_state_dict
    ap = argparse.ArgumentParser(description="adapter training, one epoch")
    ap.add_argument("--basepath", required=True); ap.add_argument("--configpath", required=True)
    ap.add_argument("--lr", type=float, default=3e-5); ap.add_argument("--bs", type=int, default=4); ap.add_argument("--exit_layer", default="2")
    ap.add_argument("--tmpdir", required=True, help="directory of hidden-state files"); ap.add_argument("--cpdir", required=True); ap.add_argument("--start", type=int, default=0)
    ap.add_argument("--dtype", default="float16")
    a = ap.parse_args(argv)
    world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl")
    shape = LlamaShape.from_hf_config(a.configpath)
    head = load_base_state_dict(a.basepath, only=("lm_head.weight",))["lm_head.weight"]
    os.makedirs(a.cpdir, exist_ok=True)
    prev = os.path.join(a.cpdir, f"state_{a.start - 1}.pt")
    g = torch.Generator().manual_seed(0)
    d = shape.hidden_size
    init = {PARAM_KEYS["q_proj"]: torch.nn.init.kaiming_uniform_(torch.empty(shape.q_dim, d), a=5 ** 0.5, generator=g),
            PARAM_KEYS["k_proj"]: torch.nn.init.kaiming_uniform_(torch.empty(shape.kv_dim, d), a=5 ** 0.5, generator=g),
            PARAM_KEYS["v_proj"]: torch.nn.init.kaiming_uniform_(torch.empty(shape.kv_dim, d), a=5 ** 0.5, generator=g),
            PARAM_KEYS["o_proj"]: torch.nn.init.kaiming_uniform_(torch.empty(d, shape.q_dim), a=5 ** 0.5, generator=g),
            PARAM_KEYS["input_layernorm"]: torch.ones(d), PARAM_KEYS["norm"]: torch.ones(d)}       # nn.Linear / LlamaRMSNorm defaults
    tr = AdapterTrainer(shape, init, head, dtype=getattr(torch, a.dtype), lr=a.lr, num_warmup_steps=2000, total_steps=800000)
    if a.start > 0:
        tr.load_state(prev)
    files = sorted(list_files(a.tmpdir))
    files = files[:int(len(files) * 0.95)]                 # train.py:207
    res = train_epoch(tr, files, a.exit_layer, a.bs, seed=a.start, rank=rank, world=world)
    if rank == 0:
        print(json.dumps({"epoch": a.start, **res}))
        tr.save_state(os.path.join(a.cpdir, f"state_{a.start}.pt"))
        torch.save(tr.state_dict(torch.float16), os.path.join(a.cpdir, "pytorch_model.bin"))


if __name__ == "__main__":
    main()
