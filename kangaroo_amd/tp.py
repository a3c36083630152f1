"""Tensor-parallel verification of the remaining layers (the >= 33B-class configuration).

The reference places layers on GPUs one after the other (`device_map="auto"`, kangaroo/kangaroo_model.py:22);
here the layers [EARLY_STOP_LAYER, N) that the verify pass of kangaroo/earlyexit.py:51-69 runs are split
Megatron-style over `tp` ranks, one process per GPU:

  q/k/v and gate/up projections   split by output rows (whole heads / slices of the MLP width): no exchange
  attention                       each rank works on its own heads and keeps their K/V cache rows
  o_proj and down_proj            split by input columns: every rank produces fp32 PARTIAL sums [rows, d]
                                  (epilogue 5 of kg_linear), which are added up across ranks

The draft stack (layers [0, L), adapter, lm_head) is replicated on every rank, and so is the accept /
reject decision: every rank sees identical numbers, so no broadcast is needed and the ranks stay in step.

The exchange is a one-shot all-reduce: every rank's partial goes to every peer (`all_gather`, <= tp x 7 rows
x d fp32, a message of under 200 KB that the fully connected xGMI links carry in one hop) and each rank adds
the partials IN RANK ORDER.  That makes the sum bit-identical on all ranks, independent of the transport, and
identical to running the shards one after the other on a single GPU (`LocalGroup`), which is how the parity
tests exercise this file where only one GPU is at hand.

The arithmetic of a layer equals the unsplit kernels' except for the order of the fp32 additions over K:
h_mid = round16(h + round16(sum_r partial_r)), as the residual epilogue computes round16(h + round16(y)).

Two ways to run it:

  NativeTPKangarooModel   the library holds the shards (kg_session_set_tp): every pass of the session, the fused
                          device-side loop (kg_generate: prefill in 128-row GEMM passes, one captured graph per round)
                          included, runs 5 kernels per shard and layer plus one exchange kernel per row-split
                          projection, which pushes the partial into the peers' exchange buffers over the direct
                          links, waits for theirs and adds them up in rank order (csrc/tp.hip).  No host step and
                          no library collective inside a round.
  TPKangarooModel         the same partition driven pass by pass from Python over the kernel-level entry points, the
                          exchange a torch.distributed all_gather (RCCL) captured with the kernels in one graph per
                          row count: the comparison point for the exchange kernel, and the path for process groups
                          whose ranks cannot map each other's memory.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib
from .config import LlamaShape
from .earlyexit import EarlyExitLlamaForCausalLM, _check_positions
from .engine import KangarooEngine, _dtype_code
from .packing import pack_gate_up, pack_qkv, pack_weight, rope_tables

EPI_STORE_F32 = 5
EPI_SILU_MUL = 3


# ---------------------------------------------------------------------------------------------------
# partitioning (pure tensor maths: runs on CPU, tested with gloo)
# ---------------------------------------------------------------------------------------------------
def padded_heads(n_heads: int, tp: int) -> int:
    """Heads per rank x tp: LLaMA-33B has 52 heads, which 8 ranks share as 7 each with 4 all-zero heads."""
    return (n_heads + tp - 1) // tp * tp


def shard_layer(sd: Dict[str, torch.Tensor], prefix: str, shape: LlamaShape, tp: int, rank: int) -> Dict[str, torch.Tensor]:
    """Rank `rank`'s slice of one decoder layer.  Heads are dealt out in contiguous blocks; a head count that
    `tp` does not divide is padded with zero heads (zero q/k/v rows give a uniform attention over zero values,
    and their o_proj columns are zero, so they add nothing).  The MLP width must divide evenly."""
    H, Hkv, hd, d, F = shape.num_attention_heads, shape.num_key_value_heads, shape.head_dim, shape.hidden_size, shape.intermediate_size
    if H != Hkv and (H % tp or Hkv % tp):
        raise ValueError("grouped-query attention needs tp to divide both head counts")
    if F % tp or (F // tp) % 32:
        raise ValueError("intermediate_size / tp must be a multiple of 32")
    Hp, Hkvp = padded_heads(H, tp), padded_heads(Hkv, tp)
    hl, hkvl, fl = Hp // tp, Hkvp // tp, F // tp

    def rows(w, n_heads_real, per_rank):            # [n_heads*hd, d] -> this rank's heads, zero rows past the real ones
        lo, hi = rank * per_rank, (rank + 1) * per_rank
        out = w.new_zeros((per_rank * hd, w.shape[1]))
        real_hi = min(hi, n_heads_real)
        if real_hi > lo:
            out[:(real_hi - lo) * hd] = w[lo * hd:real_hi * hd]
        return out

    wo = sd[prefix + "self_attn.o_proj.weight"]
    wo_l = wo.new_zeros((d, hl * hd))
    lo, real_hi = rank * hl, min((rank + 1) * hl, H)
    if real_hi > lo:
        wo_l[:, :(real_hi - lo) * hd] = wo[:, lo * hd:real_hi * hd]
    return {
        "wq": rows(sd[prefix + "self_attn.q_proj.weight"], H, hl),
        "wk": rows(sd[prefix + "self_attn.k_proj.weight"], Hkv, hkvl),
        "wv": rows(sd[prefix + "self_attn.v_proj.weight"], Hkv, hkvl),
        "wo": wo_l,
        "wg": sd[prefix + "mlp.gate_proj.weight"][rank * fl:(rank + 1) * fl],
        "wu": sd[prefix + "mlp.up_proj.weight"][rank * fl:(rank + 1) * fl],
        "wd": sd[prefix + "mlp.down_proj.weight"][:, rank * fl:(rank + 1) * fl],
        "norm1": sd[prefix + "input_layernorm.weight"],
        "norm2": sd[prefix + "post_attention_layernorm.weight"],
    }


def ordered_sum(parts: torch.Tensor) -> torch.Tensor:
    """[tp, rows, d] fp32 partial sums -> their sum, added in rank order (one fp32 addition per rank, so every
    rank, and a single-GPU rehearsal, gets the same bits)."""
    if parts.shape[0] == 1:
        return parts[0] * 1                   # exact, and a kernel rather than a copy
    acc = parts[0] + parts[1]                 # elementwise kernels only: inside a captured graph they keep their order
    for r in range(2, parts.shape[0]):
        acc += parts[r]
    return acc


def add_residual(resid: torch.Tensor, total: torch.Tensor) -> torch.Tensor:
    """round16(resid + round16(total)): the residual epilogue's arithmetic (kangaroo/adapter.py:304 on 16-bit tensors)."""
    return (resid.float() + total.to(resid.dtype).float()).to(resid.dtype)


class TPPlan:
    """Which ranks' shards a session holds and how the ranks swap the handles of their exchange buffers.
    TPPlan.local(tp): all ranks on this GPU (single-GPU rehearsal).  TPPlan.distributed(group): this process is one
    rank of a torch.distributed group (any backend: only 64-byte handles travel through it, once, at set-up)."""

    def __init__(self, size: int, local_ranks: Sequence[int], group=None):
        self.size, self.local_ranks, self.group = size, list(local_ranks), group

    @classmethod
    def local(cls, tp: int) -> "TPPlan":
        return cls(tp, range(tp))

    @classmethod
    def distributed(cls, group=None) -> "TPPlan":
        import torch.distributed as dist
        return cls(dist.get_world_size(group), [dist.get_rank(group)], group if group is not None else True)

    def exchange_handles(self, mine: bytes) -> List[bytes]:
        import torch.distributed as dist
        group = None if self.group is True else self.group
        out = [None] * self.size
        dist.all_gather_object(out, mine, group=group)
        return out


class LocalGroup:
    """All ranks in one process, run one after the other (single-GPU rehearsal and tests)."""

    def __init__(self, tp: int):
        self.tp = tp

    def all_partials(self, mine: Sequence[torch.Tensor]) -> torch.Tensor:
        return torch.stack(list(mine), 0)


class DistGroup:
    """One rank per process over torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in
    the CPU tests).  all_partials is the one-shot exchange: every rank's partial to every rank."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.tp = dist.get_world_size(group)
        self.rank = dist.get_rank(group)

    def all_partials(self, mine: Sequence[torch.Tensor]) -> torch.Tensor:
        (part,) = mine
        out = torch.empty((self.tp,) + tuple(part.shape), dtype=part.dtype, device=part.device)
        self.dist.all_gather(list(out.unbind(0)), part.contiguous(), group=self.group)     # views of one buffer: no copy afterwards
        return out


# ---------------------------------------------------------------------------------------------------
# one rank's shard of the remaining layers on its GPU
# ---------------------------------------------------------------------------------------------------
class TPShard:
    def __init__(self, shape: LlamaShape, base_sd: Dict[str, torch.Tensor], exit_layer: int, tp: int, rank: int,
                 dtype: torch.dtype = torch.bfloat16, device="cuda", max_length: int = 2048):
        if not torch.cuda.is_available():
            raise _lib.KangarooHipError("no GPU visible: the tensor-parallel verify path has no CPU fallback")
        self.lib = _lib.lib()
        self.shape, self.tp, self.rank, self.dtype, self.device = shape, tp, rank, dtype, torch.device(device)
        self.code = _dtype_code(dtype)
        self.exit_layer = exit_layer
        hd, d = shape.head_dim, shape.hidden_size
        self.hl = padded_heads(shape.num_attention_heads, tp) // tp
        self.hkvl = padded_heads(shape.num_key_value_heads, tp) // tp
        self.fl = shape.intermediate_size // tp
        self.t_max = (max_length + 31) // 32 * 32
        self.vt = 1 if hd >= 32 else 0
        dev = self.device
        self.layers: List[Dict[str, torch.Tensor]] = []
        for l in range(exit_layer, shape.num_hidden_layers):
            s = {k: v.to(device=dev, dtype=dtype) for k, v in shard_layer(base_sd, f"model.layers.{l}.", shape, tp, rank).items()}
            self.layers.append({
                "wqkv": pack_qkv(s["wq"], s["wk"], s["wv"], self.hl, self.hkvl, hd),
                "wo": pack_weight(s["wo"].contiguous()), "wgu": pack_gate_up(s["wg"].contiguous(), s["wu"].contiguous()),
                "wdown": pack_weight(s["wd"].contiguous()), "norm1": s["norm1"].contiguous(), "norm2": s["norm2"].contiguous()})
        n = len(self.layers)
        self.k_cache = torch.zeros((n, self.hkvl, self.t_max, hd), dtype=dtype, device=dev)
        self.v_cache = torch.zeros((n, self.hkvl, hd, self.t_max) if self.vt else (n, self.hkvl, self.t_max, hd), dtype=dtype, device=dev)
        self.cos, self.sin = rope_tables(hd, shape.max_position_embeddings, shape.rope_base, device=dev)
        self.nsplit = 16
        self.scratch = torch.zeros(256, dtype=torch.uint8, device=dev)
        self.ascratch = torch.zeros(self.lib.kg_attention_scratch_bytes(self.hl, hd, self.nsplit), dtype=torch.uint8, device=dev)
        # the int32 at byte 128 of a scratch block: where a captured pass reads its first position (POS_IN_SCRATCH)
        self.pos_slots = [self.scratch[128:132].view(torch.int32), self.ascratch[128:132].view(torch.int32)]

    def _p(self, t):
        return C.c_void_p(t.data_ptr())

    def _st(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def kv_views(self, i: int, length: int):
        k = self.k_cache[i, :, :length, :].unsqueeze(0)
        v = self.v_cache[i, :, :, :length].transpose(-1, -2) if self.vt else self.v_cache[i, :, :length, :]
        return k, v.unsqueeze(0)

    def attn_partial(self, i: int, h: torch.Tensor, pos0: int) -> torch.Tensor:
        """RMSNorm + this rank's heads of q/k/v (+RoPE, cache append) + attention + its columns of o_proj -> fp32 [rows, d]."""
        n, d, hd = h.shape[0], self.shape.hidden_size, self.shape.head_dim
        L, lib = self.layers[i], self.lib
        nr = -1 if n == 1 else n            # a single row runs in a verify row's arithmetic: plain decoding over these kernels stays bit-identical
        q = torch.empty((n, self.hl * hd), dtype=self.dtype, device=self.device)
        _lib.check(lib.kg_qkv_rope_append(self.code, self._p(L["wqkv"]), d, self.hl, self.hkvl, hd, self._p(h), nr, pos0, self._p(L["norm1"]),
                                          self.shape.rms_norm_eps, self._p(self.cos), self._p(self.sin), self._p(q), self._p(self.k_cache[i]),
                                          self._p(self.v_cache[i]), self.t_max, self.vt, self._p(self.scratch), 0, 0, self._st()), "kg_qkv_rope_append")
        o = torch.empty_like(q)
        _lib.check(lib.kg_attention(self.code, self._p(q), n, pos0, self._p(self.k_cache[i]), self._p(self.v_cache[i]), self.hl, self.hkvl, hd,
                                    self.t_max, self.vt, self.nsplit, self._p(o), self._p(self.ascratch), self._st()), "kg_attention")
        part = torch.empty((n, d), dtype=torch.float32, device=self.device)
        _lib.check(lib.kg_linear(self.code, self._p(L["wo"]), d, self.hl * hd, self._p(o), nr, 0, None, 0.0, EPI_STORE_F32, None,
                                 self._p(part), d, self._p(self.scratch), 0, 0, self._st()), "kg_linear(o_proj shard)")
        return part

    def mlp_partial(self, i: int, h_mid: torch.Tensor) -> torch.Tensor:
        """RMSNorm + this rank's slice of gate/up (SiLU * mul) + its columns of down_proj -> fp32 [rows, d]."""
        n, d = h_mid.shape[0], self.shape.hidden_size
        L, lib = self.layers[i], self.lib
        nr = -1 if n == 1 else n
        act = torch.empty((n, self.fl), dtype=self.dtype, device=self.device)
        _lib.check(lib.kg_linear(self.code, self._p(L["wgu"]), 2 * self.fl, d, self._p(h_mid), nr, 1, self._p(L["norm2"]), self.shape.rms_norm_eps,
                                 EPI_SILU_MUL, None, self._p(act), self.fl, self._p(self.scratch), 0, 0, self._st()), "kg_linear(gate/up shard)")
        part = torch.empty((n, d), dtype=torch.float32, device=self.device)
        _lib.check(lib.kg_linear(self.code, self._p(L["wdown"]), d, self.fl, self._p(act), nr, 0, None, 0.0, EPI_STORE_F32, None,
                                 self._p(part), d, self._p(self.scratch), 0, 0, self._st()), "kg_linear(down_proj shard)")
        return part


def tp_forward(shards: Sequence[TPShard], group, x: torch.Tensor, pos0: int, trace: Optional[list] = None) -> torch.Tensor:
    """Layers [L, N) on rows x [rows <= 16, d] at positions pos0.. .  `shards`: this process's shard (one entry,
    with a DistGroup) or every rank's (LocalGroup).  Returns the hidden rows, identical on every rank.
    pos0 == POS_IN_SCRATCH: the kernels read the first position from the shards' position slots (TPGraphs)."""
    h = x.contiguous()
    for i in range(len(shards[0].layers)):
        parts = group.all_partials([s.attn_partial(i, h, pos0) for s in shards])
        if trace is not None:
            trace.append(parts)
        h_mid = add_residual(h, ordered_sum(parts))
        parts = group.all_partials([s.mlp_partial(i, h_mid) for s in shards])
        if trace is not None:
            trace.append(parts)
        h = add_residual(h_mid, ordered_sum(parts))
    return h


POS_IN_SCRATCH = -2     # KG_POS_IN_SCRATCH of include/kangaroo_hip.h


class TPGraphs:
    """tp_forward captured once per row count and replayed: ~9 kernel launches and two exchanges per layer and shard
    leave the host as one graph launch, so the verify pass is no longer bound by the launch rate of the interpreter.
    The first position of a pass lives in device memory (one int32 per process, copied into every shard's
    position slots by the graph itself) and is the only thing that changes between replays besides the input rows.
    With a DistGroup the all-gathers are captured with the kernels (RCCL supports stream capture); every rank
    must capture and replay the same row counts in the same order."""

    def __init__(self, shards: Sequence[TPShard], group, dtype: torch.dtype, max_rows: int = 16):
        self.shards, self.group, self.dtype, self.max_rows = list(shards), group, dtype, max_rows
        dev = self.shards[0].device
        self.pos = torch.zeros(1, dtype=torch.int32, device=dev)
        self.graphs: Dict[int, tuple] = {}
        self.pool = None

    def _body(self, x: torch.Tensor) -> torch.Tensor:
        for s in self.shards:
            for slot in s.pos_slots:
                torch.add(self.pos, 0, out=slot)      # a kernel, not a copy node (see kg_attention's counters)
        return tp_forward(self.shards, self.group, x, POS_IN_SCRATCH)

    def _capture(self, x: torch.Tensor, pos0: int):
        # Captured on the pass that first needs this row count, with its own rows and position: the eager run that
        # precedes the capture then writes this pass's true K/V rows (a dummy pass would overwrite cache rows in use),
        # and the replay that follows computes the same values again.
        n = x.shape[0]
        xs = x.clone()
        self.pos.fill_(pos0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # one eager run on a side stream first, as graph capture asks
            self._body(xs)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=self.pool):
            out = self._body(xs)
        if self.pool is None:
            self.pool = g.pool()
        self.graphs[n] = (g, xs, out)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor, pos0: int) -> torch.Tensor:
        n = x.shape[0]
        if n > self.max_rows:
            return tp_forward(self.shards, self.group, x, pos0)
        if pos0 < 0 or pos0 + n > self.shards[0].t_max:
            raise ValueError("positions outside the cache")
        if n not in self.graphs:
            err = None
            try:
                self._capture(x, pos0)
            except RuntimeError as e:                 # e.g. a process-group backend that cannot be captured
                err = e
            # Every rank must take the same path from here on (a rank replaying a graph while its peer launches the pass
            # kernel by kernel would meet it at different collectives): the ranks agree on the outcome before it is used.
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=x.device)
            if isinstance(self.group, DistGroup):
                torch.cuda.synchronize()
                self.group.dist.all_reduce(ok, op=self.group.dist.ReduceOp.MIN, group=self.group.group)
            if int(ok[0]) == 0:
                import warnings
                warnings.warn(f"tensor-parallel pass of {n} rows could not be captured on every rank ({err}); launching it kernel by kernel")
                self.max_rows = 0
                self.graphs.clear()
                torch.cuda.synchronize()
                return tp_forward(self.shards, self.group, x, pos0)
        g, xs, out = self.graphs[n]
        xs.copy_(x)
        self.pos.fill_(pos0)                          # the value travels with the fill kernel: later calls cannot overtake it
        g.replay()
        return out.clone()


class TPEarlyExitLlamaForCausalLM(EarlyExitLlamaForCausalLM):
    """EarlyExitLlamaForCausalLM (kangaroo/earlyexit.py:6-71) whose remaining layers are tensor-parallel shards.
    The engine holds the replicated draft stack only (exit_layer == its layer count)."""

    def __init__(self, engine: KangarooEngine, shards: Sequence[TPShard], group, EARLY_STOP_LAYER: int, n_layers: int, graphs: bool = True):
        super().__init__(engine, EARLY_STOP_LAYER)
        self.shards, self.group, self.n_layers = list(shards), group, n_layers
        self.graphs = TPGraphs(self.shards, group, engine.dtype) if graphs else None

    def _views(self, layer: int, length: int):
        if layer < self.early_exit_layer:
            return self.engine.kv_views(layer, length)
        return self.shards[0].kv_views(layer - self.early_exit_layer, length)

    def _deep(self, x: torch.Tensor, pos0: int) -> torch.Tensor:
        x = x.to(self.engine.dtype)
        if self.graphs is not None:
            return self.graphs(x, pos0)
        return tp_forward(self.shards, self.group, x, pos0)

    @torch.no_grad()
    def __call__(self, input_ids, position_ids=None, output_hidden_states=True, logits_rows="all"):
        from .earlyexit import BaseModelOutput, _HiddenStates
        e = self.engine
        ids = input_ids.reshape(-1)
        T = ids.numel()
        if T > e.max_length:
            raise ValueError("prompt longer than the cache")
        _check_positions(position_ids, 0, T)
        e.set_tokens(ids, 0)
        L, N, d = self.early_exit_layer, self.n_layers, e.shape.hidden_size
        Cr = self._chunk_rows()
        early = torch.empty((T, d), dtype=e.dtype, device=e.device)
        final = torch.empty((T, d), dtype=e.dtype, device=e.device)
        for c0 in range(0, T, Cr):
            n = min(Cr, T - c0)
            early[c0:c0 + n] = e.forward_layers(0, L, None, c0, n)
            final[c0:c0 + n] = e.final_norm(self._deep(early[c0:c0 + n], c0))
        rows = final if logits_rows == "all" else final[T - 1:T]
        lg = torch.cat([e.head_logits(rows[i:i + 16]) for i in range(0, rows.shape[0], 16)], 0).unsqueeze(0)
        pkv = tuple(self._views(l, T) for l in range(N))
        return BaseModelOutput(lg, _HiddenStates(N, {L: early.unsqueeze(0), N: final.unsqueeze(0)}), pkv)

    forward = __call__

    @torch.no_grad()
    def forward_draft_or_large_model(self, in_tokens_small=None, in_features_large=None, position_ids=None):
        if in_features_large is None or in_tokens_small is not None:
            out = super().forward_draft_or_large_model(in_tokens_small=in_tokens_small, in_features_large=in_features_large, position_ids=position_ids)
            return out
        assert self.past_key_values is not None
        seq_length = in_features_large.shape[1]
        past = self.past_key_values[self.early_exit_layer][0].shape[2]
        _check_positions(position_ids, past, seq_length)
        h = self._deep(in_features_large[0], past)
        for l in range(self.early_exit_layer, self.n_layers):
            self.past_key_values[l] = self._views(l, past + seq_length)
        return h.unsqueeze(0), self.engine.final_norm(h).unsqueeze(0)


class TPKangarooModel:
    """KangarooModel (kangaroo/kangaroo_model.py:12-47) with a tensor-parallel verify stack: `base_model`,
    `adapter_model`, `head_model` as there.  group = DistGroup() in a one-rank-per-GPU job, or LocalGroup(tp) to
    rehearse every rank on one GPU."""

    def __init__(self, shape: LlamaShape, base_sd, adapter_sd, dtype, EARLY_STOP_LAYER: int, group, rank: Optional[int] = None,
                 max_length: int = 2048, device="cuda", graphs: bool = True):
        from .adapter import AdapterModel
        from .kangaroo_model import HeadModel
        L = EARLY_STOP_LAYER
        draft_shape = type(shape)(**{**shape.to_dict(), "num_hidden_layers": L})
        self.engine = KangarooEngine(draft_shape, base_sd, adapter_sd, exit_layer=L, dtype=dtype, device=device, max_length=max_length)
        ranks = range(group.tp) if isinstance(group, LocalGroup) else [group.rank if rank is None else rank]
        self.shards = [TPShard(shape, base_sd, L, group.tp, r, dtype, device, max_length) for r in ranks]
        self.base_model = TPEarlyExitLlamaForCausalLM(self.engine, self.shards, group, L, shape.num_hidden_layers, graphs=graphs)
        self.adapter_model = AdapterModel(self.engine)
        self.head_model = HeadModel(self.engine)

    def eval(self):
        return self

    def forward(self):
        raise NotImplementedError


class NativeTPKangarooModel:
    """KangarooModel (kangaroo/kangaroo_model.py:12-47) whose remaining layers are tensor-parallel shards held by the
    library (kg_session_set_tp): `base_model`, `adapter_model`, `head_model` and `engine` as KangarooModel has them, so
    kangaroo_forward (fused, device-side rounds) and kangaroo_forward_stepwise both run on it unchanged."""

    def __init__(self, shape: LlamaShape, base_sd, adapter_sd, dtype, EARLY_STOP_LAYER: int, plan: TPPlan,
                 max_length: int = 2048, device="cuda"):
        from .adapter import AdapterModel
        from .kangaroo_model import HeadModel
        L = EARLY_STOP_LAYER
        self.engine = KangarooEngine(shape, base_sd, adapter_sd, exit_layer=L, dtype=dtype, device=device, max_length=max_length, tp_plan=plan)
        self.base_model = EarlyExitLlamaForCausalLM(self.engine, L)
        self.adapter_model = AdapterModel(self.engine)
        self.head_model = HeadModel(self.engine)

    def eval(self):
        return self

    def forward(self):
        raise NotImplementedError
