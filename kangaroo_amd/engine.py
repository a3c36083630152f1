"""Device engine: packs weights, owns the HBM-resident buffers (PyTorch allocates them)
and calls libkangaroo_hip.so through its C ABI.

Everything the decode path touches stays resident on the GPU: packed weights, the
in-place KV caches [layers, Hkv, t_max, hd] of the base model and the adapter, and one
workspace for activations, loop state and the token buffer.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib
from .config import LlamaShape
from .packing import pack_weight, pack_qkv, pack_gate_up, rope_tables


def _dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float16:
        return _lib.DT_F16
    if dtype == torch.bfloat16:
        return _lib.DT_BF16
    raise ValueError(f"the HIP decode path stores weights in float16 or bfloat16, not {dtype}")


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class GenStats:
    n_tokens: int
    new_tokens: int
    rounds: int
    drafted_total: int
    stopped_on_eos: bool
    prefill_ms: float
    decode_ms: float
    accept_lengths: List[int]
    draft_lengths: List[int]


class KangarooEngine:
    """Packed weights + caches + session for one (base model, adapter) pair on one GPU."""

    def __init__(self, shape: LlamaShape, base_sd: Dict[str, torch.Tensor], adapter_sd: Dict[str, torch.Tensor],
                 exit_layer: int = 2, dtype: torch.dtype = torch.float16, device="cuda", max_length: int = 2048,
                 release_source: bool = False, tp_plan=None):
        """tp_plan (kangaroo_amd.tp.TPPlan): the layers [exit_layer, N) are held as tensor-parallel shards (this
        rank's, or every rank's to rehearse them on one GPU) and every pass of the session exchanges their partial
        sums through the library's one-shot exchange kernel; the draft stack, adapter and head stay whole."""
        if not torch.cuda.is_available():
            raise _lib.KangarooHipError("no GPU visible: the Kangaroo decode path has no CPU fallback")
        self.lib = _lib.lib()
        self.shape = shape
        self.exit_layer = exit_layer
        self.dtype = dtype
        self.device = torch.device(device)
        self.max_length = max_length
        self._handle = C.c_void_p(0)
        dev = self.device

        def get(sd, k):
            t = sd[k]
            if release_source:
                sd[k] = None
            return t.to(device=dev, dtype=dtype)

        H, Hkv, hd = shape.num_attention_heads, shape.num_key_value_heads, shape.head_dim
        self._keep: List[torch.Tensor] = []     # every tensor whose address the session holds

        def keep(t):
            self._keep.append(t)
            return t

        self.tp_plan = tp_plan
        n_whole = exit_layer if tp_plan is not None else shape.num_hidden_layers        # sharded layers are packed below
        layers = (_lib.KgLayerWeights * shape.num_hidden_layers)()
        for i in range(n_whole):
            p = f"model.layers.{i}."
            lw = layers[i]
            lw.wqkv = keep(pack_qkv(get(base_sd, p + "self_attn.q_proj.weight"), get(base_sd, p + "self_attn.k_proj.weight"),
                                    get(base_sd, p + "self_attn.v_proj.weight"), H, Hkv, hd)).data_ptr()
            lw.wo = keep(pack_weight(get(base_sd, p + "self_attn.o_proj.weight"))).data_ptr()
            lw.wgu = keep(pack_gate_up(get(base_sd, p + "mlp.gate_proj.weight"), get(base_sd, p + "mlp.up_proj.weight"))).data_ptr()
            lw.wdown = keep(pack_weight(get(base_sd, p + "mlp.down_proj.weight"))).data_ptr()
            lw.norm1 = keep(get(base_sd, p + "input_layernorm.weight").contiguous()).data_ptr()
            lw.norm2 = keep(get(base_sd, p + "post_attention_layernorm.weight").contiguous()).data_ptr()
        self._layers = layers
        w = _lib.KgWeights()
        self.embed = keep(get(base_sd, "model.embed_tokens.weight").contiguous())
        w.embed = self.embed.data_ptr()
        w.final_norm = keep(get(base_sd, "model.norm.weight").contiguous()).data_ptr()
        w.lm_head = keep(pack_weight(get(base_sd, "lm_head.weight"))).data_ptr()
        w.layers = layers
        a = "layers.0."
        w.adapter.wqkv = keep(pack_qkv(get(adapter_sd, a + "self_attn.q_proj.weight"), get(adapter_sd, a + "self_attn.k_proj.weight"),
                                       get(adapter_sd, a + "self_attn.v_proj.weight"), H, Hkv, hd)).data_ptr()
        w.adapter.wo = keep(pack_weight(get(adapter_sd, a + "self_attn.o_proj.weight"))).data_ptr()
        w.adapter.wgu = None
        w.adapter.wdown = None
        w.adapter.norm1 = keep(get(adapter_sd, a + "input_layernorm.weight").contiguous()).data_ptr()
        w.adapter.norm2 = None
        w.adapter_norm = keep(get(adapter_sd, "norm.weight").contiguous()).data_ptr()
        cos, sin = rope_tables(hd, shape.max_position_embeddings, shape.rope_base, device=dev)
        w.rope_cos = keep(cos).data_ptr()
        w.rope_sin = keep(sin).data_ptr()
        self._weights = w

        ks = _lib.KgShape(_dtype_code(dtype), shape.hidden_size, shape.num_hidden_layers, H, Hkv, hd,
                          shape.intermediate_size, shape.vocab_size, shape.rms_norm_eps,
                          shape.max_position_embeddings, exit_layer)
        self._kshape = ks
        t_max = (max_length + 31) // 32 * 32
        # V is cached transposed ([.., hd, t_max]) when the MFMA attention kernel applies
        self.v_transposed = hd >= 32
        self.t_max = t_max
        self.k_cache = torch.zeros((n_whole, Hkv, t_max, hd), dtype=dtype, device=dev)
        self.ak_cache = torch.zeros((Hkv, t_max, hd), dtype=dtype, device=dev)
        if self.v_transposed:
            self.v_cache = torch.zeros((n_whole, Hkv, hd, t_max), dtype=dtype, device=dev)
            self.av_cache = torch.zeros((Hkv, hd, t_max), dtype=dtype, device=dev)
        else:
            self.v_cache = torch.zeros_like(self.k_cache)
            self.av_cache = torch.zeros_like(self.ak_cache)
        wbytes = self.lib.kg_workspace_bytes(C.byref(ks), t_max)
        self.workspace = torch.zeros(wbytes, dtype=torch.uint8, device=dev)
        b = _lib.KgBuffers(self.k_cache.data_ptr(), self.v_cache.data_ptr(), self.ak_cache.data_ptr(), self.av_cache.data_ptr(),
                           self.workspace.data_ptr(), wbytes, t_max)
        self._buffers = b
        h = C.c_void_p(0)
        _lib.check(self.lib.kg_session_create(C.byref(ks), C.byref(w), C.byref(b), C.byref(h)), "kg_session_create")
        self._handle = h
        assert self.lib.kg_session_get(h, b"v_transposed") == int(self.v_transposed)
        self._xbuf = None
        self._peer_maps: List[int] = []
        if tp_plan is not None:
            self._setup_tp(base_sd, tp_plan)
        self.tokens = self._region("tokens", t_max * 4).view(torch.int32)
        self.hs_early = self._region("hs_early", t_max * shape.hidden_size * 2).view(dtype).view(t_max, shape.hidden_size)

    # ------------------------------------------------------------------
    def _setup_tp(self, base_sd, plan):
        """Shard, pack and register the remaining layers; create this GPU's exchange buffer and map the peers'."""
        from .tp import padded_heads, shard_layer
        shape, dev, dtype, L = self.shape, self.device, self.dtype, self.exit_layer
        hd, tp = shape.head_dim, plan.size
        hl, hkvl, fl = padded_heads(shape.num_attention_heads, tp) // tp, padded_heads(shape.num_key_value_heads, tp) // tp, shape.intermediate_size // tp
        n_deep = shape.num_hidden_layers - L
        cfg = _lib.KgTpConfig()
        cfg.tp_size, cfg.n_local, cfg.heads_local, cfg.kv_heads_local, cfg.ffn_local = tp, len(plan.local_ranks), hl, hkvl, fl
        self.tp_k, self.tp_v, self._tp_layers = [], [], []
        for j, r in enumerate(plan.local_ranks):
            arr = (_lib.KgLayerWeights * n_deep)()
            for i in range(n_deep):
                s = {k: v.to(device=dev, dtype=dtype) for k, v in shard_layer(base_sd, f"model.layers.{L + i}.", shape, tp, r).items()}
                lw = arr[i]
                keep = self._keep.append
                t = pack_qkv(s["wq"], s["wk"], s["wv"], hl, hkvl, hd); keep(t); lw.wqkv = t.data_ptr()
                t = pack_weight(s["wo"].contiguous()); keep(t); lw.wo = t.data_ptr()
                t = pack_gate_up(s["wg"].contiguous(), s["wu"].contiguous()); keep(t); lw.wgu = t.data_ptr()
                t = pack_weight(s["wd"].contiguous()); keep(t); lw.wdown = t.data_ptr()
                t = s["norm1"].contiguous(); keep(t); lw.norm1 = t.data_ptr()
                t = s["norm2"].contiguous(); keep(t); lw.norm2 = t.data_ptr()
            self._tp_layers.append(arr)
            k = torch.zeros((n_deep, hkvl, self.t_max, hd), dtype=dtype, device=dev)
            v = torch.zeros((n_deep, hkvl, hd, self.t_max) if self.v_transposed else (n_deep, hkvl, self.t_max, hd), dtype=dtype, device=dev)
            self.tp_k.append(k); self.tp_v.append(v)
            cfg.local_rank[j] = r
            cfg.layers[j] = arr
            cfg.k_cache[j], cfg.v_cache[j] = k.data_ptr(), v.data_ptr()
        nbytes = self.lib.kg_tp_xbuf_bytes(tp, shape.hidden_size)
        handle = C.create_string_buffer(64)
        xb = C.c_void_p(0)
        with torch.cuda.device(dev):
            _lib.check(self.lib.kg_tp_xbuf_create(nbytes, C.byref(xb), handle), "kg_tp_xbuf_create")
        self._xbuf, self.xbuf_bytes = xb.value, nbytes
        cfg.xbuf = xb.value
        if len(plan.local_ranks) < tp:
            handles = plan.exchange_handles(bytes(handle.raw))          # every rank's handle, by rank
            for r in range(tp):
                if r in plan.local_ranks:
                    continue
                m = C.c_void_p(0)
                with torch.cuda.device(dev):
                    _lib.check(self.lib.kg_tp_xbuf_open(C.create_string_buffer(handles[r], 64), C.byref(m)), "kg_tp_xbuf_open")
                self._peer_maps.append(m.value)
                cfg.peer_xbuf[r] = m.value
        self._tp_cfg = cfg
        _lib.check(self.lib.kg_session_set_tp(self._handle, C.byref(cfg)), "kg_session_set_tp")

    def tp_partials(self, exchange: int) -> torch.Tensor:
        """The fp32 partial sums [tp, rows_cap, d] the ranks left in this GPU's exchange buffer for exchange index
        `exchange` of the last pass (2 * layer for o_proj, + 1 for down_proj): a copy, for tests."""
        tp, d, cap = self.tp_plan.size, self.shape.hidden_size, 128
        off = (exchange & 1) * tp * cap * d * 4

        class _Raw:         # the library owns this memory: wrap the address, copy out
            __cuda_array_interface__ = {"shape": (tp, cap, d), "typestr": "<f4", "data": (self._xbuf + off, False), "version": 2}

        torch.cuda.synchronize(self.device)
        return torch.as_tensor(_Raw(), device=self.device).clone()

    def _region(self, name: str, nbytes: int) -> torch.Tensor:
        p = self.lib.kg_session_ptr(self._handle, name.encode())
        if not p:
            raise _lib.KangarooHipError(f"unknown workspace region {name}")
        off = p - self.workspace.data_ptr()
        return self.workspace[off:off + nbytes]

    def kv_views(self, layer: int, length: int):
        """(k, v) of one base layer as [1, Hkv, length, hd] views of the in-place caches (a sharded layer: the heads
        of this process's first shard)."""
        kc, vc, i = self.k_cache, self.v_cache, layer
        if self.tp_plan is not None and layer >= self.exit_layer:
            kc, vc, i = self.tp_k[0], self.tp_v[0], layer - self.exit_layer
        k = kc[i, :, :length, :].unsqueeze(0)
        v = vc[i, :, :, :length].transpose(-1, -2) if self.v_transposed else vc[i, :, :length, :]
        return k, v.unsqueeze(0)

    def adapter_kv_views(self, length: int):
        k = self.ak_cache[:, :length, :].unsqueeze(0)
        v = self.av_cache[:, :, :length].transpose(-1, -2) if self.v_transposed else self.av_cache[:, :length, :]
        return k, v.unsqueeze(0)

    def close(self):
        if self._handle:
            self.lib.kg_session_destroy(self._handle)
            self._handle = C.c_void_p(0)
            for m in self._peer_maps:
                self.lib.kg_tp_xbuf_close(C.c_void_p(m))
            self._peer_maps = []
            if self._xbuf:
                self.lib.kg_tp_xbuf_destroy(C.c_void_p(self._xbuf))
                self._xbuf = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tuning(self, **kw):
        for k, v in kw.items():
            _lib.check(self.lib.kg_session_set(self._handle, k.encode(), int(v)), f"kg_session_set({k})")

    def get_tuning(self, key: str) -> int:
        return int(self.lib.kg_session_get(self._handle, key.encode()))

    def weight_bytes(self) -> Dict[str, int]:
        s = self.shape
        return {"layer": s.layer_weight_bytes(), "adapter": s.attn_only_weight_bytes(), "head": s.head_weight_bytes(),
                "kv_per_pos_layer": s.kv_bytes_per_position()}

    # ---- passes (reference-interface mirror classes call these) -------------------
    def forward_layers(self, layer_begin: int, layer_end: int, x_in: Optional[torch.Tensor], pos0: int, n_rows: int) -> torch.Tensor:
        out = torch.empty((n_rows, self.shape.hidden_size), dtype=self.dtype, device=self.device)
        if x_in is not None:
            x_in = x_in.reshape(n_rows, self.shape.hidden_size).to(self.dtype).contiguous()
        _lib.check(self.lib.kg_forward_layers(self._handle, layer_begin, layer_end, _ptr(x_in), pos0, n_rows, _ptr(out), _stream()),
                   "kg_forward_layers")
        return out

    def whole_prompt_rows(self) -> int:
        """Prompts of at least this many tokens go through the layers as a whole (0: this session cannot)."""
        return 0 if self.tp_plan is not None else self.get_tuning("rows_prefill_min")

    def prefill_rows(self, n_tokens: int, all_logits: bool):
        """kg_prefill_rows: (hidden_states[exit_layer], hidden_states[-1], logits of every row or None) for tokens [0, n)."""
        d = self.shape.hidden_size
        early = torch.empty((n_tokens, d), dtype=self.dtype, device=self.device)
        final = torch.empty((n_tokens, d), dtype=self.dtype, device=self.device)
        logits = torch.empty((n_tokens, self.shape.vocab_size), dtype=self.dtype, device=self.device) if all_logits else None
        _lib.check(self.lib.kg_prefill_rows(self._handle, n_tokens, _ptr(early), _ptr(final), _ptr(logits), _stream()), "kg_prefill_rows")
        return early, final, logits

    def adapter_forward_rows(self, x_in: torch.Tensor, n_rows: int) -> torch.Tensor:
        x_in = x_in.reshape(n_rows, self.shape.hidden_size).to(self.dtype).contiguous()
        out = torch.empty_like(x_in)
        _lib.check(self.lib.kg_adapter_forward_rows(self._handle, _ptr(x_in), n_rows, _ptr(out), _stream()), "kg_adapter_forward_rows")
        return out

    def final_norm(self, x: torch.Tensor) -> torch.Tensor:
        n = x.shape[0]
        x = x.contiguous()
        out = torch.empty_like(x)
        _lib.check(self.lib.kg_final_norm(self._handle, _ptr(x), n, _ptr(out), _stream()), "kg_final_norm")
        return out

    def adapter_forward(self, x_in: torch.Tensor, pos0: int, n_rows: int) -> torch.Tensor:
        x_in = x_in.reshape(n_rows, self.shape.hidden_size).to(self.dtype).contiguous()
        out = torch.empty_like(x_in)
        _lib.check(self.lib.kg_adapter_forward(self._handle, _ptr(x_in), pos0, n_rows, _ptr(out), _stream()), "kg_adapter_forward")
        return out

    def head_logits(self, x: torch.Tensor, want_stats: bool = False, verify_arithmetic: bool = False):
        """verify_arithmetic: a single row goes through the multi-row (MFMA) kernel, bit-identical to that row inside a verify pass."""
        n = x.shape[0]
        nr = -1 if (verify_arithmetic and n == 1) else n
        x = x.to(self.dtype).contiguous()
        logits = torch.empty((n, self.shape.vocab_size), dtype=self.dtype, device=self.device)
        if want_stats:
            am = torch.empty(n, dtype=torch.int32, device=self.device)
            pm = torch.empty(n, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.kg_head_logits(self._handle, _ptr(x), nr, _ptr(logits), _ptr(am), _ptr(pm), _stream()), "kg_head_logits")
            return logits, am, pm
        _lib.check(self.lib.kg_head_logits(self._handle, _ptr(x), nr, _ptr(logits), None, None, _stream()), "kg_head_logits")
        return logits

    def set_tokens(self, ids: torch.Tensor, pos0: int):
        ids = ids.reshape(-1)
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.shape.vocab_size):
            raise ValueError("token id out of range")          # the embedding gather has no bounds check on the device
        self.tokens[pos0:pos0 + ids.numel()] = ids.to(device=self.device, dtype=torch.int32)

    def bench_op(self, op: str, rows: int, iters: int = 64):
        """Mean microseconds per launch of one linear kernel of the path (HIP events on the
        launching stream) and its weight bytes.  op: qkv | o | gateup | down | head."""
        code = {"qkv": 0, "o": 1, "gateup": 2, "down": 3, "head": 4}[op]
        us = C.c_float(0)
        nbytes = C.c_size_t(0)
        torch.cuda.synchronize()
        _lib.check(self.lib.kg_bench_op(self._handle, code, rows, iters, C.byref(us), C.byref(nbytes)), "kg_bench_op")
        return us.value, nbytes.value

    # ---- fused loops ---------------------------------------------------------------
    def generate(self, prompt_ids, max_new_tokens: int, mode: str = "kangaroo", steps: int = 6, threshold: float = 0.6,
                 eos_token_id: int = 2, max_length: Optional[int] = None, accept_top_k: int = 1,
                 accept_top_p: Optional[float] = None, temperature: float = 1.0, accept_typical=None):
        """Whole kangaroo_forward (mode='kangaroo') or plain greedy decoding (mode='vanilla')
        on the device.  accept_top_k > 1: top-k verification (a drafted token is kept while it is
        among the k largest verify logits; evaluation/inference_kangaroo_MTBench.py:377-389) instead
        of the greedy rule.  accept_top_p: top-p verification on softmax(logits / temperature)
        (a drafted token is kept while it lies in the pooled nuclei of the round's rows, :356-374).
        accept_typical = (epsilon, delta): the "typical" rule at `temperature` (:293-314), reproduced as the
        reference executes it (start_index moves without a token being written; the draft threshold is rebound).
        Returns (output_ids list, GenStats)."""
        if mode == "kangaroo":
            if self.get_tuning("accept_top_k") != int(accept_top_k):
                self.set_tuning(accept_top_k=int(accept_top_k))
            want_p = C.c_float(float(accept_top_p) if accept_top_p else 0.0).value
            if self.lib.kg_session_getf(self._handle, b"accept_top_p") != want_p or \
                    (want_p > 0 and self.lib.kg_session_getf(self._handle, b"temperature") != C.c_float(temperature).value):
                _lib.check(self.lib.kg_session_setf(self._handle, b"temperature", float(temperature)), "kg_session_setf(temperature)")
                _lib.check(self.lib.kg_session_setf(self._handle, b"accept_top_p", want_p), "kg_session_setf(accept_top_p)")
            want_t = 2 if accept_typical else 1
            if self.get_tuning("accept_typical") != want_t - 1 or accept_typical:
                if accept_typical:
                    _lib.check(self.lib.kg_session_setf(self._handle, b"temperature", float(temperature)), "kg_session_setf(temperature)")
                    _lib.check(self.lib.kg_session_setf(self._handle, b"typical_epsilon", float(accept_typical[0])), "kg_session_setf")
                    _lib.check(self.lib.kg_session_setf(self._handle, b"typical_delta", float(accept_typical[1])), "kg_session_setf")
                self.set_tuning(accept_typical=want_t)
        if isinstance(prompt_ids, torch.Tensor):
            prompt = prompt_ids.reshape(-1).to("cpu", torch.int32).contiguous()
        else:
            prompt = torch.tensor(list(prompt_ids), dtype=torch.int32)
        T0 = prompt.numel()
        max_length = max_length or self.max_length
        out = torch.empty(max_length + 8, dtype=torch.int32)
        cap = max_length + 8
        log = torch.zeros(2 * cap, dtype=torch.int32)
        res = _lib.KgGenResult()
        i32p = C.POINTER(C.c_int32)
        rc = self.lib.kg_generate(self._handle, 0 if mode == "kangaroo" else 1,
                                  C.cast(prompt.data_ptr(), i32p), T0, max_new_tokens, max_length, steps, float(threshold),
                                  eos_token_id, C.cast(out.data_ptr(), i32p), C.cast(log.data_ptr(), i32p), cap,
                                  C.byref(res), _stream())
        _lib.check(rc, "kg_generate")
        ids = out[:res.n_tokens].tolist()
        rounds = res.rounds if mode == "kangaroo" else 0
        acc = log[0:2 * rounds:2].tolist()
        drafted = log[1:2 * rounds:2].tolist()
        return ids, GenStats(res.n_tokens, res.new_tokens, res.rounds, res.drafted_total, bool(res.stopped_on_eos),
                             res.prefill_ms, res.decode_ms, acc, drafted)
