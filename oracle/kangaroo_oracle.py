"""CPU restatement of Kangaroo's double-early-exit greedy decode path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package may import this
module: it exists so that tests, the smoke check and the benchmark's CPU
baseline leg can compare the HIP path against an independent restatement of
the reference algorithm.  All arithmetic is plain fp32 torch on the CPU.

Each function cites the reference file:line it follows (paths relative to the
reference checkout):

* state machine ........ evaluation/inference_kangaroo.py:15-129
* split base forward ... kangaroo/earlyexit.py:14-71
* adapter .............. kangaroo/adapter.py:378-460 (+ layer classes :132-314)
* model assembly ....... kangaroo/kangaroo_model.py:22-44

Parity pinning: tests/golden/*.npz hold input/output vectors produced by the
reference's own code (see oracle/gen_golden.py, which imports the reference in
the build container); tests/test_oracle_golden.py checks this restatement
against every one of them.

`round_to` (None | torch.float16 | torch.bfloat16) optionally rounds at the op
boundaries where the reference, executed in that dtype on an accelerator,
stores a tensor of that dtype.  With None the restatement is pure fp32.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch


# --------------------------------------------------------------------------
# shapes
# --------------------------------------------------------------------------
@dataclass
class Shape:
    """LlamaConfig fields the path uses (data/vicuna_7B_config.json:8-21)."""
    hidden_size: int
    num_hidden_layers: int
    num_attention_heads: int
    num_key_value_heads: int
    intermediate_size: int
    vocab_size: int
    rms_norm_eps: float = 1e-6
    max_position_embeddings: int = 2048
    rope_base: float = 10000.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


def _rnd(x: torch.Tensor, round_to) -> torch.Tensor:
    if round_to is None:
        return x
    return x.to(round_to).to(torch.float32)


# --------------------------------------------------------------------------
# building blocks (kangaroo/adapter.py)
# --------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, round_to=None) -> torch.Tensor:
    """kangaroo/adapter.py:254-259: fp32 normalise, cast to the input dtype, then scale."""
    x32 = x.to(torch.float32)
    var = x32.pow(2).mean(-1, keepdim=True)
    y = x32 * torch.rsqrt(var + eps)
    y = _rnd(y, round_to)
    return _rnd(weight.to(torch.float32) * y, round_to)


def rope_tables(head_dim: int, max_pos: int, base: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """kangaroo/adapter.py:104,112-120: cos/sin tables [max_pos, head_dim] in fp32."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rotate_half(x: torch.Tensor) -> torch.Tensor:
    """kangaroo/adapter.py:80-84."""
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin, position_ids, round_to=None):
    """kangaroo/adapter.py:87-95.  q,k: [H, q_len, hd]; position_ids: [q_len]."""
    c = _rnd(cos, round_to)[position_ids].unsqueeze(0)
    s = _rnd(sin, round_to)[position_ids].unsqueeze(0)
    q_e = _rnd(_rnd(q * c, round_to) + _rnd(_rotate_half(q) * s, round_to), round_to)
    k_e = _rnd(_rnd(k * c, round_to) + _rnd(_rotate_half(k) * s, round_to), round_to)
    return q_e, k_e


def attention(x, position_ids, wq, wk, wv, wo, past_kv, shape: Shape, tables, round_to=None):
    """kangaroo/adapter.py:159-227 for batch 1.

    x: [q_len, d].  past_kv: None or (k [Hkv, T, hd], v [Hkv, T, hd]).
    Returns (out [q_len, d], (k, v)) with K cached after RoPE (:182-189).
    The causal mask (:38-52, :327-348) is applied as an exclusion of keys whose
    cache slot lies beyond the query's slot.
    """
    q_len = x.shape[0]
    H, Hkv, hd = shape.num_attention_heads, shape.num_key_value_heads, shape.head_dim
    q = _rnd(x @ wq.T, round_to).view(q_len, H, hd).transpose(0, 1)
    k = _rnd(x @ wk.T, round_to).view(q_len, Hkv, hd).transpose(0, 1)
    v = _rnd(x @ wv.T, round_to).view(q_len, Hkv, hd).transpose(0, 1)
    cos, sin = tables
    q, k = apply_rope(q, k, cos, sin, position_ids, round_to)
    past = 0
    if past_kv is not None:
        past = past_kv[0].shape[1]
        k = torch.cat([past_kv[0], k], dim=1)
        v = torch.cat([past_kv[1], v], dim=1)
    present = (k, v)
    n_rep = H // Hkv
    kk = k.repeat_interleave(n_rep, dim=0) if n_rep > 1 else k
    vv = v.repeat_interleave(n_rep, dim=0) if n_rep > 1 else v
    scores = _rnd(_rnd(q @ kk.transpose(1, 2), round_to) / math.sqrt(hd), round_to)
    T = kk.shape[1]
    if q_len > 1:
        # query row i sits at cache slot past+i and sees slots <= past+i
        slot = torch.arange(T).unsqueeze(0)
        lim = (past + torch.arange(q_len)).unsqueeze(1)
        scores = scores.masked_fill((slot > lim).unsqueeze(0), torch.finfo(torch.float32).min)
    p = _rnd(torch.softmax(scores, dim=-1, dtype=torch.float32), round_to)
    o = _rnd(p @ vv, round_to).transpose(0, 1).reshape(q_len, H * hd)
    return _rnd(o @ wo.T, round_to), present


def mlp(x, wg, wu, wd, round_to=None):
    """kangaroo/adapter.py:241-243: down(silu(gate(x)) * up(x))."""
    g = _rnd(x @ wg.T, round_to)
    u = _rnd(x @ wu.T, round_to)
    a = _rnd(_rnd(torch.nn.functional.silu(g), round_to) * u, round_to)
    return _rnd(a @ wd.T, round_to)


# --------------------------------------------------------------------------
# models
# --------------------------------------------------------------------------
@dataclass
class BaseOutput:
    logits: torch.Tensor
    hidden_states: Tuple[torch.Tensor, ...]
    past_key_values: Tuple[Tuple[torch.Tensor, torch.Tensor], ...]


class OracleBase:
    """Early-exit base model: kangaroo/earlyexit.py:6-71 over a Llama stack whose
    layer arithmetic is restated from kangaroo/adapter.py (attention, MLP, RMSNorm)
    with the two-half decoder layer x+attn(norm1(x)); x+mlp(norm2(x))."""

    def __init__(self, shape: Shape, weights: Dict[str, torch.Tensor], early_exit_layer: int, round_to=None):
        self.shape = shape
        self.w = {k: v.to(torch.float32) for k, v in weights.items()}
        self.early_exit_layer = early_exit_layer
        self.past_key_values: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None
        self.round_to = round_to
        self.tables = rope_tables(shape.head_dim, shape.max_position_embeddings, shape.rope_base)

    # tensors are [1, ...] at the API (reference convention), [..] inside
    def _layer(self, i: int, h: torch.Tensor, pos: torch.Tensor, past):
        p = f"model.layers.{i}."
        w, r = self.w, self.round_to
        x = rmsnorm(h, w[p + "input_layernorm.weight"], self.shape.rms_norm_eps, r)
        a, present = attention(x, pos, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.k_proj.weight"],
                               w[p + "self_attn.v_proj.weight"], w[p + "self_attn.o_proj.weight"],
                               past, self.shape, self.tables, r)
        h = _rnd(h + a, r)
        x = rmsnorm(h, w[p + "post_attention_layernorm.weight"], self.shape.rms_norm_eps, r)
        m = mlp(x, w[p + "mlp.gate_proj.weight"], w[p + "mlp.up_proj.weight"], w[p + "mlp.down_proj.weight"], r)
        return _rnd(h + m, r), present

    def __call__(self, input_ids: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                 output_hidden_states: bool = True) -> BaseOutput:
        """Prefill as the loop calls it (evaluation/inference_kangaroo.py:30-35):
        all layers, logits for every row, hidden_states[i] = input of layer i,
        hidden_states[-1] = after the final norm."""
        ids = input_ids.view(-1)
        T = ids.shape[0]
        pos = position_ids.view(-1) if position_ids is not None else torch.arange(T)
        h = _rnd(self.w["model.embed_tokens.weight"][ids], self.round_to)
        hs, kvs = [], []
        for i in range(self.shape.num_hidden_layers):
            hs.append(h.unsqueeze(0))
            h, present = self._layer(i, h, pos, None)
            kvs.append((present[0].unsqueeze(0), present[1].unsqueeze(0)))
        hn = rmsnorm(h, self.w["model.norm.weight"], self.shape.rms_norm_eps, self.round_to)
        hs.append(hn.unsqueeze(0))
        logits = _rnd(hn @ self.w["lm_head.weight"].T, self.round_to)
        return BaseOutput(logits.unsqueeze(0), tuple(hs), tuple(kvs))

    def forward_draft_or_large_model(self, in_tokens_small=None, in_features_large=None, position_ids=None):
        """kangaroo/earlyexit.py:14-71."""
        assert self.past_key_values is not None
        if in_tokens_small is not None and in_features_large is not None:
            raise ValueError("You cannot specify both in_tokens_small and in_features_large at the same time")
        elif in_tokens_small is not None:
            seq_length = in_tokens_small.shape[1]
        elif in_features_large is not None:
            seq_length = in_features_large.shape[1]
        else:
            raise ValueError("You have to specify either in_tokens_small or in_features_large")
        small = in_tokens_small is not None
        focus = 0 if small else self.early_exit_layer
        past_len = self.past_key_values[focus][0].shape[2]
        if position_ids is None:
            position_ids = torch.arange(past_len, past_len + seq_length).view(1, seq_length)
        pos = position_ids.view(-1)
        if small:
            h = _rnd(self.w["model.embed_tokens.weight"][in_tokens_small.view(-1)], self.round_to)
            layers = range(0, self.early_exit_layer)
        else:
            h = in_features_large[0].to(torch.float32)
            layers = range(self.early_exit_layer, self.shape.num_hidden_layers)
        for i in layers:
            k, v = self.past_key_values[i]
            h, present = self._layer(i, h, pos, (k[0], v[0]))
            self.past_key_values[i] = (present[0].unsqueeze(0), present[1].unsqueeze(0))
        if not small:
            hn = rmsnorm(h, self.w["model.norm.weight"], self.shape.rms_norm_eps, self.round_to)
            return h.unsqueeze(0), hn.unsqueeze(0)
        return h.unsqueeze(0)


class OracleAdapter:
    """kangaroo/adapter.py:317-460: one attention-only decoder layer (:262-314,
    residual at :304, no MLP, no post-attention norm) + final RMSNorm (:456)."""

    def __init__(self, shape: Shape, weights: Dict[str, torch.Tensor], round_to=None):
        self.shape = shape
        self.w = {k: v.to(torch.float32) for k, v in weights.items()}
        self.round_to = round_to
        self.tables = rope_tables(shape.head_dim, shape.max_position_embeddings, shape.rope_base)

    def forward_early_stop(self, inputs_embeds=None, position_ids=None, past_key_values=None, use_cache=True):
        x = inputs_embeds[0].to(torch.float32)
        q_len = x.shape[0]
        past_len = past_key_values[0][0].shape[2] if past_key_values is not None else 0
        if position_ids is None:
            pos = torch.arange(past_len, past_len + q_len)
        else:
            pos = position_ids.view(-1).long()
        w, r = self.w, self.round_to
        past = None if past_key_values is None else (past_key_values[0][0][0], past_key_values[0][1][0])
        xn = rmsnorm(x, w["layers.0.input_layernorm.weight"], self.shape.rms_norm_eps, r)
        a, present = attention(xn, pos, w["layers.0.self_attn.q_proj.weight"], w["layers.0.self_attn.k_proj.weight"],
                               w["layers.0.self_attn.v_proj.weight"], w["layers.0.self_attn.o_proj.weight"],
                               past, self.shape, self.tables, r)
        h = _rnd(x + a, r)
        h = rmsnorm(h, w["norm.weight"], self.shape.rms_norm_eps, r)
        if use_cache:
            return h.unsqueeze(0), ((present[0].unsqueeze(0), present[1].unsqueeze(0)),)
        return h.unsqueeze(0)


class OracleHead:
    """kangaroo/kangaroo_model.py:34-44: bias-free Linear(d, V) holding lm_head.weight."""

    def __init__(self, weight: torch.Tensor, round_to=None):
        self.weight = weight.to(torch.float32)
        self.round_to = round_to

    def __call__(self, h: torch.Tensor) -> torch.Tensor:
        return _rnd(h.to(torch.float32) @ self.weight.T, self.round_to)


class OracleKangarooModel:
    """kangaroo/kangaroo_model.py:12-47 (attribute surface only; weights are given)."""

    def __init__(self, shape: Shape, base_weights, adapter_weights, early_exit_layer: int = 2, round_to=None):
        self.base_model = OracleBase(shape, base_weights, early_exit_layer, round_to)
        self.adapter_model = OracleAdapter(shape, adapter_weights, round_to)
        self.head_model = OracleHead(base_weights["lm_head.weight"], round_to)


# --------------------------------------------------------------------------
# the loop
# --------------------------------------------------------------------------
@dataclass
class Trace:
    """Per-call record used by parity tests (not part of the reference's return)."""
    draft_lengths: List[int] = field(default_factory=list)       # drafted tokens per verify round
    p_max: List[float] = field(default_factory=list)             # confidence per draft step
    draft_tokens: List[int] = field(default_factory=list)        # argmax per draft step
    draft_logits: List[torch.Tensor] = field(default_factory=list)
    verify_logits: List[torch.Tensor] = field(default_factory=list)
    prefill_last_logits: Optional[torch.Tensor] = None


def typical_block(logits, global_tokens, start_index, end_index, eos_token_id, temperature, epsilon, delta, max_infer_steps):
    """The "typical" branch of evaluation/inference_kangaroo_MTBench.py:293-314, as written there: at every row
    whose argmax ends the run of matches (last row, eos, or a token other than the drafted one) the row's
    posterior is tested against min(epsilon, exp(-entropy) * delta); start_index moves on by ONE either way
    (:307,309), no token is written, a failed test or an eos stops the whole generation (:310,312), and the loop
    carries on with the next row against the SHIFTED start_index unless it stopped or ran out of budget (:313).
    The branch also rebinds the name `threshold` (:301), which is the draft loop's confidence threshold of :258
    from then on; the last value is returned for the caller to do the same.
    Returns (new start_index, stop, threshold tensor or None)."""
    output_tokens = torch.argmax(logits, dim=-1)                                                  # :276
    output_length = end_index - start_index                                                       # :277
    stop = False
    threshold = None
    for i in range(output_length):                                                                # :294
        if i == output_length - 1 or output_tokens[0, i] == eos_token_id or \
                output_tokens[0, i] != global_tokens[0, start_index + 1 + i]:                     # :295
            posterior_prob = torch.softmax(logits[:, i, :] / temperature, dim=-1)                 # :297
            posterior_entropy = -torch.sum(posterior_prob * torch.log(posterior_prob + 1e-5), dim=-1)   # :298-300
            threshold = torch.minimum(torch.ones_like(posterior_entropy) * epsilon,
                                      torch.exp(-posterior_entropy) * delta)                      # :301-304
            is_accepted = posterior_prob.max(dim=-1).values > threshold                           # :305
            if is_accepted:
                start_index += 1                                                                  # :307
            else:
                start_index += 1                                                                  # :309
                stop = True
            if output_tokens[0, i] == eos_token_id:                                               # :311
                stop = True
            if stop or start_index >= max_infer_steps:                                            # :313
                break
    return start_index, stop, threshold


def accept_block(logits, global_tokens, start_index, end_index, eos_token_id, accept_top_k=1, accept_top_p=None, temperature=1.0):
    """The verification block of one round, evaluation/inference_kangaroo.py:92-102 (greedy) with the
    membership tests of evaluation/inference_kangaroo_MTBench.py:356-389 swapped in for :97 when asked.
    logits: [1, q, V] fp32 verify logits of rows start_index .. end_index-1; global_tokens is updated in
    place (:98).  Returns (new start_index, stop)."""
    output_tokens = torch.argmax(logits, dim=-1)                                                  # :92
    stop = False
    output_length = end_index - start_index                                                       # :95
    if accept_top_p is not None:
        probs = torch.softmax(logits / temperature, dim=-1)                                       # MTBench :358-359
        # MTBench :360 calls torch.sort(probs, descending=True), which leaves the order of EQUAL probabilities to
        # the backend; stable=True pins it (token-id order) and changes nothing where no two probabilities tie
        sorted_probs, sorted_indices = torch.sort(probs, descending=True, stable=True)
        keep = torch.roll(sorted_probs.cumsum(dim=-1) <= accept_top_p, 1, dims=-1)                # MTBench :361-363
        keep[..., 0] = True                                                                       # MTBench :364
        pooled_nucleus = sorted_indices[keep]                                                     # MTBench :365
    elif accept_top_k > 1:
        output_topk_tokens = torch.topk(logits, k=accept_top_k, dim=-1).indices                   # MTBench :379
    for i in range(output_length):                                                                # :96
        if accept_top_p is not None:
            rejected = global_tokens[0, start_index + 1 + i] not in pooled_nucleus                # MTBench :369
        elif accept_top_k > 1:
            rejected = global_tokens[0, start_index + 1 + i] not in output_topk_tokens[0, i, :]   # MTBench :384
        else:
            rejected = output_tokens[0, i] != global_tokens[0, start_index + 1 + i]               # :97
        if i == output_length - 1 or output_tokens[0, i] == eos_token_id or rejected:
            global_tokens[0, start_index + 1 + i] = output_tokens[0, i]                           # :98
            start_index = start_index + 1 + i                                                     # :99
            if output_tokens[0, i] == eos_token_id:                                               # :100
                stop = True
            break
    return start_index, stop


def kangaroo_forward(input_ids: torch.Tensor, model, eos_token_id: int, max_new_tokens: int,
                     max_length: int = 2048, EARLY_STOP_LAYER: int = 2, SPECULATIVE_DECODING_STEPS: int = 6,
                     threshold: float = 0.6, trace: Optional[Trace] = None, keep_logits: bool = False,
                     accept_top_k: int = 1, accept_top_p: Optional[float] = None, temperature: float = 1.0,
                     accept_typical: Optional[Tuple[float, float]] = None):
    """evaluation/inference_kangaroo.py:15-129, statement for statement, with KV
    truncation expressed on the [1,Hkv,T,hd] tensors exactly as :108-119 does.

    accept_top_k > 1 swaps the greedy test of :97 for the top-k membership test of
    evaluation/inference_kangaroo_MTBench.py:377-389 (a drafted token is kept while it is among
    the k largest verify logits of its row; the first one that is not is replaced by the row's
    argmax).  accept_top_k == 1 is the greedy rule.

    accept_top_p swaps it for the top-p test of evaluation/inference_kangaroo_MTBench.py:356-374:
    every verify row's nucleus (tokens, by falling softmax(logits / temperature), whose predecessors
    sum to <= accept_top_p; the first always) is pooled over the round's rows (:365 flattens them),
    and a drafted token is kept while it is in the pool.

    accept_typical = (epsilon, delta) swaps it for the "typical" branch of
    evaluation/inference_kangaroo_MTBench.py:293-314 at `temperature` (see typical_block: no token is written,
    and the draft loop's threshold is replaced by the branch's last posterior threshold).

    Returns ([output_ids], new_token, idx, accept_length_list)."""
    context_tokens = input_ids
    device = context_tokens.device                                                                # :17
    batch_size, context_length = context_tokens.shape
    global_tokens = torch.ones((batch_size, max_length), dtype=torch.long, device=device) * eos_token_id   # :20
    global_position_ids = torch.arange(max_length, device=device).view(1, -1)                    # :21
    accept_length_list = [1]                                                                      # :22
    start_index = context_length                                                                  # :24
    global_tokens[:, :start_index] = context_tokens                                               # :25

    position_ids = global_position_ids[:, :start_index]                                           # :29
    output = model.base_model(context_tokens, position_ids=position_ids, output_hidden_states=True)  # :30
    model.base_model.past_key_values = list(output.past_key_values)                               # :31
    logits = output.logits                                                                        # :33
    global_tokens[:, start_index] = int(torch.argmax(logits[:, -1, :], dim=-1).item())            # :34
    if trace is not None and keep_logits:
        trace.prefill_last_logits = logits[0, -1].clone()
    hidden_state_early = output.hidden_states[EARLY_STOP_LAYER]                                   # :35
    _, adapter_past_key_values = model.adapter_model.forward_early_stop(                          # :38
        inputs_embeds=hidden_state_early, position_ids=global_position_ids[:, :context_length], use_cache=True)

    max_infer_steps = min(max_length, start_index + max_new_tokens)                               # :43
    stop = False
    predict_score = 0.0
    exited_hidden_states = None
    while start_index < max_infer_steps - 1 - SPECULATIVE_DECODING_STEPS:                         # :46
        start_index_copy = start_index
        end_index = start_index + 1                                                               # :49
        for step in range(1 + SPECULATIVE_DECODING_STEPS):                                        # :52
            assert adapter_past_key_values[0][0].shape[2] <= end_index - 1                        # :53
            in_tokens_small = global_tokens[:, end_index - 1:end_index]                           # :54
            if adapter_past_key_values[0][0].shape[2] < end_index - 1:                            # :55
                position_ids = global_position_ids[:, start_index - 1:end_index]                  # :57
                hidden_state_early_last = exited_hidden_states[:, -1:, :]                         # :58
            else:
                position_ids = global_position_ids[:, end_index - 1:end_index]                    # :60
                hidden_state_early_last = None
            hidden_state_early = model.base_model.forward_draft_or_large_model(                   # :63
                in_tokens_small=in_tokens_small[:, -1:], position_ids=position_ids[:, -1:])
            if step == 0:
                exited_hidden_states = None                                                       # :65-66
            exited_hidden_states = hidden_state_early if exited_hidden_states is None else \
                torch.cat([exited_hidden_states, hidden_state_early], dim=1)                      # :68
            if hidden_state_early_last is not None:
                hidden_state_early = torch.cat([hidden_state_early_last, hidden_state_early], dim=1)  # :71
            if step == SPECULATIVE_DECODING_STEPS or (step > 0 and predict_score < threshold):    # :74
                break
            hidden_state, adapter_past_key_values = model.adapter_model.forward_early_stop(       # :77
                inputs_embeds=hidden_state_early, position_ids=position_ids,
                past_key_values=adapter_past_key_values, use_cache=True)
            predict_logits = model.head_model(hidden_state[:, -1:, :]).float()                    # :79
            global_tokens[:, end_index] = torch.argmax(predict_logits[:, -1, :], dim=-1)          # :80
            end_index += 1                                                                        # :82
            predict_score = predict_logits.softmax(dim=-1).max().item()                           # :83
            if trace is not None:
                trace.p_max.append(predict_score)
                trace.draft_tokens.append(int(global_tokens[0, end_index - 1]))
                if keep_logits:
                    trace.draft_logits.append(predict_logits[0, -1].clone())

        position_ids = global_position_ids[:, start_index:end_index]                              # :86
        assert model.base_model.past_key_values[EARLY_STOP_LAYER][0].shape[2] == start_index      # :87
        assert exited_hidden_states.shape[1] == position_ids.shape[1]                             # :88
        _, hidden_state = model.base_model.forward_draft_or_large_model(                          # :89
            in_features_large=exited_hidden_states, position_ids=position_ids)
        logits = model.head_model(hidden_state).float()                                           # :91
        output_tokens = torch.argmax(logits, dim=-1)                                              # :92
        if trace is not None:
            trace.draft_lengths.append(end_index - start_index - 1)
            if keep_logits:
                trace.verify_logits.append(logits[0].clone())

        if accept_typical is not None:
            start_index, stop_now, new_threshold = typical_block(logits, global_tokens, start_index, end_index, eos_token_id,
                                                                 temperature, accept_typical[0], accept_typical[1], max_infer_steps)
            if new_threshold is not None:
                threshold = new_threshold                                                         # MTBench :301 rebinds the name
        else:
            start_index, stop_now = accept_block(logits, global_tokens, start_index, end_index, eos_token_id,   # :95-102
                                                 accept_top_k, accept_top_p, temperature)
        stop = stop or stop_now
        accept_length_list.append(start_index - start_index_copy)                                 # :104

        if model.base_model.past_key_values[0][0].shape[2] > start_index:                         # :108
            model.base_model.past_key_values = [
                (k[:, :, :start_index, :], v[:, :, :start_index, :]) for k, v in model.base_model.past_key_values]
        if adapter_past_key_values[0][0].shape[2] > start_index:                                  # :114
            adapter_past_key_values = tuple(
                (k[:, :, :start_index, :], v[:, :, :start_index, :]) for k, v in adapter_past_key_values)
        if stop:                                                                                  # :123
            break

    output_ids = global_tokens[0, :start_index + 1].tolist()                                      # :126
    new_token = start_index - context_length + 1                                                  # :127
    idx = len(accept_length_list) - 1                                                             # :128
    return [output_ids], new_token, idx, accept_length_list


def vanilla_forward(input_ids: torch.Tensor, model, eos_token_id: int, max_new_tokens: int,
                    keep_logits: bool = False):
    """Greedy one-token-at-a-time decode over the same base model: what
    evaluation/inference_baseline.py:14-25 asks HF `generate(do_sample=False)` to do.
    Stops after max_new_tokens or on eos.  Returns (output_ids, [logits rows])."""
    base = model.base_model
    T0 = input_ids.shape[1]
    out = base(input_ids, position_ids=torch.arange(T0).view(1, -1))
    base.past_key_values = list(out.past_key_values)
    ids = input_ids[0].tolist()
    rows = []
    logits = out.logits[0, -1]
    saved_exit = base.early_exit_layer
    for _ in range(max_new_tokens):
        tok = int(torch.argmax(logits))
        if keep_logits:
            rows.append(logits.clone())
        ids.append(tok)
        if tok == eos_token_id or len(ids) - T0 >= max_new_tokens:
            break
        # run all layers on one token: shallow part then deep part of the same stack
        pos = torch.tensor([[len(ids) - 1]])
        h = base.forward_draft_or_large_model(in_tokens_small=torch.tensor([[tok]]), position_ids=pos)
        _, hn = base.forward_draft_or_large_model(in_features_large=h, position_ids=pos)
        logits = model.head_model(hn)[0, -1]
    base.early_exit_layer = saved_exit
    return ids, rows
