"""Synthetic weights (no checkpoints exist offline) and checkpoint readers.

State-dict key names follow HF LlamaForCausalLM for the base model and the
reference's AdapterModel (kangaroo/adapter.py:324-325) for the adapter, so a real
checkpoint read by `load_base_state_dict` / `load_adapter_state_dict` and a
synthetic one are interchangeable.

Recipes steer how often the draft agrees with the target when all weights are
random (a random adapter never agrees):
  deep_scale   multiplies o_proj / down_proj of layers >= exit layer (0 => the deep
               stack is the identity on the residual stream)
  head_scale   multiplies lm_head (moves the draft's top-1 probability against the
               threshold)
  adapter_o_zero         zero the adapter's o_proj (adapter = final norm only)
  adapter_norm_from_base copy base final-norm weight into the adapter's final norm
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from .config import LlamaShape


@dataclass
class Recipe:
    deep_scale: float = 1.0
    head_scale: float = 1.0
    adapter_o_zero: bool = False
    adapter_norm_from_base: bool = False
    gain: float = 1.0          # Linear weights ~ N(0, (gain / sqrt(fan_in))^2): shape-independent activations
    norm_jitter: float = 0.1

    @classmethod
    def agreeable(cls, deep_scale: float, head_scale: float) -> "Recipe":
        return cls(deep_scale=deep_scale, head_scale=head_scale, adapter_o_zero=True, adapter_norm_from_base=True)


def _randn(shape, gen: torch.Generator, std: float, dtype, device):
    # generate in fp32 in row chunks to bound peak memory at large shapes
    rows = shape[0]
    out = torch.empty(shape, dtype=dtype, device=device)
    step = max(1, (64 << 20) // max(1, (shape[1] if len(shape) > 1 else 1)))
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        blk = torch.randn((r1 - r0,) + tuple(shape[1:]), generator=gen, dtype=torch.float32, device=device) * std
        out[r0:r1] = blk.to(dtype)
    return out


def synth_state_dicts(shape: LlamaShape, exit_layer: int, seed: int = 0, dtype=torch.float16,
                      recipe: Optional[Recipe] = None, device="cpu") -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
    """Seeded random base + adapter state dicts in `dtype` on `device`.

    Values depend on (seed, device type); CPU and GPU generators differ, so compare
    only runs whose weights came from one call."""
    recipe = recipe or Recipe()
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    d, f, v = shape.hidden_size, shape.intermediate_size, shape.vocab_size
    qd, kvd = shape.q_dim, shape.kv_dim

    def lin(n, k, scale=1.0):
        return _randn((n, k), gen, recipe.gain * scale / (k ** 0.5), dtype, device)

    def norm():
        w = 1.0 + recipe.norm_jitter * torch.randn((d,), generator=gen, dtype=torch.float32, device=device)
        return w.to(dtype)

    base: Dict[str, torch.Tensor] = {}
    # unit-variance-ish embeddings keep early residual streams well conditioned
    base["model.embed_tokens.weight"] = _randn((v, d), gen, 1.0, dtype, device)
    for i in range(shape.num_hidden_layers):
        p = f"model.layers.{i}."
        s = recipe.deep_scale if i >= exit_layer else 1.0
        base[p + "self_attn.q_proj.weight"] = lin(qd, d)
        base[p + "self_attn.k_proj.weight"] = lin(kvd, d)
        base[p + "self_attn.v_proj.weight"] = lin(kvd, d)
        base[p + "self_attn.o_proj.weight"] = lin(d, qd, s)
        base[p + "mlp.gate_proj.weight"] = lin(f, d)
        base[p + "mlp.up_proj.weight"] = lin(f, d)
        base[p + "mlp.down_proj.weight"] = lin(d, f, s)
        base[p + "input_layernorm.weight"] = norm()
        base[p + "post_attention_layernorm.weight"] = norm()
    base["model.norm.weight"] = norm()
    base["lm_head.weight"] = lin(v, d, recipe.head_scale)

    adapter: Dict[str, torch.Tensor] = {}
    a = "layers.0."
    adapter[a + "self_attn.q_proj.weight"] = lin(qd, d)
    adapter[a + "self_attn.k_proj.weight"] = lin(kvd, d)
    adapter[a + "self_attn.v_proj.weight"] = lin(kvd, d)
    adapter[a + "self_attn.o_proj.weight"] = lin(d, qd, 0.0 if recipe.adapter_o_zero else 1.0)
    adapter[a + "input_layernorm.weight"] = norm()
    adapter["norm.weight"] = base["model.norm.weight"].clone() if recipe.adapter_norm_from_base else norm()
    return base, adapter


# --------------------------------------------------------------------------
# on-disk formats the reference reads (kangaroo/kangaroo_model.py:22-44)
# --------------------------------------------------------------------------
def _torch_load(path: str):
    return torch.load(path, map_location="cpu", weights_only=True)


def load_base_state_dict(model_dir: str, only=None) -> Dict[str, torch.Tensor]:
    """HF sharded checkpoint: pytorch_model.bin.index.json -> shard files, or a single
    pytorch_model.bin, or safetensors equivalents.  only = names to keep: with an index file just the
    shards that hold them are read (train.py:69-86 fetches lm_head.weight this way)."""
    idx = os.path.join(model_dir, "pytorch_model.bin.index.json")
    sidx = os.path.join(model_dir, "model.safetensors.index.json")
    out: Dict[str, torch.Tensor] = {}
    if os.path.exists(idx):
        with open(idx, "r") as f:
            wm = json.load(f)["weight_map"]
        for shard in sorted(set(wm.values()) if only is None else set(wm[k] for k in only)):
            out.update(_torch_load(os.path.join(model_dir, shard)))
    elif os.path.exists(sidx):
        from safetensors.torch import load_file
        with open(sidx, "r") as f:
            wm = json.load(f)["weight_map"]
        for shard in sorted(set(wm.values()) if only is None else set(wm[k] for k in only)):
            out.update(load_file(os.path.join(model_dir, shard)))
    elif os.path.exists(os.path.join(model_dir, "pytorch_model.bin")):
        out = _torch_load(os.path.join(model_dir, "pytorch_model.bin"))
    elif os.path.exists(os.path.join(model_dir, "model.safetensors")):
        from safetensors.torch import load_file
        out = load_file(os.path.join(model_dir, "model.safetensors"))
    else:
        raise FileNotFoundError(f"no checkpoint found under {model_dir}")
    if only is not None:
        out = {k: out[k] for k in only}
    return out


def load_adapter_state_dict(adapter_dir: str) -> Dict[str, torch.Tensor]:
    """adapter_dir/pytorch_model.bin, loaded non-strictly as kangaroo_model.py:28 does
    (unknown keys are ignored by the consumer)."""
    return _torch_load(os.path.join(adapter_dir, "pytorch_model.bin"))


def save_synthetic_checkpoint(model_dir: str, adapter_dir: str, shape: LlamaShape, base, adapter, n_shards: int = 2):
    """Write base/adapter in the reference's on-disk layout (used by loader tests)."""
    os.makedirs(model_dir, exist_ok=True)
    os.makedirs(adapter_dir, exist_ok=True)
    keys = list(base.keys())
    per = (len(keys) + n_shards - 1) // n_shards
    wm = {}
    for s in range(n_shards):
        name = f"pytorch_model-{s + 1:05d}-of-{n_shards:05d}.bin"
        part = {k: base[k].cpu() for k in keys[s * per:(s + 1) * per]}
        torch.save(part, os.path.join(model_dir, name))
        for k in part:
            wm[k] = name
    with open(os.path.join(model_dir, "pytorch_model.bin.index.json"), "w") as f:
        json.dump({"metadata": {}, "weight_map": wm}, f)
    cfg = {
        "architectures": ["LlamaForCausalLM"], "hidden_act": "silu", "model_type": "llama",
        "hidden_size": shape.hidden_size, "intermediate_size": shape.intermediate_size,
        "num_attention_heads": shape.num_attention_heads, "num_key_value_heads": shape.num_key_value_heads,
        "num_hidden_layers": shape.num_hidden_layers, "vocab_size": shape.vocab_size,
        "rms_norm_eps": shape.rms_norm_eps, "max_position_embeddings": shape.max_position_embeddings,
        "bos_token_id": 1, "eos_token_id": 2, "pad_token_id": 0,
    }
    with open(os.path.join(model_dir, "config.json"), "w") as f:
        json.dump(cfg, f)
    acfg = dict(cfg)
    acfg["num_hidden_layers"] = 1
    with open(os.path.join(adapter_dir, "config.json"), "w") as f:
        json.dump(acfg, f)
    torch.save({k: v.cpu() for k, v in adapter.items()}, os.path.join(adapter_dir, "pytorch_model.bin"))
