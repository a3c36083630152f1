"""Model shapes for the Kangaroo decode path.

The 7B / 13B adapter shapes come from the reference's data/vicuna_7B_config.json:8-21
and data/vicuna_13B_config.json:8-21; base layer counts (32 / 40) from
data/ge_data_all_vicuna.py:142,149-152 (exit layer 2 for 7B, 3 for 13B).
TinyLlama-1.1B and LLaMA-33B are the public architectures of those names.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, asdict


@dataclass
class LlamaShape:
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

    @property
    def q_dim(self) -> int:
        return self.num_attention_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_key_value_heads * self.head_dim

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_hf_config(cls, path_or_dict) -> "LlamaShape":
        """Read the fields of a HF LlamaConfig json (kangaroo/kangaroo_model.py:25)."""
        cfg = path_or_dict
        if not isinstance(cfg, dict):
            p = path_or_dict
            if os.path.isdir(p):
                p = os.path.join(p, "config.json")
            with open(p, "r") as f:
                cfg = json.load(f)
        heads = cfg["num_attention_heads"]
        return cls(
            hidden_size=cfg["hidden_size"],
            num_hidden_layers=cfg["num_hidden_layers"],
            num_attention_heads=heads,
            num_key_value_heads=cfg.get("num_key_value_heads") or heads,
            intermediate_size=cfg["intermediate_size"],
            vocab_size=cfg["vocab_size"],
            rms_norm_eps=cfg.get("rms_norm_eps", 1e-6),
            max_position_embeddings=cfg.get("max_position_embeddings", 2048),
            rope_base=cfg.get("rope_theta", 10000.0),
        )

    # ---- byte accounting used by the benchmark's roofline figures -----------------
    def layer_weight_bytes(self, elt: int = 2) -> int:
        d, f = self.hidden_size, self.intermediate_size
        return (2 * d * self.q_dim + 2 * d * self.kv_dim + 3 * d * f) * elt

    def attn_only_weight_bytes(self, elt: int = 2) -> int:
        d = self.hidden_size
        return (2 * d * self.q_dim + 2 * d * self.kv_dim) * elt

    def head_weight_bytes(self, elt: int = 2) -> int:
        return self.vocab_size * self.hidden_size * elt

    def kv_bytes_per_position(self, elt: int = 2) -> int:
        return 2 * self.kv_dim * elt


SHAPES = {
    # small model for unit / golden tests: GQA (Hkv < H), odd vocabulary
    "tiny": LlamaShape(64, 4, 4, 2, 128, 97, 1e-6, 2048),
    # tiny with MHA and one more layer (exit-layer-3 coverage)
    "tiny-mha": LlamaShape(64, 5, 4, 4, 96, 131, 1e-6, 2048),
    "small": LlamaShape(256, 6, 8, 8, 704, 1000, 1e-6, 2048),
    "tinyllama-1.1b": LlamaShape(2048, 22, 32, 4, 5632, 32000, 1e-5, 2048),
    "vicuna-7b": LlamaShape(4096, 32, 32, 32, 11008, 32000, 1e-6, 2048),
    "vicuna-13b": LlamaShape(5120, 40, 40, 40, 13824, 32000, 1e-6, 2048),
    "llama-33b": LlamaShape(6656, 60, 52, 52, 17920, 32000, 1e-6, 2048),
}

DEFAULT_EXIT_LAYER = {"vicuna-7b": 2, "vicuna-13b": 3}


def get_shape(name: str) -> LlamaShape:
    return SHAPES[name]
