"""KangarooModel: base model + adapter + separate lm_head, on one MI355X.

Mirror of the reference's KangarooModel (kangaroo/kangaroo_model.py:12-47): same
constructor signature `(base_model_name_or_path, adapter_model_path, args,
EARLY_STOP_LAYER=2)`, same attributes `base_model`, `adapter_model`, `head_model`, and
`forward()` raises NotImplementedError (:46-47).  Reads the same on-disk formats (HF
sharded checkpoint through pytorch_model.bin.index.json; adapter directory with
config.json + pytorch_model.bin loaded non-strictly).  `args.dtype` "float16" or
"bfloat16" selects the storage type of every part (the reference leaves adapter and
head in fp32 when dtype is bfloat16, :31-32,43-44; here all three follow dtype).
`from_state_dicts` builds the same object from in-memory weights (synthetic runs).
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from .adapter import AdapterModel
from .config import LlamaShape
from .earlyexit import EarlyExitLlamaForCausalLM
from .engine import KangarooEngine
from .weights import load_adapter_state_dict, load_base_state_dict

_DTYPES = {"float16": torch.float16, "bfloat16": torch.bfloat16, "half": torch.float16}


class HeadModel:
    """head_model: bias-free Linear(d, V) holding lm_head.weight (kangaroo_model.py:34-44)."""

    def __init__(self, engine: KangarooEngine):
        self.engine = engine
        # True: a lone row is computed by the multi-row kernel, i.e. exactly as a row of a verify pass; plain decoding
        # driven pass by pass (no draft steps) sets it so that its tokens equal the Kangaroo loop's bit for bit
        self.single_rows_in_verify_arithmetic = False

    def eval(self):
        return self

    def half(self):
        return self

    @torch.no_grad()
    def __call__(self, hidden: torch.Tensor) -> torch.Tensor:
        e = self.engine
        lead = hidden.shape[:-1]
        x = hidden.reshape(-1, hidden.shape[-1])
        rows = [e.head_logits(x[i:i + 16], verify_arithmetic=self.single_rows_in_verify_arithmetic) for i in range(0, x.shape[0], 16)]
        return torch.cat(rows, 0).reshape(*lead, e.shape.vocab_size)

    forward = __call__


class KangarooModel:
    def __init__(self, base_model_name_or_path: str, adapter_model_path: str, args, EARLY_STOP_LAYER: int = 2,
                 max_length: int = 2048, device="cuda"):
        dtype_name = getattr(args, "dtype", "float16")
        if dtype_name not in _DTYPES:
            raise ValueError(f"dtype {dtype_name!r}: the HIP decode path runs float16 or bfloat16")
        shape = LlamaShape.from_hf_config(base_model_name_or_path)
        adapter_shape = LlamaShape.from_hf_config(os.path.join(adapter_model_path, "config.json"))
        if adapter_shape.hidden_size != shape.hidden_size or adapter_shape.vocab_size != shape.vocab_size:
            raise ValueError("adapter config does not match the base model")
        base_sd = load_base_state_dict(base_model_name_or_path)
        adapter_sd = load_adapter_state_dict(adapter_model_path)
        self._build(shape, base_sd, adapter_sd, _DTYPES[dtype_name], EARLY_STOP_LAYER, max_length, device)

    @classmethod
    def from_state_dicts(cls, shape: LlamaShape, base_sd: Dict[str, torch.Tensor], adapter_sd: Dict[str, torch.Tensor],
                         dtype: torch.dtype = torch.float16, EARLY_STOP_LAYER: int = 2, max_length: int = 2048,
                         device="cuda", release_source: bool = False) -> "KangarooModel":
        self = cls.__new__(cls)
        self._build(shape, base_sd, adapter_sd, dtype, EARLY_STOP_LAYER, max_length, device, release_source)
        return self

    def _build(self, shape, base_sd, adapter_sd, dtype, exit_layer, max_length, device, release_source=False):
        self.engine = KangarooEngine(shape, base_sd, adapter_sd, exit_layer=exit_layer, dtype=dtype, device=device,
                                     max_length=max_length, release_source=release_source)
        self.base_model = EarlyExitLlamaForCausalLM(self.engine, exit_layer)
        self.adapter_model = AdapterModel(self.engine)
        self.head_model = HeadModel(self.engine)

    def eval(self):
        return self

    def forward(self):
        raise NotImplementedError
