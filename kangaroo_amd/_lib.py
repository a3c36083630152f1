"""ctypes binding of libkangaroo_hip.so (declared in include/kangaroo_hip.h).

There is no CPU fallback: if the library is missing the import of this module's
`lib()` raises, and every op built on it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkangaroo_hip.so")

KG_MAX_ROWS = 16
DT_F16, DT_BF16 = 0, 1


class KgShape(C.Structure):
    _fields_ = [("dtype", C.c_int), ("hidden_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
                ("num_kv_heads", C.c_int), ("head_dim", C.c_int), ("intermediate_size", C.c_int), ("vocab_size", C.c_int),
                ("rms_eps", C.c_float), ("max_positions", C.c_int), ("exit_layer", C.c_int)]


class KgLayerWeights(C.Structure):
    _fields_ = [("wqkv", C.c_void_p), ("wo", C.c_void_p), ("wgu", C.c_void_p), ("wdown", C.c_void_p),
                ("norm1", C.c_void_p), ("norm2", C.c_void_p)]


class KgWeights(C.Structure):
    _fields_ = [("embed", C.c_void_p), ("final_norm", C.c_void_p), ("lm_head", C.c_void_p),
                ("layers", C.POINTER(KgLayerWeights)), ("adapter", KgLayerWeights), ("adapter_norm", C.c_void_p),
                ("rope_cos", C.c_void_p), ("rope_sin", C.c_void_p)]


class KgBuffers(C.Structure):
    _fields_ = [("k_cache", C.c_void_p), ("v_cache", C.c_void_p), ("ak_cache", C.c_void_p), ("av_cache", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("t_max", C.c_int)]


class KgGenResult(C.Structure):
    _fields_ = [("n_tokens", C.c_int), ("new_tokens", C.c_int), ("rounds", C.c_int), ("drafted_total", C.c_int),
                ("stopped_on_eos", C.c_int), ("prefill_ms", C.c_float), ("decode_ms", C.c_float)]


KG_TP_MAX = 8


class KgTpConfig(C.Structure):
    _fields_ = [("tp_size", C.c_int), ("n_local", C.c_int), ("local_rank", C.c_int * KG_TP_MAX),
                ("heads_local", C.c_int), ("kv_heads_local", C.c_int), ("ffn_local", C.c_int),
                ("layers", C.POINTER(KgLayerWeights) * KG_TP_MAX), ("k_cache", C.c_void_p * KG_TP_MAX), ("v_cache", C.c_void_p * KG_TP_MAX),
                ("xbuf", C.c_void_p), ("peer_xbuf", C.c_void_p * KG_TP_MAX)]


# every symbol include/kangaroo_hip.h declares: (name, restype, argtypes)
_VP, _I, _F, _D, _SZ = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
SYMBOLS = {
    "kg_version": (_I, []),
    "kg_last_error": (C.c_char_p, []),
    "kg_linear": (_I, [_I, _VP, _I, _I, _VP, _I, _I, _VP, _F, _I, _VP, _VP, _I, _VP, _I, _I, _VP]),
    "kg_linear_scratch_bytes": (_SZ, [_I, _I]),
    "kg_qkv_rope_append": (_I, [_I, _VP, _I, _I, _I, _I, _VP, _I, _I, _VP, _F, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _I, _I, _VP]),
    "kg_attention": (_I, [_I, _VP, _I, _I, _VP, _VP, _I, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "kg_attention_scratch_bytes": (_SZ, [_I, _I, _I]),
    "kg_head": (_I, [_I, _VP, _I, _I, _VP, _I, _I, _VP, _F, _VP, _VP, _VP, _VP, _I, _I, _VP]),
    "kg_head_scratch_bytes": (_SZ, [_I]),
    "kg_accept": (_I, [_I, _VP, _I, _I, C.POINTER(C.c_int32), _I, _I, _I, _I, _F, _F, _F, _F, _I, _I, _I, C.POINTER(C.c_int32), _VP, _VP]),
    "kg_draft_step": (_I, [_I, _VP, _I, _I, _I, _D, _I, _I, _I, C.POINTER(C.c_int32), _VP, _VP]),
    "kg_control_scratch_bytes": (_SZ, []),
    "kg_head_noalloc_init": (_I, []),
    "kg_head_noalloc_max_vocab": (_SZ, []),
    "kg_head_noalloc": (_I, [_I, _VP, _I, _I, _VP, _I, _I, _VP, _F, _VP, _VP, _VP, _I, _I, _VP]),
    "kg_workspace_bytes": (_SZ, [C.POINTER(KgShape), _I]),
    "kg_session_create": (_I, [C.POINTER(KgShape), C.POINTER(KgWeights), C.POINTER(KgBuffers), C.POINTER(_VP)]),
    "kg_session_destroy": (_I, [_VP]),
    "kg_session_set": (_I, [_VP, C.c_char_p, _I]),
    "kg_session_get": (_I, [_VP, C.c_char_p]),
    "kg_session_setf": (_I, [_VP, C.c_char_p, C.c_float]),
    "kg_session_getf": (C.c_float, [_VP, C.c_char_p]),
    "kg_session_ptr": (_VP, [_VP, C.c_char_p]),
    "kg_set_tokens": (_I, [_VP, C.POINTER(C.c_int32), _I, _I, _VP]),
    "kg_forward_layers": (_I, [_VP, _I, _I, _VP, _I, _I, _VP, _VP]),
    "kg_final_norm": (_I, [_VP, _VP, _I, _VP, _VP]),
    "kg_prefill_rows": (_I, [_VP, _I, _VP, _VP, _VP, _VP]),
    "kg_adapter_forward_rows": (_I, [_VP, _VP, _I, _VP, _VP]),
    "kg_adapter_forward": (_I, [_VP, _VP, _I, _I, _VP, _VP]),
    "kg_head_logits": (_I, [_VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "kg_tp_xbuf_bytes": (_SZ, [_I, _I]),
    "kg_tp_xbuf_create": (_I, [_SZ, C.POINTER(_VP), _VP]),
    "kg_tp_xbuf_open": (_I, [_VP, C.POINTER(_VP)]),
    "kg_tp_xbuf_close": (_I, [_VP]),
    "kg_tp_xbuf_destroy": (_I, [_VP]),
    "kg_session_set_tp": (_I, [_VP, C.POINTER(KgTpConfig)]),
    "kg_matmul_nt": (_I, [_I, _VP, _VP, _VP, _I, _I, _I, C.c_longlong, C.c_longlong, C.c_longlong, _I, C.c_longlong, C.c_longlong, C.c_longlong, _I, _I, _I, _F, _VP]),
    "kg_matmul_nt_heads": (_I, [_I, _VP, _VP, _VP, _I, _I, _I, C.c_longlong, C.c_longlong, C.c_longlong, _I, C.c_longlong, C.c_longlong, C.c_longlong,
                                _I, C.c_longlong, C.c_longlong, C.c_longlong, _I, _I, _I, _I, _F, _VP]),
    "kg_matmul_nt_ranges": (_I, [_I, _VP, _VP, _VP, _I, _I, _I, C.c_longlong, C.c_longlong, C.c_longlong, _I, _I, C.c_longlong, _VP]),
    "kg_transpose16": (_I, [_VP, _VP, _I, _I, C.c_longlong, C.c_longlong, _I, C.c_longlong, C.c_longlong, _VP]),
    "kg_bench_op": (_I, [_VP, _I, _I, _I, C.POINTER(C.c_float), C.POINTER(C.c_size_t)]),
    "kg_train_rmsnorm_fwd": (_I, [_I, _VP, _VP, _F, _I, _I, _VP, _VP, _VP]),
    "kg_train_rmsnorm_bwd_scratch": (_SZ, [_I, _I]),
    "kg_train_rmsnorm_bwd": (_I, [_I, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP]),
    "kg_train_rope": (_I, [_I, _VP, C.c_longlong, _I, _I, _I, _VP, _VP, _I, _VP]),
    "kg_train_softmax_fwd": (_I, [_I, _VP, _VP, _I, _I, _I, _F, _VP]),
    "kg_train_softmax_bwd": (_I, [_I, _VP, _VP, _I, _I, _I, _F, _VP]),
    "kg_train_softce": (_I, [_I, _VP, _VP, _VP, _I, _I, C.c_longlong, _F, _VP, _VP, _VP, _VP]),
    "kg_train_add": (_I, [_I, _VP, _VP, C.c_longlong, _VP]),
    "kg_train_nonfinite": (_I, [_VP, C.c_longlong, _VP, _VP]),
    "kg_train_adamw": (_I, [_VP, _VP, _VP, _VP, _I, _F, _F, _F, _F, _F, _F, _F, _I, C.c_longlong, _VP, _I, _VP, _VP]),
    "kg_generate": (_I, [_VP, _I, C.POINTER(C.c_int32), _I, _I, _I, _I, _D, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I,
                         C.POINTER(KgGenResult), _VP]),
}

_lib = None


class KangarooHipError(RuntimeError):
    pass


def lib():
    """Load the shared library once and bind the signatures."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KangarooHipError(
            f"{LIB_PATH} is missing: build it with `make -C kangaroo_amd/csrc` (hipcc, gfx950). "
            "There is no CPU fallback for the decode path.")
    h = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(h, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = h
    return h


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().kg_last_error().decode("utf-8", "replace")
        raise KangarooHipError(f"{what} failed with code {rc}: {msg}")
