"""Weight packing for libkangaroo_hip.so (layout described in include/kangaroo_hip.h).

A Linear weight W[N, K] becomes 1 KiB tiles of 16 rows x 32 k so that one wave-level
load is one contiguous tile and the 16 bytes a lane reads are its MFMA A fragment.
Pure torch; works on CPU and GPU tensors.
"""
from __future__ import annotations

import torch

TILE_ROWS = 16
TILE_K = 32


def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """[N, K] -> flat packed tensor of ceil16(N) * K elements (same dtype / device)."""
    assert w.dim() == 2
    n, k = w.shape
    if k % TILE_K:
        raise ValueError(f"in_features {k} must be a multiple of {TILE_K}")
    n_pad = (n + TILE_ROWS - 1) // TILE_ROWS * TILE_ROWS
    if n_pad != n:
        w = torch.cat([w, w.new_zeros(n_pad - n, k)], dim=0)
    g, kt = n_pad // TILE_ROWS, k // TILE_K
    # [rg, r, kt, kg, j] -> [rg, kt, kg, r, j]
    t = w.view(g, TILE_ROWS, kt, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
    return t.view(-1)


def unpack_weight(p: torch.Tensor, n: int, k: int) -> torch.Tensor:
    """Inverse of pack_weight (tests)."""
    n_pad = (n + TILE_ROWS - 1) // TILE_ROWS * TILE_ROWS
    g, kt = n_pad // TILE_ROWS, k // TILE_K
    w = p.view(g, kt, 4, TILE_ROWS, 8).permute(0, 3, 1, 2, 4).contiguous().view(n_pad, k)
    return w[:n]


def rope_pair_order(n_heads: int, head_dim: int) -> torch.Tensor:
    """Row permutation that makes each rotation pair (j, j + hd/2) adjacent:
    new row 2j <- old row j, new row 2j+1 <- old row j + hd/2, per head."""
    half = head_dim // 2
    j = torch.arange(half)
    per_head = torch.stack([j, j + half], dim=1).reshape(-1)            # [hd]
    return (torch.arange(n_heads).unsqueeze(1) * head_dim + per_head.unsqueeze(0)).reshape(-1)


def pack_qkv(wq: torch.Tensor, wk: torch.Tensor, wv: torch.Tensor, n_heads: int, n_kv_heads: int, head_dim: int) -> torch.Tensor:
    """Concatenate q, k, v projections with rotation pairs adjacent, then pack."""
    oq = rope_pair_order(n_heads, head_dim).to(wq.device)
    okv = rope_pair_order(n_kv_heads, head_dim).to(wq.device)
    return pack_weight(torch.cat([wq[oq], wk[okv], wv[okv]], dim=0))


def pack_gate_up(wg: torch.Tensor, wu: torch.Tensor) -> torch.Tensor:
    """Rows (2i, 2i+1) = (gate_i, up_i), then pack."""
    f, d = wg.shape
    return pack_weight(torch.stack([wg, wu], dim=1).reshape(2 * f, d))


def rope_tables(head_dim: int, max_pos: int, base: float = 10000.0, device="cpu"):
    """cos / sin tables [max_pos, head_dim/2] in fp32, computed as the reference's
    LlamaRotaryEmbedding does (kangaroo/adapter.py:104,112-120); the second half of the
    reference's table repeats the first, so half is stored."""
    inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2).float() / head_dim))
    t = torch.arange(max_pos, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return freqs.cos().contiguous().to(device), freqs.sin().contiguous().to(device)
