"""kg_matmul_nt / kg_matmul_nt_heads / kg_transpose16 (csrc/mm.hip) against fp32 torch math: the matrix products of the
adapter training step (train.py:258-296 through kangaroo/adapter.py:170-222) and of the whole-prompt pass
(evaluation/inference_kangaroo.py:30).  Tolerances: inputs are 16-bit, accumulation fp32; an fp32 result must agree to
fp32 summation noise, a 16-bit result to one rounding of the output type."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from kangaroo_amd import _lib
from kangaroo_amd.packing import pack_weight

DEV = "cuda"


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def mm(A, B, out_f32=False, layout=0, causal=0, alpha=1.0, N=None, ldb=None):
    lib = _lib.lib()
    dt = _lib.DT_F16 if A.dtype == torch.float16 else _lib.DT_BF16
    batch = A.shape[0] if A.dim() == 3 else 1
    M, K = A.shape[-2], A.shape[-1]
    if N is None:
        N = B.shape[-1] if layout == 2 else B.shape[-2]
    out = torch.empty((batch, M, N) if A.dim() == 3 else (M, N), dtype=torch.float32 if out_f32 else A.dtype, device=A.device)
    sB = 0 if (layout == 1 or B.dim() == 2) else B.stride(0)
    _lib.check(lib.kg_matmul_nt(dt, A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, A.stride(-2), 0 if layout == 1 else B.stride(-2), N,
                                batch, A.stride(0) if A.dim() == 3 else 0, sB, M * N if A.dim() == 3 else 0, layout, 1 if out_f32 else 0, causal, alpha, st()), "kg_matmul_nt")
    return out


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
def test_products_against_fp32(dtype, tile, monkeypatch):
    # 1 / 2 / 3 / 4: every shape through the 128 x 128 tile / the 256 x 256 tile staged through registers / the 256 x 256
    # and the 128 x 256 tile whose loads write LDS directly (K-major B: the register-staged ones)
    monkeypatch.setenv("KG_MM_TILE", str(tile))
    g = torch.Generator(device=DEV).manual_seed(tile)
    out_tol = 2e-3 if dtype == torch.float16 else 1.2e-2
    for (M, N, K) in ((128, 128, 64), (200, 136, 72), (1000, 520, 4096), (77, 4096, 256), (257, 129, 8), (513, 700, 1032), (300, 260, 192), (256, 512, 128), (130, 300, 320), (128, 256, 448)):
        A = torch.randn(M, K, device=DEV, generator=g).to(dtype); B = torch.randn(N, K, device=DEV, generator=g).to(dtype)
        ref = A.float() @ B.float().T
        assert float((mm(A, B, out_f32=True) - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-4
        assert float((mm(A, B).float() - ref).abs().max()) <= out_tol * float(ref.abs().max())
        # K-major B: the same product from B^T stored [K][N]
        Bt = B.T.contiguous() if N % 8 == 0 else None
        if Bt is not None:
            assert torch.equal(mm(A, Bt, out_f32=True, layout=2), mm(A, B, out_f32=True))
    # the decode path's packed weights: same products, same order
    W = torch.randn(528, 4096, device=DEV, generator=g).to(dtype); X = torch.randn(300, 4096, device=DEV, generator=g).to(dtype)
    assert torch.equal(mm(X, pack_weight(W), out_f32=True, layout=1, N=528), mm(X, W, out_f32=True))
    # a view into wider rows, alpha
    Xw = torch.randn(300, 8192, device=DEV, generator=g).to(dtype)
    assert torch.equal(mm(Xw[:, 1024:5120], W, out_f32=True), mm(Xw[:, 1024:5120].contiguous(), W, out_f32=True))
    assert torch.allclose(mm(X, W, out_f32=True, alpha=0.25), 0.25 * mm(X, W, out_f32=True))


def test_direct_to_lds_tiles_repeat_bit_for_bit_and_equal_the_register_staged_tile(monkeypatch):
    """The two tile forms whose loads write LDS directly run several half tiles ahead of their readers with counted waits
    between raw barriers; a read placed a phase too early, or a refill a phase too soon, shows as a rare wrong tile, not as a
    crash.  Every form walks K in the same order per accumulator, so the fp32 results must equal the register-staged
    tile's bit for bit, launch after launch, at sizes with many steps, many tiles per CU, ragged edges, odd step counts and
    K ranges on grid.y."""
    g = torch.Generator(device=DEV).manual_seed(11)
    lib = _lib.lib()
    for (M, N, K) in ((2048, 4096, 4096), (640, 4352, 1984), (1500, 1024, 8192), (300, 520, 4160), (256, 256, 64), (1100, 2000, 192)):
        A = torch.randn(M, K, device=DEV, generator=g).to(torch.float16); B = torch.randn(N, K, device=DEV, generator=g).to(torch.float16)
        Bp = pack_weight(B) if K % 32 == 0 else None
        monkeypatch.setenv("KG_MM_TILE", "2")
        want = mm(A, B, out_f32=True)
        for tile in ("3", "4"):
            monkeypatch.setenv("KG_MM_TILE", tile)
            for rep in range(12):
                assert torch.equal(mm(A, B, out_f32=True), want), (M, N, K, tile, rep)
                if Bp is not None:
                    assert torch.equal(mm(A, Bp, out_f32=True, layout=1, N=N), want), (M, N, K, tile, rep, "packed")
    # K ranges: partial products of whole 64-steps, added in range order, against the one-range product in fp32
    monkeypatch.delenv("KG_MM_TILE")
    M, N, K = 640, 4096, 11008
    A = torch.randn(M, K, device=DEV, generator=g).to(torch.float16); B = torch.randn(N, K, device=DEV, generator=g).to(torch.float16)
    ref = A.float() @ B.float().T
    for n in (2, 3, 8):
        part = torch.full((n, M, N), float("nan"), dtype=torch.float32, device=DEV)
        _lib.check(lib.kg_matmul_nt_ranges(_lib.DT_F16, A.data_ptr(), B.data_ptr(), part.data_ptr(), M, N, K, K, K, N, 0, n, M * N, st()), "kg_matmul_nt_ranges")
        assert not bool(torch.isnan(part).any())
        assert float((part.sum(0) - ref).abs().max()) <= 2e-5 * float(ref.abs().max()) + 1e-4
        for rep in range(4):
            again = torch.empty_like(part)
            _lib.check(lib.kg_matmul_nt_ranges(_lib.DT_F16, A.data_ptr(), B.data_ptr(), again.data_ptr(), M, N, K, K, K, N, 0, n, M * N, st()), "kg_matmul_nt_ranges")
            assert torch.equal(again, part)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("tile", [0, 3, 4])
def test_causal_and_head_strided_products(dtype, tile, monkeypatch):
    monkeypatch.setenv("KG_MM_TILE", str(tile))
    lib = _lib.lib()
    dtc = _lib.DT_F16 if dtype == torch.float16 else _lib.DT_BF16
    g = torch.Generator(device=DEV).manual_seed(3)
    B_, S, H, Hkv, hd = 2, 296, 4, 2, 64
    rep, qd, kvd = H // Hkv, H * hd, Hkv * hd
    q = torch.randn(B_, S, H, hd, device=DEV, generator=g).to(dtype); k = torch.randn(B_, S, Hkv, hd, device=DEV, generator=g).to(dtype)
    tril = torch.ones(S, S, device=DEV).tril().bool()
    # scores of every (batch, head) from the [B, S, H, hd] tensors in place; grouped heads share a key head; tiles above the diagonal are left alone
    sc = torch.full((B_, H, S, S), float("nan"), dtype=torch.float32, device=DEV)
    _lib.check(lib.kg_matmul_nt_heads(dtc, q.data_ptr(), k.data_ptr(), sc.data_ptr(), S, S, hd, qd, kvd, S, B_ * H, S * qd, S * kvd, H * S * S,
                                      H, hd, hd, S * S, rep, 0, 1, 1, 1.0, st()), "scores")
    ref = torch.einsum("bshd,bthd->bhst", q.float(), k.float().repeat_interleave(rep, dim=2))
    assert float(((sc - ref)[:, :, tril]).abs().max()) < 1e-3
    assert bool(torch.isnan(sc[:, :, 0, 256:]).all())                    # the tiles entirely above the diagonal were skipped
    # probabilities times values: lower-triangular A, K walked to the diagonal only
    P = (torch.rand(B_, H, S, S, device=DEV, generator=g) * tril).to(dtype)
    vt = torch.randn(B_, kvd, S, device=DEV, generator=g).to(dtype)
    ctx = torch.empty(B_, S, H, hd, dtype=dtype, device=DEV)
    _lib.check(lib.kg_matmul_nt_heads(dtc, P.data_ptr(), vt.data_ptr(), ctx.data_ptr(), S, hd, S, S, S, qd, B_ * H, H * S * S, kvd * S, S * qd,
                                      H, S * S, hd * S, hd, rep, 0, 0, 2, 1.0, st()), "context")
    refc = torch.einsum("bhst,bhdt->bshd", P.float(), vt.float().view(B_, Hkv, hd, S).repeat_interleave(rep, dim=1))
    assert float((ctx.float() - refc).abs().max()) <= (4e-3 if dtype == torch.float16 else 2e-2) * float(refc.abs().max())
    # dV^T = dO^T . P over the rows of P (K-major B), only the queries at or after each key
    dOt = torch.randn(B_, qd, S, device=DEV, generator=g).to(dtype)
    for causal in (0, 4):
        dvt = torch.empty(B_, qd, S, dtype=torch.float32, device=DEV)
        _lib.check(lib.kg_matmul_nt_heads(dtc, dOt.data_ptr(), P.data_ptr(), dvt.data_ptr(), hd, S, S, S, S, S, B_ * H, qd * S, H * S * S, qd * S,
                                          H, hd * S, S * S, hd * S, 1, 2, 1, causal, 1.0, st()), "dV^T")
        refv = torch.einsum("bhds,bhst->bhdt", dOt.float().view(B_, H, hd, S), P.float()).reshape(B_, qd, S)
        assert float((dvt - refv).abs().max()) < 1e-3 * S ** 0.5, causal
    # transposes
    T_ = torch.randn(3, 70, 200, device=DEV, generator=g).to(dtype)
    out = torch.empty(3, 200, 70, dtype=dtype, device=DEV)
    _lib.check(lib.kg_transpose16(T_.data_ptr(), out.data_ptr(), 70, 200, 200, 70, 3, 70 * 200, 70 * 200, st()), "kg_transpose16")
    assert torch.equal(out, T_.transpose(1, 2))
    # whole 16-byte pieces on both sides (the fast form), ragged tile edges, rows inside wider rows
    Tw = torch.randn(2, 136, 328, device=DEV, generator=g).to(dtype)
    out = torch.zeros(2, 200, 136, dtype=dtype, device=DEV)
    _lib.check(lib.kg_transpose16(Tw[:, :, 64:].data_ptr(), out.data_ptr(), 136, 200, 328, 136, 2, 136 * 328, 200 * 136, st()), "kg_transpose16")
    assert torch.equal(out, Tw[:, :, 64:264].transpose(1, 2))


def test_bad_shapes_are_refused():
    lib = _lib.lib()
    A = torch.zeros(16, 12, dtype=torch.float16, device=DEV); B = torch.zeros(16, 12, dtype=torch.float16, device=DEV)
    out = torch.zeros(16, 16, dtype=torch.float16, device=DEV)
    assert lib.kg_matmul_nt(0, A.data_ptr(), B.data_ptr(), out.data_ptr(), 16, 16, 12, 12, 12, 16, 1, 0, 0, 0, 0, 0, 0, 1.0, st()) == -1       # K % 8
    assert b"multiples of 8" in lib.kg_last_error()
    assert lib.kg_matmul_nt(0, A.data_ptr(), B.data_ptr(), out.data_ptr(), 16, 16, 8, 12, 12, 16, 1, 0, 0, 0, 0, 0, 4, 1.0, st()) == -1        # causal 4 without a K-major B
    assert lib.kg_matmul_nt(2, A.data_ptr(), B.data_ptr(), out.data_ptr(), 16, 16, 8, 16, 16, 16, 1, 0, 0, 0, 0, 0, 0, 1.0, st()) == -1        # dtype
