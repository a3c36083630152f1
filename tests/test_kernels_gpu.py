"""Kernel-level parity (GPU): each C-ABI kernel entry point against fp32 torch math on the
same fp16/bf16-rounded inputs, and against the oracle's layer functions.

Tolerances: outputs are stored in the 16-bit model dtype; the bound is 1e-2 relative to
the output's max-abs (the tolerance the task states for logits), with most cases far
inside it."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kangaroo_amd import _lib
from kangaroo_amd.packing import pack_weight, pack_qkv, pack_gate_up, rope_tables
from oracle import kangaroo_oracle as ko

DEV = "cuda"


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


DTYPES = [(torch.float16, _lib.DT_F16), (torch.bfloat16, _lib.DT_BF16)]


@pytest.mark.parametrize("dtype,code", DTYPES, ids=["f16", "bf16"])
@pytest.mark.parametrize("N,K", [(64, 64), (100, 96), (4096, 4096), (1024, 11008), (2000, 2048)])
@pytest.mark.parametrize("rows", [1, 2, 7, 16, 17, 40, 64, 100, 128])      # > 16: the prefill GEMM kernel
def test_linear_store(dtype, code, N, K, rows):
    lib = _lib.lib()
    g = torch.Generator().manual_seed(N * 7 + K + rows)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    x = torch.randn(rows, K, generator=g).to(dtype)
    ref = x.float() @ w.float().T
    wp = pack_weight(w.to(DEV))
    xd = x.to(DEV)
    out = torch.full((rows, N), float("nan"), dtype=dtype, device=DEV)
    scratch = torch.zeros(lib.kg_linear_scratch_bytes(rows, K), dtype=torch.uint8, device=DEV)
    for nblocks, nwaves in [(0, 0), (3, 4), (1000, 2)]:
        out.fill_(float("nan"))
        _lib.check(lib.kg_linear(code, vp(wp), N, K, vp(xd), rows, 0, None, 0.0, 0, None, vp(out), N, vp(scratch), nblocks, nwaves, stream()), "kg_linear")
        torch.cuda.synchronize()
        tol = 4e-3 if dtype == torch.float16 else 1.6e-2
        assert rel_err(out, ref) < tol, (nblocks, nwaves)


@pytest.mark.parametrize("dtype,code", DTYPES, ids=["f16", "bf16"])
@pytest.mark.parametrize("rows", [1, 5, 33, 64, 128])
def test_linear_rmsnorm_resid_silu(dtype, code, rows):
    lib = _lib.lib()
    d, F = 256, 704
    g = torch.Generator().manual_seed(5 + rows)
    x = torch.randn(rows, d, generator=g).to(dtype)
    nw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
    wg = (torch.randn(F, d, generator=g) / math.sqrt(d)).to(dtype)
    wu = (torch.randn(F, d, generator=g) / math.sqrt(d)).to(dtype)
    wd = (torch.randn(d, F, generator=g) / math.sqrt(F)).to(dtype)
    scratch = torch.zeros(lib.kg_linear_scratch_bytes(rows, F), dtype=torch.uint8, device=DEV)
    # gate/up with RMSNorm prologue and SiLU*mul epilogue
    xn = ko.rmsnorm(x.float(), nw.float(), 1e-6, round_to=dtype)
    act_ref = torch.nn.functional.silu(xn @ wg.float().T) * (xn @ wu.float().T)
    act = torch.empty((rows, F), dtype=dtype, device=DEV)
    wgu_d, x_d, nw_d, wd_d = pack_gate_up(wg.to(DEV), wu.to(DEV)), x.to(DEV), nw.to(DEV), pack_weight(wd.to(DEV))   # keep alive
    _lib.check(lib.kg_linear(code, vp(wgu_d), 2 * F, d, vp(x_d), rows, 1, vp(nw_d), 1e-6,
                             3, None, vp(act), F, vp(scratch), 0, 0, stream()), "kg_linear")
    torch.cuda.synchronize()
    tol = 6e-3 if dtype == torch.float16 else 3e-2
    assert rel_err(act, act_ref) < tol
    # down projection with residual epilogue
    y_ref = x.float() + act.float().cpu() @ wd.float().T
    y = torch.empty((rows, d), dtype=dtype, device=DEV)
    _lib.check(lib.kg_linear(code, vp(wd_d), d, F, vp(act), rows, 0, None, 0.0, 1, vp(x_d), vp(y), d,
                             vp(scratch), 0, 0, stream()), "kg_linear")
    torch.cuda.synchronize()
    assert rel_err(y, y_ref) < tol


@pytest.mark.parametrize("dtype,code", DTYPES, ids=["f16", "bf16"])
@pytest.mark.parametrize("xscale,gscale", [(1e-3, 0.02), (1e-3, 3.0), (1e3, 0.02), (1e3, 3.0)])
@pytest.mark.parametrize("rows", [1, 7, 40])
def test_rmsnorm_prologue_keeps_the_reference_rounding_points(dtype, code, xscale, gscale, rows):
    """RMSNorm as kangaroo/adapter.py:254-259 orders it: x * rsqrt(mean x^2 + eps) is rounded to the model dtype
    BEFORE the product with gamma, so activations thousands of times larger than gamma (or embeddings of 1e-3
    under a gamma of 0.02) neither leave the fp16 range nor sink into its subnormals on the way.  Checked
    against the oracle's rmsnorm with the same rounding points followed by an fp32 linear."""
    lib = _lib.lib()
    d, N = 4096, 256
    g = torch.Generator().manual_seed(rows + int(gscale * 10))
    x = (torch.randn(rows, d, generator=g) * xscale).to(dtype)
    x[:, 5] = x[:, 5] * 10                                   # one outlier channel, as real checkpoints have
    nw = (gscale * (1 + 0.1 * torch.randn(d, generator=g))).to(dtype)
    w = (torch.randn(N, d, generator=g) / math.sqrt(d)).to(dtype)
    assert torch.isfinite(x.float()).all()
    xn = ko.rmsnorm(x.float(), nw.float(), 1e-6, round_to=dtype)
    ref = xn @ w.float().T
    out = torch.full((rows, N), float("nan"), dtype=dtype, device=DEV)
    wp, xd, nwd = pack_weight(w.to(DEV)), x.to(DEV), nw.to(DEV)
    scratch = torch.zeros(lib.kg_linear_scratch_bytes(rows, d), dtype=torch.uint8, device=DEV)
    _lib.check(lib.kg_linear(code, vp(wp), N, d, vp(xd), rows, 1, vp(nwd), 1e-6, 0, None, vp(out), N, vp(scratch), 0, 0, stream()), "kg_linear")
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, ref) < (4e-3 if dtype == torch.float16 else 1.6e-2)


@pytest.mark.parametrize("dtype,code", DTYPES, ids=["f16", "bf16"])
def test_sixteen_row_pass_with_x_streamed_through_wave_slots(dtype, code):
    """16 rows x K = 4096 with the residual epilogue and 8 waves: X does not fit LDS beside two blocks per CU,
    so the kernel streams the X columns of each weight batch through per-wave LDS slots (about 149 KiB of
    dynamic LDS, which needs the opt-in the library sets at start-up)."""
    lib = _lib.lib()
    N = K = 4096
    rows = 16
    g = torch.Generator().manual_seed(16)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype)
    x = torch.randn(rows, K, generator=g).to(dtype)
    r = torch.randn(rows, N, generator=g).to(dtype)
    ref = r.float() + x.float() @ w.float().T
    wp, xd, rd = pack_weight(w.to(DEV)), x.to(DEV), r.to(DEV)
    out = torch.full((rows, N), float("nan"), dtype=dtype, device=DEV)
    scratch = torch.zeros(256, dtype=torch.uint8, device=DEV)
    _lib.check(lib.kg_linear(code, vp(wp), N, K, vp(xd), rows, 0, None, 0.0, 1, vp(rd), vp(out), N, vp(scratch), 256, 8, stream()), "kg_linear")
    torch.cuda.synchronize()
    assert rel_err(out, ref) < (4e-3 if dtype == torch.float16 else 1.6e-2)


@pytest.mark.parametrize("dtype,code", DTYPES, ids=["f16", "bf16"])
@pytest.mark.parametrize("H,Hkv,hd", [(4, 2, 16), (8, 8, 32), (4, 1, 64), (32, 32, 128)])
@pytest.mark.parametrize("rows,pos0", [(1, 0), (1, 37), (2, 5), (7, 130), (16, 0), (16, 48), (40, 3), (64, 100), (100, 20), (128, 100)])
def test_qkv_rope_attention(dtype, code, H, Hkv, hd, rows, pos0):
    """Fused RMSNorm+QKV+RoPE+append, then attention, against the oracle's attention()."""
    lib = _lib.lib()
    d = H * hd
    t_max = 256
    shape = ko.Shape(d, 1, H, Hkv, 4 * d, 100, 1e-6, 2048)
    g = torch.Generator().manual_seed(H * 100 + hd + rows + pos0)
    r = lambda *s: torch.randn(*s, generator=g)
    wq, wk, wv, wo = [(r(n, d) / math.sqrt(d)).to(dtype) for n in (H * hd, Hkv * hd, Hkv * hd, d)]
    nw = (1 + 0.1 * r(d)).to(dtype)
    x = r(rows, d).to(dtype)
    past_k = r(Hkv, pos0, hd).to(dtype)
    past_v = r(Hkv, pos0, hd).to(dtype)
    tables = ko.rope_tables(hd, 2048)
    # oracle: same rounding points as the device dtype
    xn = ko.rmsnorm(x.float(), nw.float(), 1e-6, round_to=dtype)
    pos = torch.arange(pos0, pos0 + rows)
    eye = torch.eye(H * hd)
    attn_ref, (k_ref, v_ref) = ko.attention(xn, pos, wq.float(), wk.float(), wv.float(), eye, (past_k.float(), past_v.float()) if pos0 else None,
                                            shape, tables, round_to=None)
    vt = 1 if hd >= 32 else 0                     # transposed V cache + MFMA attention
    kc = torch.zeros((Hkv, t_max, hd), dtype=dtype, device=DEV)
    vc = torch.zeros((Hkv, hd, t_max) if vt else (Hkv, t_max, hd), dtype=dtype, device=DEV)
    vrows = (lambda n0, n1: vc[:, :, n0:n1].transpose(-1, -2)) if vt else (lambda n0, n1: vc[:, n0:n1])
    kc[:, :pos0] = past_k.to(DEV)
    vrows(0, pos0).copy_(past_v.to(DEV))
    cos, sin = rope_tables(hd, 2048, device=DEV)
    q = torch.empty((rows, H * hd), dtype=dtype, device=DEV)
    scratch = torch.zeros(lib.kg_linear_scratch_bytes(rows, d), dtype=torch.uint8, device=DEV)
    wqkv_d, x_d, nw_d = pack_qkv(wq.to(DEV), wk.to(DEV), wv.to(DEV), H, Hkv, hd), x.to(DEV), nw.to(DEV)   # keep alive
    _lib.check(lib.kg_qkv_rope_append(code, vp(wqkv_d), d, H, Hkv, hd, vp(x_d), rows, pos0,
                                      vp(nw_d), 1e-6, vp(cos), vp(sin), vp(q), vp(kc), vp(vc), t_max, vt, vp(scratch), 0, 0, stream()),
               "kg_qkv_rope_append")
    torch.cuda.synchronize()
    tol = 6e-3 if dtype == torch.float16 else 3e-2
    assert rel_err(kc[:, :pos0 + rows], k_ref) < tol
    assert rel_err(vrows(0, pos0 + rows), v_ref) < tol
    assert torch.equal(kc[:, :pos0].cpu(), past_k) and float(kc[:, pos0 + rows:].abs().max()) == 0.0   # only the new slots written
    if rows > 16 and not vt:
        return                                        # the head_dim-16 attention fallback takes <= 16 rows
    for nsplit in (1, 4):
        ascr = torch.zeros(lib.kg_attention_scratch_bytes(H, hd, nsplit), dtype=torch.uint8, device=DEV)
        o = torch.full((rows, H * hd), float("nan"), dtype=dtype, device=DEV)
        _lib.check(lib.kg_attention(code, vp(q), rows, pos0, vp(kc), vp(vc), H, Hkv, hd, t_max, vt, nsplit, vp(o), vp(ascr), stream()), "kg_attention")
        torch.cuda.synchronize()
        assert rel_err(o, attn_ref) < (1e-2 if dtype == torch.float16 else 5e-2), nsplit
    # first position read on the device (the int32 at scratch + 128): the same bits as with the position in the call,
    # and a position that would leave the cache turns the pass into a no-op
    POS_IN_SCRATCH = -2
    kc2, vc2, q2 = kc.clone(), vc.clone(), torch.empty_like(q)
    kc2[:, pos0:] = 0
    (vc2[:, :, pos0:] if vt else vc2[:, pos0:]).zero_()
    slot, aslot = scratch[128:132].view(torch.int32), ascr[128:132].view(torch.int32)
    for first, live in ((pos0, True), (t_max - rows + 1, False)):
        slot.fill_(first); aslot.fill_(first)
        before = (kc2.clone(), vc2.clone())
        _lib.check(lib.kg_qkv_rope_append(code, vp(wqkv_d), d, H, Hkv, hd, vp(x_d), rows, POS_IN_SCRATCH,
                                          vp(nw_d), 1e-6, vp(cos), vp(sin), vp(q2), vp(kc2), vp(vc2), t_max, vt, vp(scratch), 0, 0, stream()),
                   "kg_qkv_rope_append")
        o2 = torch.full((rows, H * hd), float("nan"), dtype=dtype, device=DEV)
        _lib.check(lib.kg_attention(code, vp(q2), rows, POS_IN_SCRATCH, vp(kc2), vp(vc2), H, Hkv, hd, t_max, vt, nsplit, vp(o2), vp(ascr), stream()), "kg_attention")
        torch.cuda.synchronize()
        if live:
            assert torch.equal(q2, q) and torch.equal(kc2, kc) and torch.equal(vc2, vc) and torch.equal(o2, o)
        else:
            assert torch.equal(kc2, before[0]) and torch.equal(vc2, before[1]) and bool(torch.isnan(o2.float()).all())


@pytest.mark.parametrize("dtype,code", DTYPES, ids=["f16", "bf16"])
@pytest.mark.parametrize("V,d", [(97, 64), (1000, 256), (32000, 4096)])
@pytest.mark.parametrize("rows", [1, 7])
def test_head_argmax_pmax(dtype, code, V, d, rows):
    lib = _lib.lib()
    g = torch.Generator().manual_seed(V + rows)
    w = (torch.randn(V, d, generator=g) * (8.0 / math.sqrt(d))).to(dtype)
    x = torch.randn(rows, d, generator=g).to(dtype)
    nw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype)
    w_d, x_d, nw_d = pack_weight(w.to(DEV)), x.to(DEV), nw.to(DEV)   # keep alive
    for pro in (0, 1):
        xin = x.float() if pro == 0 else ko.rmsnorm(x.float(), nw.float(), 1e-6, round_to=dtype)
        ref = (xin @ w.float().T).to(dtype).float()             # logits as the reference stores them
        logits = torch.empty((rows, V), dtype=dtype, device=DEV)
        am = torch.empty(rows, dtype=torch.int32, device=DEV)
        pm = torch.empty(rows, dtype=torch.float32, device=DEV)
        scratch = torch.zeros(lib.kg_head_scratch_bytes(V), dtype=torch.uint8, device=DEV)
        _lib.check(lib.kg_head(code, vp(w_d), V, d, vp(x_d), rows, pro, vp(nw_d) if pro else None, 1e-6,
                               vp(logits), vp(am), vp(pm), vp(scratch), 0, 0, stream()), "kg_head")
        torch.cuda.synchronize()
        assert rel_err(logits, ref) < (4e-3 if dtype == torch.float16 else 2e-2)
        # argmax / p_max are functions of the device's own stored logits: exact / 1e-5
        lg = logits.float().cpu()
        assert am.cpu().tolist() == lg.argmax(-1).tolist()
        np.testing.assert_allclose(pm.cpu().numpy(), lg.softmax(-1).max(-1).values.numpy(), rtol=2e-5)


def test_bad_arguments_are_refused():
    lib = _lib.lib()
    t = torch.zeros(1024, dtype=torch.float16, device=DEV)
    assert lib.kg_linear(0, vp(t), 16, 48, vp(t), 1, 0, None, 0.0, 0, None, vp(t), 16, vp(t), 0, 0, stream()) != 0   # K % 32
    assert lib.kg_linear(0, vp(t), 16, 64, vp(t), 129, 0, None, 0.0, 0, None, vp(t), 16, vp(t), 0, 0, stream()) != 0  # rows
    assert lib.kg_linear(0, vp(t), 16, 64, vp(t), 1, 1, None, 0.0, 0, None, vp(t), 16, vp(t), 0, 0, stream()) != 0   # norm_w
    assert b"" != lib.kg_last_error()
