"""kg_matmul_nt against torch.matmul (hipBLASLt): results and time at the shapes of the adapter training step and
the prompt pass.  python scripts/mm_bench.py [--quick]"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd import _lib
from kangaroo_amd.packing import pack_weight

lib = _lib.lib()
dev = "cuda"


def mm(A, B, out_f32=False, packed=None, causal=0, alpha=1.0, N=None):
    """A [.., M, K], B [.., N, K] (or packed, with N given) -> C [.., M, N]"""
    dt = _lib.DT_F16 if A.dtype == torch.float16 else _lib.DT_BF16
    batch = A.shape[0] if A.dim() == 3 else 1
    M, K = A.shape[-2], A.shape[-1]
    N = N if packed is not None else B.shape[-2]
    Cc = torch.empty((batch, M, N) if A.dim() == 3 else (M, N), dtype=torch.float32 if out_f32 else A.dtype, device=A.device)
    Bp = packed if packed is not None else B
    _lib.check(lib.kg_matmul_nt(dt, A.data_ptr(), Bp.data_ptr(), Cc.data_ptr(), M, N, K, A.stride(-2), 0 if packed is not None else B.stride(-2), N,
                                batch, A.stride(0) if A.dim() == 3 else 0, B.stride(0) if (packed is None and B.dim() == 3) else 0, M * N if A.dim() == 3 else 0,
                                1 if packed is not None else 0, 1 if out_f32 else 0, causal, alpha, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "kg_matmul_nt")
    return Cc


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        f()
    t1.record(); torch.cuda.synchronize()
    return t0.elapsed_time(t1) / n


g = torch.Generator(device=dev).manual_seed(0)
# ---- results ----
for dt in (torch.float16, torch.bfloat16):
    for (M, N, K) in ((128, 128, 64), (200, 136, 72), (1000, 520, 4096), (77, 4096, 256), (257, 129, 8)):
        A = torch.randn(M, K, device=dev, generator=g).to(dt); B = torch.randn(N, K, device=dev, generator=g).to(dt)
        ref = A.float() @ B.float().T
        for f32 in (False, True):
            got = mm(A, B, out_f32=f32).float()
            err = float((got - ref).abs().max()) / float(ref.abs().max())
            assert err < (1e-5 if f32 else (2e-3 if dt == torch.float16 else 1.2e-2)), (dt, M, N, K, f32, err)
    W = torch.randn(528, 4096, device=dev, generator=g).to(dt); X = torch.randn(300, 4096, device=dev, generator=g).to(dt)
    got = mm(X, None, out_f32=True, packed=pack_weight(W), N=528)
    assert float((got - X.float() @ W.float().T).abs().max()) < 1e-4 * 4096 ** 0.5 * 4
    assert torch.equal(got, mm(X, W, out_f32=True))          # same products, same order: the packed layout only changes addresses
    # batched, causal
    Q = torch.randn(6, 300, 128, device=dev, generator=g).to(dt); Kk = torch.randn(6, 300, 128, device=dev, generator=g).to(dt)
    S = mm(Q, Kk, out_f32=True, causal=1, alpha=0.25)
    ref = 0.25 * (Q.float() @ Kk.float().transpose(1, 2))
    mask = torch.ones(300, 300, device=dev).tril().bool()
    assert float(((S - ref) * mask).abs().max()) < 1e-3
    P = (torch.rand(6, 300, 300, device=dev, generator=g) * mask).to(dt); Vt = torch.randn(6, 128, 300, device=dev, generator=g).to(dt)
    pad = lambda t: torch.nn.functional.pad(t, (0, 4))         # K = 304: a multiple of 8
    O = mm(pad(P), pad(Vt), out_f32=True, causal=2)
    assert float((O - P.float() @ Vt.float().transpose(1, 2)).abs().max()) < 1e-3 * 300 ** 0.5
    # K-major B: dV^T[hd, S_k] = dO^T[hd, S_q] . P[S_q, S_k], only queries at or after each key (causal 4)
    Sq = 296
    Pm = (torch.rand(6, Sq, Sq, device=dev, generator=g) * torch.ones(Sq, Sq, device=dev).tril()).to(dt); dOt = torch.randn(6, 128, Sq, device=dev, generator=g).to(dt)
    for caus in (0, 4):
        out = torch.empty(6, 128, Sq, dtype=torch.float32, device=dev)
        _lib.check(lib.kg_matmul_nt(dtc := (_lib.DT_F16 if dt == torch.float16 else _lib.DT_BF16), dOt.data_ptr(), Pm.data_ptr(), out.data_ptr(), 128, Sq, Sq, Sq, Sq, Sq,
                                    6, 128 * Sq, Sq * Sq, 128 * Sq, 2, 1, caus, 1.0, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "kg_matmul_nt (K-major B)")
        assert float((out - dOt.float() @ Pm.float()).abs().max()) < 1e-3 * Sq ** 0.5, caus
    Bk = torch.randn(520, 136, device=dev, generator=g).to(dt); Ak = torch.randn(77, 520, device=dev, generator=g).to(dt)
    out = torch.empty(77, 136, dtype=torch.float32, device=dev)
    _lib.check(lib.kg_matmul_nt(dtc, Ak.data_ptr(), Bk.data_ptr(), out.data_ptr(), 77, 136, 520, 520, 136, 136, 1, 0, 0, 0, 2, 1, 0, 1.0, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "kg_matmul_nt (K-major B)")
    assert float((out - Ak.float() @ Bk.float()).abs().max()) < 1e-3 * 520 ** 0.5
    T_ = torch.randn(3, 70, 200, device=dev, generator=g).to(dt)
    out = torch.empty(3, 200, 70, dtype=dt, device=dev)
    _lib.check(lib.kg_transpose16(T_.data_ptr(), out.data_ptr(), 70, 200, 200, 70, 3, 70 * 200, 70 * 200, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "kg_transpose16")
    assert torch.equal(out, T_.transpose(1, 2))
print("results ok", flush=True)
if "--quick" in sys.argv:
    sys.exit(0)
# ---- time ----
dt = torch.float16
for name, (M, N, K), batch in (("projection 8192x4096x4096", (8192, 4096, 4096), 1), ("head 8192x32000x4096", (8192, 32000, 4096), 1),
                               ("dhn 8192x4096x32000", (8192, 4096, 32000), 1), ("weight grad 4096x4096x8192", (4096, 4096, 8192), 1),
                               ("prefill 256 rows x (22016 x 4096)", (256, 22016, 4096), 1), ("prefill 640 rows x (22016 x 4096)", (640, 22016, 4096), 1),
                               ("prefill 640 x (12288 x 4096)", (640, 12288, 4096), 1), ("prefill 640 x (4096 x 4096)", (640, 4096, 4096), 1), ("prefill 640 x (4096 x 11008)", (640, 4096, 11008), 1),
                               ("prefill 1500 x (22016 x 4096)", (1500, 22016, 4096), 1), ("prefill 1500 x (12288 x 4096)", (1500, 12288, 4096), 1), ("prefill 1500 x (4096 x 11008)", (1500, 4096, 11008), 1),
                               ("scores 128 x (2048x2048x128)", (2048, 2048, 128), 128), ("context 128 x (2048x128x2048)", (2048, 128, 2048), 128)):
    shp = (batch, M, K) if batch > 1 else (M, K)
    A = (torch.rand(shp, device=dev, generator=g) * 2 - 1).to(dt)
    B = (torch.rand((batch, N, K) if batch > 1 else (N, K), device=dev, generator=g) * 2 - 1).to(dt)
    fl = 2.0 * M * N * K * batch
    Bp = pack_weight(B) if (batch == 1 and name.startswith("prefill")) else None     # the prompt pass reads the decode path's packed weights
    res = {}
    for rnd in range(2):                                   # interleaved rounds in one process; the better of the two
        for tile in ("1", "2", "3", "4"):                  # 128 x 128, 256 x 256 register-staged, 256 x 256 and 128 x 256 direct-to-LDS
            os.environ["KG_MM_TILE"] = tile
            res[tile] = min(res.get(tile, 1e9), timeit(lambda: mm(A, B, packed=Bp, N=N)))
        res["lib"] = min(res.get("lib", 1e9), timeit(lambda: torch.matmul(A, B.transpose(-1, -2))))
    os.environ.pop("KG_MM_TILE")
    t_auto = timeit(lambda: mm(A, B, packed=Bp, N=N))
    tf = lambda t: fl / t / 1e9
    print(f"{name:40s} 128^2 {tf(res['1']):6.0f}  256^2 reg {tf(res['2']):6.0f}  256^2 lds {tf(res['3']):6.0f}  128x256 lds {tf(res['4']):6.0f}  chosen {tf(t_auto):6.0f} ({t_auto * 1e3:7.1f} us)   library {tf(res['lib']):6.0f} TFLOP/s", flush=True)
