"""In-kernel timeline of the MFMA attention kernel (needs a build with the timeline stamps)."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd import _lib
lib = _lib.lib()
H, Hkv, hd, t_max, nsplit = 32, 32, 128, 2048, 16
dt = torch.float16
kc = torch.randn(Hkv, t_max, hd, device="cuda").to(dt); vc = torch.randn(Hkv, hd, t_max, device="cuda").to(dt)
vp = lambda t: C.c_void_p(t.data_ptr())
scr = torch.zeros(lib.kg_attention_scratch_bytes(H, hd, nsplit), dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
raw = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kangaroo_amd", "libkangaroo_hip.so"))
for rows, L in [(7, 128), (7, 256), (7, 512), (7, 2000), (1, 256)]:
    q = torch.randn(rows, H * hd, device="cuda").to(dt); o = torch.empty_like(q)
    raw.kg_attn_tlog_clear()
    for _ in range(1):
        _lib.check(lib.kg_attention(_lib.DT_F16, vp(q), rows, L - rows, vp(kc), vp(vc), H, Hkv, hd, t_max, 1, nsplit, vp(o), vp(scr), st), "attn")
    torch.cuda.synchronize()
    n = H * nsplit
    buf = np.zeros(n * 8, dtype=np.uint64)
    raw.kg_attn_tlog_read(buf.ctypes.data_as(C.c_void_p), n * 8)
    t = buf.reshape(nsplit, H, 8).astype(np.float64) * 0.01      # us (100 MHz)
    nact = (L + 127) // 128
    act = t[:nact]                                               # [chunk][head][stamp]
    t0 = act[:, :, 0].min()
    def rel(i, f=np.mean): 
        v = act[:, :, i]; v = v[v > 0]
        return f(v - t0) if v.size else float("nan")
    print(f"rows {rows} T {L} chunks {nact}: entry {rel(0):.2f} | params known {rel(1):.2f} | q requested {rel(2):.2f} | first tile requested {rel(3):.2f} | tiles computed {rel(4):.2f} max {rel(4, np.max):.2f}", flush=True)
