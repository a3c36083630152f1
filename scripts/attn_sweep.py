"""Attention kernel time vs context length (run under rocprofv3 --kernel-trace; durations come from the trace).
Each configuration launches the kernel 20 times; the trace is grouped by launch order."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd import _lib
lib = _lib.lib()
H, Hkv, hd, t_max = 32, 32, 128, 2048
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dt = torch.float16
kc = torch.randn(Hkv, t_max, hd, device="cuda").to(dt)
vc = torch.randn(Hkv, hd, t_max, device="cuda").to(dt)
vp = lambda t: C.c_void_p(t.data_ptr())
scr = torch.zeros(lib.kg_attention_scratch_bytes(H, hd, nsplit), dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
cfgs = [(rows, L) for rows in (1, 7) for L in (64, 128, 256, 512, 1024, 1536, 2000)]
for rows, L in cfgs:
    q = torch.randn(rows, H * hd, device="cuda").to(dt)
    o = torch.empty_like(q)
    for _ in range(20):
        _lib.check(lib.kg_attention(_lib.DT_F16, vp(q), rows, L - rows, vp(kc), vp(vc), H, Hkv, hd, t_max, 1, nsplit, vp(o), vp(scr), st), "attn")
    torch.cuda.synchronize()
print("configs:", cfgs)
