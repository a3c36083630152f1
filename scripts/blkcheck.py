import ctypes as C, math, sys, os
sys.path.insert(0, "/root/repo")
import torch
from kangaroo_amd import _lib
from kangaroo_amd.packing import pack_weight
lib = _lib.lib()
vp = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(1)
for (N, K) in [(11264, 2048), (2048, 5632), (2560, 2048), (2048, 2048), (32000, 2048)]:
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).half().cuda()
    wp = pack_weight(w)
    for rows in (1, 7):
        x = torch.randn(rows, K, generator=g).half().cuda()
        scratch = torch.zeros(256, dtype=torch.uint8, device="cuda")
        outs = {}
        for nw in (2, 4, 8):
            for nb in (128, 160, 235, 250, 256, 512, 704):
                out = torch.full((rows, N), float("nan"), dtype=torch.float16, device="cuda")
                rc = lib.kg_linear(_lib.DT_F16, vp(wp), N, K, vp(x), rows, 0, None, 0.0, 0, None, vp(out), N, vp(scratch), nb, nw, st)
                torch.cuda.synchronize()
                if rc != 0: continue
                outs[(nw, nb)] = out
            ref = outs[(nw, 256)]
            bad = [nb for (w_, nb), o in outs.items() if w_ == nw and not torch.equal(o, ref)]
            print(N, K, "rows", rows, "waves", nw, "blocks differing from 256:", bad, flush=True)
