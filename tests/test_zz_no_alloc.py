"""kg_head_noalloc (GPU): the lm_head + argmax + confidence entry point that allocates nothing, against kg_head
with a caller-provided scratch buffer.  Same kernels, same launch geometry: logits, argmax and confidence must
be bit-identical.  Two calls in a row share the built-in scratch array; the second is smaller (fewer rows, a
smaller vocabulary), so stale records of the first call lie beyond what it reads."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from kangaroo_amd import _lib
from kangaroo_amd.packing import pack_weight

DEV = "cuda"


def vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def head_both(lib, code, dtype, V, d, rows, seed, pro):
    g = torch.Generator().manual_seed(seed)
    w = pack_weight((torch.randn(V, d, generator=g) * (6.0 / math.sqrt(d))).to(dtype).to(DEV))
    x = torch.randn(rows, d, generator=g).to(dtype).to(DEV)
    nw = (1 + 0.1 * torch.randn(d, generator=g)).to(dtype).to(DEV)
    res = []
    for noalloc in (False, True):
        logits = torch.full((rows, V), float("nan"), dtype=dtype, device=DEV)
        am = torch.full((rows,), -1, dtype=torch.int32, device=DEV)
        pm = torch.full((rows,), -1.0, dtype=torch.float32, device=DEV)
        if noalloc:
            before = torch.cuda.memory_allocated()
            _lib.check(lib.kg_head_noalloc(code, vp(w), V, d, vp(x), rows, pro, vp(nw) if pro else None, 1e-6,
                                           vp(logits), vp(am), vp(pm), 0, 0, stream()), "kg_head_noalloc")
            assert torch.cuda.memory_allocated() == before
        else:
            scratch = torch.zeros(lib.kg_head_scratch_bytes(V), dtype=torch.uint8, device=DEV)
            _lib.check(lib.kg_head(code, vp(w), V, d, vp(x), rows, pro, vp(nw) if pro else None, 1e-6,
                                   vp(logits), vp(am), vp(pm), vp(scratch), 0, 0, stream()), "kg_head")
        torch.cuda.synchronize()
        res.append((logits.cpu(), am.cpu(), pm.cpu()))
    return res


@pytest.mark.parametrize("dtype,code", [(torch.float16, _lib.DT_F16), (torch.bfloat16, _lib.DT_BF16)], ids=["f16", "bf16"])
def test_noalloc_head_matches_the_scratch_argument_form(dtype, code):
    lib = _lib.lib()
    assert lib.kg_head_noalloc_init() == 0
    assert lib.kg_head_noalloc_max_vocab() == 65536
    # first call: 7 rows at the full vocabulary; second: 1 row at a smaller one, then 3 rows in between
    for V, d, rows, pro in ((32000, 4096, 7, 1), (1000, 256, 1, 0), (5000, 512, 3, 1)):
        (l0, a0, p0), (l1, a1, p1) = head_both(lib, code, dtype, V, d, rows, V + rows, pro)
        assert torch.equal(l0.view(torch.int16), l1.view(torch.int16))
        assert torch.equal(a0, a1) and torch.equal(p0, p1)
        assert a0.tolist() == l0.float().argmax(-1).tolist()
    # beyond the built-in scratch: refused, nothing launched
    t = torch.zeros(64, dtype=dtype, device=DEV)
    i = torch.zeros(4, dtype=torch.int32, device=DEV)
    f = torch.zeros(4, dtype=torch.float32, device=DEV)
    assert lib.kg_head_noalloc(code, vp(t), 65537, 32, vp(t), 1, 0, None, 0.0, None, vp(i), vp(f), 0, 0, stream()) == -1
    assert b"65536" in lib.kg_last_error()
