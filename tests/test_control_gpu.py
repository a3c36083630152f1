"""Kernel-level parity (GPU) of the loop's control kernels through their own C entry points:
kg_accept (accept / reject, rollback lengths, loop bound) against the oracle's accept block
(oracle/kangaroo_oracle.py accept_block = evaluation/inference_kangaroo.py:92-104 and the rules of
evaluation/inference_kangaroo_MTBench.py:356-389), and kg_draft_step (adapter rows, drafted token,
confidence, second early exit) against the statements of evaluation/inference_kangaroo.py:55-61,74,79-83.
Integer results: bit-exact.  The logits handed to the device are the stored 16-bit rows, and the oracle sees
exactly those values, so no tolerance is involved anywhere except the fp32 confidence (1e-5) and the typical
rule's fp32 posterior threshold (2e-4)."""
import ctypes as C
import struct

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from kangaroo_amd import _lib
from oracle import kangaroo_oracle as ko

DEV = "cuda"
V = 32000
I32P = C.POINTER(C.c_int32)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _scratch(lib):
    return torch.zeros(lib.kg_control_scratch_bytes(), dtype=torch.uint8, device=DEV)


def run_accept(lib, logits, drafted, s, eos, rule=0, top_k=1, top_p=0.0, temperature=1.0, steps=6, max_infer=2048, adapter_len=None, code=_lib.DT_F16,
               epsilon=0.3, delta=0.09):
    q = logits.shape[0]
    lg = logits.to(DEV).contiguous()
    dr = (C.c_int32 * max(1, q - 1))(*drafted)
    out = (C.c_int32 * 10)()
    scr = _scratch(lib)
    alen = s + q - 1 if adapter_len is None else adapter_len
    _lib.check(lib.kg_accept(code, C.c_void_p(lg.data_ptr()), logits.shape[1], q, C.cast(dr, I32P), s, eos, rule, top_k, top_p, temperature,
                             epsilon, delta, steps, max_infer, alen, C.cast(out, I32P), C.c_void_p(scr.data_ptr()), _stream()), "kg_accept")
    return list(out)


def oracle_accept(logits, drafted, s, eos, **kw):
    q = logits.shape[0]
    toks = torch.full((1, s + q + 2), -7, dtype=torch.long)
    toks[0, s + 1:s + q] = torch.tensor(drafted, dtype=torch.long)
    snew, stop = ko.accept_block(logits.float().unsqueeze(0), toks, s, s + q, eos, **kw)
    return snew - s, int(toks[0, snew]), int(stop), snew


def rows(q, seed, dtype=torch.float16, vocab=V):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(q, vocab, generator=g) * 3).to(dtype)


def distinct_rows(q, seed, scale=1.0):
    """fp16 rows whose V entries are pairwise different (a random choice among the finite fp16 values of
    magnitude [2^-14, 16)), so that sorting and top-k selection have one answer."""
    g = torch.Generator().manual_seed(seed)
    mags = torch.arange(0x0400, 0x4C00, dtype=torch.int32)              # bit patterns of 2^-14 .. just under 16
    pats = torch.cat([mags, mags | 0x8000])
    out = torch.empty(q, V, dtype=torch.float16)
    for i in range(q):
        pick = pats[torch.randperm(pats.numel(), generator=g)[:V]]
        out[i] = pick.to(torch.int16).view(torch.float16)
    return (out.float() * scale).to(torch.float16) if scale != 1.0 else out


@pytest.mark.parametrize("q", [1, 2, 4, 7])
@pytest.mark.parametrize("dtype,code", [(torch.float16, _lib.DT_F16), (torch.bfloat16, _lib.DT_BF16)], ids=["f16", "bf16"])
def test_greedy_accept_every_outcome(q, dtype, code):
    lib = _lib.lib()
    lg = rows(q, 100 + q, dtype)
    am = lg.float().argmax(-1).tolist()
    s = 37
    cases = [("full accept", am[:q - 1], -1)]
    for i in range(q - 1):
        d = list(am[:q - 1]); d[i] = (d[i] + 1) % V
        cases.append((f"reject at {i}", d, -1))
    for i in range(q):
        cases.append((f"eos at row {i}, drafts agree", am[:q - 1], am[i]))                      # :97 stops on a verify-side eos even when the draft agrees
        d = list(am[:q - 1])
        if q > 1:
            d[0] = (d[0] + 5) % V
        cases.append((f"eos at row {i}, reject at 0", d, am[i]))
    for name, drafted, eos in cases:
        got = run_accept(lib, lg, drafted, s, eos, code=code)
        want = oracle_accept(lg, drafted, s, eos)
        assert tuple(got[:4]) == want, (name, got, want)
        assert got[4] == min(s + q - 1, got[3])          # adapter cache: truncated to the new start index when longer (:114-118)
        assert got[6] == q - 1 and got[7] == 1


def test_argmax_ties_take_the_lowest_index():
    """torch.argmax returns the first maximum; so does the record merge (lowest index among equal maxima),
    wherever in the vocabulary (and in whichever record slice) the tied entries sit."""
    lib = _lib.lib()
    lg = rows(3, 7)
    top = float(lg.float().max()) + 1.0
    for a, b in ((5, 31999), (16000, 16001), (127, 128), (0, 1)):
        row = lg.clone()
        row[1, a] = top; row[1, b] = top
        drafted = [int(row[0].float().argmax()), a]
        got = run_accept(lib, row, [drafted[0], b], 10, -1)       # drafting the HIGHER index of the tie is a mismatch
        assert tuple(got[:4]) == oracle_accept(row, [drafted[0], b], 10, -1) and got[0] == 2 and got[1] == a
        got = run_accept(lib, row, drafted, 10, -1)
        assert tuple(got[:4]) == oracle_accept(row, drafted, 10, -1) and got[0] == 3


def test_loop_bound_and_adapter_length():
    """out[5]: the next round runs iff no eos and start < max_infer - 1 - steps (:46)."""
    lib = _lib.lib()
    lg = rows(3, 11)
    am = lg.float().argmax(-1).tolist()
    for max_infer, steps in ((64, 6), (48, 6), (47, 6), (46, 2), (42, 0)):
        got = run_accept(lib, lg, am[:2], 38, -1, steps=steps, max_infer=max_infer)
        assert got[3] == 41 and got[5] == int(41 < max_infer - 1 - steps), (max_infer, steps, got)
    got = run_accept(lib, lg, am[:2], 38, am[2])
    assert got[2] == 1 and got[5] == 0
    # adapter cache shorter than the new start index is left alone (full accept leaves it one short, :55)
    got = run_accept(lib, lg, am[:2], 38, -1, adapter_len=40)
    assert got[3] == 41 and got[4] == 40


@pytest.mark.parametrize("k", [2, 3, 10])
def test_top_k_membership(k):
    lib = _lib.lib()
    q = 6
    lg = distinct_rows(q, 200 + k)
    srt = lg.float().argsort(-1, descending=True)
    am = srt[:, 0].tolist()
    inside = [int(srt[i, k - 1]) for i in range(q - 1)]           # exactly the k-th largest: still a member
    outside = [int(srt[i, k]) for i in range(q - 1)]              # the (k+1)-th: not a member
    for name, drafted in (("k-th", inside), ("argmax", am[:q - 1]), ("(k+1)-th at 2", inside[:2] + [outside[2]] + inside[3:]), ("(k+1)-th at 0", outside)):
        got = run_accept(lib, lg, drafted, 20, -1, rule=1, top_k=k)
        want = oracle_accept(lg, drafted, 20, -1, accept_top_k=k)
        assert tuple(got[:4]) == want, (name, got, want)
    # eos among the verify argmaxes stops the round even when the drafted token is a member (:384 with the eos test of :97)
    got = run_accept(lib, lg, inside, 20, am[1], rule=1, top_k=k)
    assert tuple(got[:4]) == oracle_accept(lg, inside, 20, am[1], accept_top_k=k) and got[2] == 1 and got[0] == 2


def test_top_k_ties_at_the_boundary():
    """Equal logits at ranks k-1 and k (one inside, one outside the k largest): a drafted token is kept while
    fewer than k logits are strictly larger, so both tied tokens count as members.  torch.topk picks one of them
    without a documented order; the tie wholly INSIDE the k largest is compared with the oracle, the straddling
    one with the rule as stated."""
    lib = _lib.lib()
    k = 3
    lg = distinct_rows(2, 77)
    srt = lg[0].float().argsort(descending=True)
    row = lg.clone()
    row[0, srt[1]] = row[0, srt[2]]                               # ranks 1 and 2 tie, both inside the top 3
    for t in (int(srt[1]), int(srt[2])):
        got = run_accept(lib, row, [t], 5, -1, rule=1, top_k=k)
        assert tuple(got[:4]) == oracle_accept(row, [t], 5, -1, accept_top_k=k) and got[0] == 2
    row = lg.clone()
    row[0, srt[3]] = row[0, srt[2]]                               # ranks 2 and 3 tie across the boundary
    for t in (int(srt[2]), int(srt[3])):
        assert run_accept(lib, row, [t], 5, -1, rule=1, top_k=k)[0] == 2
    assert run_accept(lib, row, [int(srt[4])], 5, -1, rule=1, top_k=k)[0] == 1


@pytest.mark.parametrize("p,temp", [(0.8, 0.7), (0.3, 1.5), (1.0, 1.0), (0.05, 1.0)])
def test_top_p_membership_and_pooled_nuclei(p, temp):
    lib = _lib.lib()
    q = 5
    lg = distinct_rows(q, int(p * 100) + 3)
    f = lg.float()
    pr = torch.softmax(f / temp, -1)
    sp, si = pr.sort(-1, descending=True)
    cum = sp.cumsum(-1)
    # last member / first non-member of every row's own nucleus
    n_in = [(int((cum[i] <= p).sum()) + 1) for i in range(q)]
    last_in = [int(si[i, min(n_in[i], V) - 1]) for i in range(q - 1)]
    am = si[:, 0].tolist()
    sets = [("argmax", am[:q - 1]), ("last member", last_in)]
    if p < 1.0:
        first_out = [int(si[i, n_in[i]]) for i in range(q - 1)]
        sets.append(("first non-member at 1", [am[0], first_out[1]] + am[2:q - 1]))
        # a token outside its own row's nucleus but inside another row's: the reference pools the nuclei (:365)
        other = [int(si[(i + 1) % q, 0]) for i in range(q - 1)]
        sets.append(("other row's top token", other))
    for name, drafted in sets:
        got = run_accept(lib, lg, drafted, 9, -1, rule=2, top_p=p, temperature=temp)
        want = oracle_accept(lg, drafted, 9, -1, accept_top_p=p, temperature=temp)
        assert tuple(got[:4]) == want, (name, got, want)


def test_top_p_equal_probabilities():
    """Tokens with exactly equal logits: a stable descending sort keeps them in token-id order, which is the
    order the nucleus kernel assumes when it adds up the mass ahead of a token."""
    lib = _lib.lib()
    lg = torch.full((2, 1000), -8.0, dtype=torch.float16)
    lg[0, [10, 20, 30, 40]] = 4.0                                  # four tokens share ~all the mass: 0.25 each
    lg[1, 7] = 9.0
    for p, members in ((0.3, [10, 20]), (0.55, [10, 20, 30]), (0.8, [10, 20, 30, 40])):
        for t in (10, 20, 30, 40, 50):
            got = run_accept(lib, lg, [t], 3, -1, rule=2, top_p=p, temperature=1.0)
            want = oracle_accept(lg, [t], 3, -1, accept_top_p=p, temperature=1.0)
            assert tuple(got[:4]) == want, (p, t, got, want)
            assert (got[0] == 2) == (t in members), (p, t)


@pytest.mark.parametrize("temp,eps,delta", [(0.7, 0.3, 0.09), (1.0, 0.9, 1.5), (1.0, 0.02, 5.0), (1.3, 0.9, 40.0)])
def test_typical_rule_as_the_reference_executes_it(temp, eps, delta):
    """evaluation/inference_kangaroo_MTBench.py:293-314 through the oracle's typical_block: start_index moves by
    one per row that ends a run of matches, nothing is written, a failed posterior test or an eos stops, later
    rows are compared against the shifted start_index, and the draft threshold is rebound to the last
    min(epsilon, exp(-entropy) * delta)."""
    lib = _lib.lib()
    eos = 2
    for q, seed, scale in ((1, 1, 1.0), (4, 2, 1.0), (7, 3, 0.5), (7, 4, 0.25), (5, 5, 0.1)):
        lg = distinct_rows(q, 300 + seed, scale)
        am = lg.float().argmax(-1).tolist()
        sets = [am[:q - 1]]
        for i in range(q - 1):
            d = list(am[:q - 1]); d[i] = (d[i] + 3) % V
            sets.append(d)
        if q > 2:
            sets.append([am[1]] + am[1:q - 1])                  # after a shift, row 1's argmax is compared with ITS OWN slot's neighbour
        for drafted in sets:
            for max_infer in (2048, 31):
                s = 25
                got = run_accept(lib, lg, drafted, s, eos, rule=3, temperature=temp, epsilon=eps, delta=delta, max_infer=max_infer)
                toks = torch.full((1, s + 2 * q + 3), eos, dtype=torch.long)
                toks[0, s + 1:s + q] = torch.tensor(drafted, dtype=torch.long)
                snew, stop, thr = ko.typical_block(lg.float().unsqueeze(0), toks, s, s + q, eos, temp, eps, delta, max_infer)
                assert (got[0], got[1], got[2], got[3]) == (snew - s, int(toks[0, snew]), int(stop), snew), (q, drafted, got)
                dev_thr = struct.unpack("d", struct.pack("ii", got[8], got[9]))[0]
                # exp(-entropy): the entropy is an fp32 sum of 32000 terms, added in another order on the device
                assert thr is not None and abs(dev_thr - float(thr)) <= 2e-4 * float(thr), (dev_thr, float(thr))
                assert got[5] == int((not stop) and snew < max_infer - 1 - 6)


@pytest.mark.parametrize("dtype,code", [(torch.float16, _lib.DT_F16), (torch.bfloat16, _lib.DT_BF16)], ids=["f16", "bf16"])
def test_draft_step_control(dtype, code):
    lib = _lib.lib()
    scr = _scratch(lib)
    g = torch.Generator().manual_seed(5)
    steps = 6
    for trial in range(12):
        scale = [0.5, 2.0, 6.0, 12.0][trial % 4]
        lg = (torch.randn(1, V, generator=g) * scale).to(dtype)
        f = lg.float()
        tok = int(f.argmax())
        score = float(torch.softmax(f, -1).max())
        step = trial % steps
        s = 40 + trial
        e = s + 1 + step
        for thr in (0.0, 0.6, score - 1e-4, score + 1e-4, 2.0):
            for alen in ((e - 1, e - 2) if step == 0 else (e - 1,)):
                out = (C.c_int32 * 8)()
                lgd = lg.to(DEV)
                _lib.check(lib.kg_draft_step(code, C.c_void_p(lgd.data_ptr()), V, step, steps, float(thr), s, e, alen, C.cast(out, I32P),
                                             C.c_void_p(scr.data_ptr()), _stream()), "kg_draft_step")
                patch = step == 0 and alen < e - 1                                      # :55
                assert out[0] == (2 if patch else 1) and out[1] == (s - 1 if patch else e - 1)   # :57 / :60
                assert out[2] == tok                                                    # :80
                got_score = struct.unpack("f", struct.pack("i", out[3]))[0]
                assert abs(got_score - score) <= 1e-5 * max(1.0, score)                 # :83
                assert out[4] == e + 1                                                  # :82
                want_on = not (step + 1 == steps or got_score < thr)                   # :74 as the NEXT step evaluates it
                assert out[5] == int(want_on), (trial, thr, got_score)
                assert out[6] == e and out[7] == e + 1 - s                              # adapter cache covers [0, e); verify rows s..e
    # adapter-row decision alone
    out = (C.c_int32 * 8)()
    _lib.check(lib.kg_draft_step(code, None, V, 0, steps, 0.6, 50, 51, 49, C.cast(out, I32P), C.c_void_p(scr.data_ptr()), _stream()), "kg_draft_step")
    assert (out[0], out[1], out[2]) == (2, 49, -1)


def test_control_entry_points_refuse_bad_arguments():
    lib = _lib.lib()
    scr = _scratch(lib)
    lg = rows(2, 1).to(DEV)
    out = (C.c_int32 * 10)()
    dr = (C.c_int32 * 1)(3)
    def call(q=2, s=5, rule=0, k=1, p=0.0, t=1.0):
        return lib.kg_accept(0, C.c_void_p(lg.data_ptr()), V, q, C.cast(dr, I32P), s, -1, rule, k, p, t, 0.3, 0.09, 6, 2048, 5, C.cast(out, I32P), C.c_void_p(scr.data_ptr()), _stream())
    assert call() == 0
    assert call(q=16) == -1 and call(q=0) == -1 and call(s=-1) == -1 and call(s=5000) == -1
    assert call(rule=4) == -1 and call(rule=3, t=0.0) == -1 and call(rule=1, k=1) == -1 and call(rule=2, p=0.0) == -1 and call(rule=2, p=0.5, t=0.0) == -1
    assert lib.kg_draft_step(0, C.c_void_p(lg.data_ptr()), V, 6, 6, 0.6, 5, 6, 5, C.cast(out, I32P), C.c_void_p(scr.data_ptr()), _stream()) == -1
    assert lib.kg_draft_step(0, C.c_void_p(lg.data_ptr()), V, 0, 6, 0.6, 0, 1, 0, C.cast(out, I32P), C.c_void_p(scr.data_ptr()), _stream()) == -1
