"""Divergence tracer for parity runs at sizes where no golden file exists.

TEST INFRASTRUCTURE ONLY (tests/, the smoke check and bench.py's cpu_baseline leg): compares one device run
of the Kangaroo loop with the CPU oracle's run on the same weights and prompt, and, where they part, decides
whether the parting is a 16-bit near-tie:

* ids differ at position p: the oracle is teacher-forced on the device's own prefix [0, p)
  (evaluation/inference_kangaroo.py:30 on that prefix); the device's token must be within `tol` x max-abs of
  the oracle's best logit of that row, and, when the device's logits row for the same prefix is supplied, the
  two rows must agree to `tol` x max-abs -- or, when `rounding_model` is given (the same oracle with every stored
  tensor rounded to the device's 16-bit type, i.e. the reference's own arithmetic in that type), to 1.5 x what THAT
  row differs from the fp32 row by, if larger: in bf16 a 32-layer model's logits sit 1.5 .. 1.9e-2 of max-abs from
  their fp32 values whoever computes them.
* ids agree but accept lengths / draft lengths differ: in the first differing round a draft confidence
  (evaluation/inference_kangaroo.py:83) must sit next to the threshold of :74, within what the rounding
  noise of stored 16-bit logits (tol / 5 x max-abs) can move a softmax maximum, or a draft argmax must be a
  top-2 near-tie.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch

TOL = 1e-2          # max-abs relative logit tolerance of the parity bar


def p_tol(p: float, scale: float, tol: float = TOL) -> float:
    """|d p_max| under the rounding noise of stored 16-bit logits, taken as a fifth of the parity tolerance
    times the row's max-abs (2e-3 x max-abs for fp16); d p / d l = p (1 - p) for the top logit and the
    opposite sign for the rest, hence the factor 2."""
    return 2.0 * p * (1.0 - p) * (0.2 * tol) * scale


def first_difference(dev_ids: Sequence[int], dev_accept: Sequence[int], dev_draft: Sequence[int],
                     ref_ids: Sequence[int], ref_accept: Sequence[int], ref_draft: Sequence[int], T0: int,
                     oracle_model, ref_pmax: Sequence[float], ref_draft_logits: Sequence[torch.Tensor], threshold: float,
                     dev_logits_row: Optional[Callable[[List[int]], torch.Tensor]] = None, tol: float = TOL,
                     rounding_model=None) -> Dict:
    """Returns {"kind": "none"} when ids, accept lengths and draft lengths all agree, else a record with
    "kind" ("ids" | "round"), "position", the measured gaps and "within_tol"."""
    dev_ids, ref_ids = list(dev_ids), list(ref_ids)
    n = min(len(dev_ids), len(ref_ids))
    first = next((i for i in range(n) if dev_ids[i] != ref_ids[i]), None)
    if first is None and list(dev_accept) == list(ref_accept) and list(dev_draft) == list(ref_draft):
        return {"kind": "none", "tokens_compared": n - T0}
    if first is not None:
        row = oracle_model.base_model(torch.tensor([dev_ids[:first]])).logits[0, -1].float()
        top2 = row.topk(2).values
        scale = float(row.abs().max())
        gap_dev = float(top2[0] - row[dev_ids[first]])
        rec = {"kind": "ids", "position": first, "new_tokens_before": first - T0, "device_token": dev_ids[first], "oracle_token": ref_ids[first],
               "oracle_argmax_on_device_prefix": int(row.argmax()), "oracle_top2_gap": float(top2[0] - top2[1]),
               "oracle_best_minus_device_token": gap_dev, "row_max_abs": scale, "tolerance": tol * scale}
        ok = gap_dev <= tol * scale
        if dev_logits_row is not None:
            drow = dev_logits_row(dev_ids[:first]).float().cpu().reshape(-1)
            rec["logits_max_abs_diff"] = float((drow - row).abs().max())
            rec["logits_max_abs_diff_rel"] = rec["logits_max_abs_diff"] / scale
            bound = tol * scale
            if rounding_model is not None:
                nrow = rounding_model(torch.tensor([dev_ids[:first]])).logits[0, -1].float()
                rec["reference_arithmetic_in_16_bit_vs_fp32_rel"] = float((nrow - row).abs().max()) / scale
                bound = max(bound, 1.5 * rec["reference_arithmetic_in_16_bit_vs_fp32_rel"] * scale)
            rec["logits_bound_rel"] = bound / scale
            ok = ok and rec["logits_max_abs_diff"] <= bound
        rec["within_tol"] = bool(ok)
        return rec
    da, ra = list(dev_accept[1:]), list(ref_accept[1:])
    i0 = 0
    for r in range(min(len(da), len(ra))):
        if (da[r], dev_draft[r]) == (ra[r], ref_draft[r]):
            i0 += int(ref_draft[r])
            continue
        rec = {"kind": "round", "round": r, "device": [da[r], int(dev_draft[r])], "oracle": [ra[r], int(ref_draft[r])], "steps": []}
        ok = False
        for j in range(int(ref_draft[r])):
            p = float(ref_pmax[i0 + j])
            row = torch.as_tensor(ref_draft_logits[i0 + j]).float()
            scale = float(row.abs().max())
            top2 = row.topk(2).values
            step = {"p_max": p, "threshold": threshold, "p_tolerance": p_tol(p, scale, tol), "draft_top2_gap": float(top2[0] - top2[1]), "row_max_abs": scale}
            step["near_tie"] = bool(abs(p - threshold) <= step["p_tolerance"] or step["draft_top2_gap"] <= tol * scale)
            ok = ok or step["near_tie"]
            rec["steps"].append(step)
        rec["within_tol"] = ok
        return rec
    return {"kind": "round", "within_tol": False, "detail": "lists differ only in length"}
