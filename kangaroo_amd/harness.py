"""Minimal evaluation harness for the decode path (no tokenizer, no fastchat).

Counterpart of the reference's run_eval / get_model_answers (evaluation/eval.py:20-247)
and speed report (evaluation/speed.py:7-68) for token-id prompts:
  * one warm-up prompt first (eval.py:92-160),
  * wall time brackets one forward_func call between two device synchronisations,
    prefill included (eval.py:181-192),
  * tokens/s per prompt = new_tokens / wall_time, reported as the mean over prompts,
    speed-up = ratio of those means, mean accept = mean of all accept lengths
    (speed.py:24-30,58-60,64),
  * one JSON line per prompt with the reference's fields `new_tokens`, `wall_time`,
    `accept_lengths`, `idxs` (eval.py:232-233).
Prompts are synthetic token ids shaped like Spec-Bench's six task groups.
"""
from __future__ import annotations

import json
import time
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence

import torch

# (category, count, (min_len, max_len)) mirroring evaluation/speed.py:73's task list
SPEC_BENCH_GROUPS = [
    ("mt_bench", 160, (60, 350)),
    ("translation", 80, (60, 80)),
    ("summarization", 80, (600, 1100)),
    ("qa", 80, (50, 70)),
    ("math_reasoning", 80, (100, 140)),
    ("rag", 80, (900, 1500)),
]


@dataclass
class Prompt:
    question_id: int
    category: str
    input_ids: torch.Tensor      # [1, T0] int64 (CPU)


def spec_bench_prompts(vocab_size: int, n: Optional[int] = None, seed: int = 0) -> List[Prompt]:
    """Deterministic synthetic prompts: ids uniform in [3, V), lengths uniform in each
    group's range, groups interleaved so any prefix covers every group."""
    per_group = []
    qid = 0
    for cat, count, (lo, hi) in SPEC_BENCH_GROUPS:
        items = []
        for i in range(count):
            g = torch.Generator().manual_seed(seed * 1000003 + qid)
            T0 = int(torch.randint(lo, hi + 1, (1,), generator=g))
            ids = torch.randint(3, vocab_size, (1, T0), generator=g)
            items.append(Prompt(qid, cat, ids))
            qid += 1
        per_group.append(items)
    out: List[Prompt] = []
    i = 0
    while any(per_group):
        grp = per_group[i % len(per_group)]
        if grp:
            out.append(grp.pop(0))
        i += 1
    return out if n is None else out[:n]


def shard_prompts(prompts: Sequence, rank: int, world: int) -> List:
    """Contiguous chunks of len // world questions, as the reference's multi-GPU fan-out cuts them
    (evaluation/eval.py:50-65).  The reference's range(0, len, chunk) yields one more task of len % world
    questions for whichever worker is free; with a fixed process per GPU the last rank takes that remainder."""
    if world <= 1:
        return list(prompts)
    chunk = len(prompts) // world
    if chunk == 0:
        return list(prompts[rank:rank + 1])
    lo = rank * chunk
    hi = len(prompts) if rank == world - 1 else lo + chunk
    return list(prompts[lo:hi])


@dataclass
class EvalRecord:
    question_id: int
    category: str
    new_tokens: int
    wall_time: float
    idxs: int
    accept_lengths: List[int]
    output_ids: List[int] = field(default_factory=list)


def run_eval(model, forward_func: Callable, prompts: Sequence[Prompt], max_new_tokens: int, eos_token_id: int,
             warmup: int = 1, answer_file: Optional[str] = None, keep_ids: bool = False, **kwargs) -> List[EvalRecord]:
    dev = model.engine.device
    for p in prompts[:warmup]:
        forward_func(p.input_ids.to(dev), model, eos_token_id, max_new_tokens, **kwargs)
    recs: List[EvalRecord] = []
    fh = open(answer_file, "a") if answer_file else None
    for p in prompts:
        ids = p.input_ids.to(dev)
        torch.cuda.synchronize()
        t0 = time.time()
        out, new_tokens, idx, accept = forward_func(ids, model, eos_token_id, max_new_tokens, **kwargs)[:4]
        torch.cuda.synchronize()
        dt = time.time() - t0
        r = EvalRecord(p.question_id, p.category, new_tokens, dt, idx, list(accept), out[0] if keep_ids else [])
        recs.append(r)
        if fh:
            fh.write(json.dumps({"question_id": r.question_id, "category": r.category,
                                 "choices": [{"index": 0, "new_tokens": [r.new_tokens], "wall_time": [r.wall_time],
                                              "idxs": [r.idxs], "accept_lengths": r.accept_lengths}]}) + "\n")
    if fh:
        fh.close()
    return recs


def speed_report(recs: Sequence[EvalRecord], base_recs: Optional[Sequence[EvalRecord]] = None) -> Dict[str, float]:
    """tokens/s = mean over prompts of new_tokens / wall_time; speed-up = ratio of means."""
    if not recs:
        return {}
    speeds = [r.new_tokens / r.wall_time for r in recs]
    acc = [a for r in recs for a in r.accept_lengths]
    rep = {"tokens_per_second": sum(speeds) / len(speeds), "mean_accept": sum(acc) / max(1, len(acc)),
           "total_tokens": float(sum(r.new_tokens for r in recs)), "total_time": float(sum(r.wall_time for r in recs))}
    if base_recs:
        b = [r.new_tokens / r.wall_time for r in base_recs]
        rep["tokens_per_second_baseline"] = sum(b) / len(b)
        rep["speedup"] = rep["tokens_per_second"] / rep["tokens_per_second_baseline"]
    return rep
