"""Question-file evaluation loop with the reference's file formats.

Counterpart of evaluation/eval.py:20-247 for a tokenizer-driven run: questions come from a
Spec-Bench / MT-Bench style JSONL (`question_id`, `category`, `turns`), every turn is wrapped in
the Vicuna conversation format, run through `forward_func(inputs, model, tokenizer,
max_new_tokens, **kwargs)` (the 4-tuple interface of evaluation/inference_kangaroo.py:15-129)
between two device synchronisations, decoded, and one JSON line per question is appended to the
answer file with the fields the reference writes (eval.py:221-246): `question_id`, `category`,
`answer_id`, `model_id`, `choices` [{`index`, `turns`, `idxs`, `new_tokens`, `wall_time`,
`accept_lengths`}], `tstamp`.  evaluation/speed.py reads exactly these.

What the reference takes from fastchat / shortuuid / ray is small and restated here: the
question loader, the Vicuna v1.1 prompt format, an answer id, and the split of the question list
over worker processes (eval.py:50-65) as a contiguous shard per rank.

The tokenizer is anything with the Hugging Face call surface used by the reference:
`tokenizer([prompt], return_tensors="pt").input_ids`, `decode(ids, spaces_between_special_tokens=False)`,
`special_tokens_map`, `eos_token_id`.
"""
from __future__ import annotations

import json
import os
import time
import uuid
from typing import Callable, Dict, List, Optional, Sequence

import torch

from .harness import shard_prompts

VICUNA_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. "
                 "The assistant gives helpful, detailed, and polite answers to the user's questions.")


class VicunaConversation:
    """Vicuna v1.1 turn format: system text, then `USER: ...` / `ASSISTANT: ...` turns; a user turn ends
    with a space, an assistant turn with `</s>`; the open assistant turn is a bare `ASSISTANT:`."""
    roles = ("USER", "ASSISTANT")
    name = "vicuna_v1.1"
    stop_str = None
    stop_token_ids = None

    def __init__(self, system: str = VICUNA_SYSTEM):
        self.system = system
        self.messages: List[List[Optional[str]]] = []

    def append_message(self, role: str, message: Optional[str]) -> None:
        self.messages.append([role, message])

    def get_prompt(self) -> str:
        seps = (" ", "</s>")
        out = self.system + seps[0]
        for i, (role, message) in enumerate(self.messages):
            out += f"{role}: {message}{seps[i % 2]}" if message else f"{role}:"
        return out


def load_questions(question_file: str, begin: Optional[int] = None, end: Optional[int] = None) -> List[Dict]:
    """One JSON object per line; the slice [begin:end] as the reference's --question-begin / --question-end."""
    with open(question_file) as f:
        questions = [json.loads(line) for line in f if line.strip()]
    return questions[begin:end]


def _clean(output_ids: Sequence[int], conv: VicunaConversation, tokenizer) -> str:
    """ids -> answer text the way eval.py:187-212 trims it: cut at a stop token id, decode, cut at the
    stop string, drop special tokens, strip."""
    ids = list(output_ids)
    if conv.stop_token_ids:
        hit = [i for i, t in enumerate(ids) if t in conv.stop_token_ids]
        if hit:
            ids = ids[:hit[0]]
    text = tokenizer.decode(ids, spaces_between_special_tokens=False)
    if conv.stop_str and text.find(conv.stop_str) > 0:
        text = text[:text.find(conv.stop_str)]
    for special in getattr(tokenizer, "special_tokens_map", {}).values():
        for tok in (special if isinstance(special, list) else [special]):
            text = text.replace(tok, "")
    return text.strip()


def _answer_turns(model, tokenizer, forward_func: Callable, question: Dict, max_new_tokens: int, device, **kwargs):
    conv = VicunaConversation()
    turns, idxs, new_tokens, wall_time, accept = [], [], [], [], []
    for qs in question["turns"]:
        conv.append_message(conv.roles[0], qs)
        conv.append_message(conv.roles[1], None)
        inputs = tokenizer([conv.get_prompt()], return_tensors="pt")
        inputs = inputs.to(device) if hasattr(inputs, "to") else inputs
        n_in = inputs.input_ids.shape[1]
        torch.cuda.synchronize()
        t0 = time.time()
        output_ids, new_token, idx, accept_lengths = forward_func(inputs, model, tokenizer, max_new_tokens, **kwargs)[:4]
        torch.cuda.synchronize()
        dt = time.time() - t0
        text = _clean(output_ids[0][n_in:], conv, tokenizer)
        turns.append(text)
        idxs.append(int(idx))
        new_tokens.append(int(new_token))
        wall_time.append(dt)
        accept.extend(int(a) for a in accept_lengths)
        conv.messages[-1][-1] = text
    return turns, idxs, new_tokens, wall_time, accept


@torch.inference_mode()
def get_model_answers(model, tokenizer, forward_func: Callable, model_id: str, questions: Sequence[Dict], answer_file: str,
                      max_new_tokens: int, num_choices: int = 1, device="cuda", **kwargs) -> float:
    """eval.py:70-247: one warm-up question, then every question `num_choices` times; appends to
    `answer_file`; returns the mean accept length over everything generated."""
    if not questions:
        return float("nan")
    _answer_turns(model, tokenizer, forward_func, questions[0], max_new_tokens, device, **kwargs)      # warm-up, not recorded
    all_accept: List[int] = []
    os.makedirs(os.path.dirname(os.path.abspath(answer_file)), exist_ok=True)
    for question in questions:
        choices = []
        for i in range(num_choices):
            turns, idxs, new_tokens, wall_time, accept = _answer_turns(model, tokenizer, forward_func, question, max_new_tokens, device, **kwargs)
            all_accept.extend(accept)
            choices.append({"index": i, "turns": turns, "idxs": idxs, "new_tokens": new_tokens, "wall_time": wall_time,
                            "accept_lengths": accept})
        with open(os.path.expanduser(answer_file), "a") as fout:
            fout.write(json.dumps({"question_id": question["question_id"], "category": question.get("category"),
                                   "answer_id": uuid.uuid4().hex[:22], "model_id": model_id, "choices": choices,
                                   "tstamp": time.time()}) + "\n")
    return sum(all_accept) / max(1, len(all_accept))


def reorg_answer_file(answer_file: str) -> None:
    """Sort the answer file by question id and drop duplicates, keeping the last (eval.py:250-262)."""
    answers = {}
    with open(answer_file) as f:
        for line in f:
            if line.strip():
                answers[json.loads(line)["question_id"]] = line if line.endswith("\n") else line + "\n"
    with open(answer_file, "w") as f:
        for qid in sorted(answers):
            f.write(answers[qid])


def run_eval(model, tokenizer, forward_func: Callable, model_id: str, question_file: str, question_begin: Optional[int],
             question_end: Optional[int], answer_file: str, max_new_tokens: int, num_choices: int = 1,
             rank: int = 0, world_size: int = 1, **kwargs) -> float:
    """eval.py:20-68.  With world_size > 1 every process (one per GPU) calls this with its rank and gets a
    contiguous share of the questions, as the reference's worker fan-out cuts them, and all append to the same
    answer file.  With an initialised torch.distributed process group the ranks meet at a barrier and rank 0
    then calls reorg_answer_file (the reference does it after ray.get, eval.py:66-68); processes started
    without a process group leave that call to whoever waits for them."""
    questions = load_questions(question_file, question_begin, question_end)
    mine = shard_prompts(questions, rank, world_size)
    mean_accept = get_model_answers(model, tokenizer, forward_func, model_id, mine, answer_file, max_new_tokens, num_choices, **kwargs)
    if world_size == 1:
        reorg_answer_file(answer_file)
    else:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.barrier()
            if rank == 0:
                reorg_answer_file(answer_file)
            dist.barrier()
    return mean_accept


MT_BENCH_CATEGORIES = ("writing", "roleplay", "reasoning", "math", "coding", "extraction", "stem", "humanities")


def _read_task(jsonl_file: str, task: str) -> List[Dict]:
    rows = []
    with open(jsonl_file, encoding="utf-8") as f:
        for line in f:
            if not line.strip():
                continue
            obj = json.loads(line)
            if task == "overall" or (task == "mt_bench" and obj.get("category") in MT_BENCH_CATEGORIES) or obj.get("category") == task:
                rows.append(obj)
    return rows


def speed(jsonl_file: str, jsonl_file_base: str, tokenizer=None, task: str = "overall") -> Dict[str, float]:
    """The report of evaluation/speed.py:7-68 from two answer files: tokens/s = mean over questions of
    sum(new_tokens) / sum(wall_time) of choice 0; the baseline's tokens are counted by re-tokenising its
    answers (len(ids) - 1 per turn, :49-52) when a tokenizer is given, else taken from its `new_tokens`;
    speed-up = ratio of the two means; mean accept length over all of the first file's accept lengths."""
    speeds, accept = [], []
    for row in _read_task(jsonl_file, task):
        c = row["choices"][0]
        speeds.append(sum(c["new_tokens"]) / sum(c["wall_time"]))
        accept.extend(c["accept_lengths"])
    speeds0 = []
    for row in _read_task(jsonl_file_base, task):
        c = row["choices"][0]
        tokens = sum(len(tokenizer(t).input_ids) - 1 for t in c["turns"]) if tokenizer is not None else sum(c["new_tokens"])
        speeds0.append(tokens / sum(c["wall_time"]))
    tps = sum(speeds) / max(1, len(speeds))
    tps0 = sum(speeds0) / max(1, len(speeds0))
    return {"tokens_per_second": tps, "tokens_per_second_baseline": tps0, "speedup": tps / tps0 if tps0 else float("nan"),
            "mean_accept": sum(accept) / max(1, len(accept)), "questions": float(len(speeds))}


def main(argv: Optional[Sequence[str]] = None) -> None:
    """Command line of evaluation/inference_kangaroo.py:131-245 (same flags; `--question-file` and
    `--baseline` added): three runs over the question file, answers to data/<bench>/<model-id>/<run>.jsonl."""
    import argparse
    import types
    ap = argparse.ArgumentParser(description="Kangaroo question-file evaluation on MI355X")
    ap.add_argument("--model-path", required=True)
    ap.add_argument("--adapter-path", required=True)
    ap.add_argument("--model-id", required=True)
    ap.add_argument("--bench-name", default="mt_bench")
    ap.add_argument("--question-file", default="data/question.jsonl")
    ap.add_argument("--question-begin", type=int)
    ap.add_argument("--question-end", type=int)
    ap.add_argument("--max-new-tokens", type=int, default=1024)
    ap.add_argument("--num-choices", type=int, default=1)
    ap.add_argument("--threshold", type=float, default=0.4)
    ap.add_argument("--exitlayer", type=int, default=2)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--dtype", default="float16", choices=["float16", "bfloat16"])
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("--baseline", action="store_true", help="plain greedy decoding on the same kernels instead of Kangaroo")
    a = ap.parse_args(argv)
    from transformers import AutoTokenizer
    from .inference import kangaroo_forward, vanilla_forward
    from .kangaroo_model import KangarooModel
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    model = KangarooModel(a.model_path, a.adapter_path, types.SimpleNamespace(dtype=a.dtype), EARLY_STOP_LAYER=a.exitlayer)
    tokenizer = AutoTokenizer.from_pretrained(a.model_path)
    if a.baseline:
        fwd, kw = (lambda inputs, m, t, n, **k: vanilla_forward(inputs, m, t, n)), {}
    else:
        fwd, kw = kangaroo_forward, dict(threshold=a.threshold, SPECULATIVE_DECODING_STEPS=a.steps, EARLY_STOP_LAYER=a.exitlayer)
    for run in range(a.runs):
        answer_file = f"data/{a.bench_name}/{a.model_id}/{run}.jsonl"
        print(f"Output to {answer_file}")
        mean_accept = run_eval(model, tokenizer, fwd, a.model_id, a.question_file, a.question_begin, a.question_end, answer_file,
                               a.max_new_tokens, a.num_choices, rank=rank, world_size=world, **kw)
        print("#Mean accepted tokens: ", mean_accept)


if __name__ == "__main__":
    main()
