"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares,
host logic (packing, shapes, checkpoint formats, prompt sharding, harness metrics), and the
product path fails loudly without a GPU instead of falling back to anything."""
import json
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "kangaroo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kg_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from kangaroo_amd import _lib
    h = _lib.lib()                      # raises if the .so is missing or a bound symbol is absent
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(h, n), f"{n} declared in include/kangaroo_hip.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes signature"
    assert set(_lib.SYMBOLS) == set(names)
    assert h.kg_version() >= 1


def test_workspace_query_needs_no_gpu():
    import ctypes as C
    from kangaroo_amd import _lib
    from kangaroo_amd.config import get_shape
    s = get_shape("vicuna-7b")
    ks = _lib.KgShape(0, s.hidden_size, s.num_hidden_layers, s.num_attention_heads, s.num_key_value_heads, s.head_dim,
                      s.intermediate_size, s.vocab_size, s.rms_norm_eps, 2048, 2)
    n = _lib.lib().kg_workspace_bytes(C.byref(ks), 2048)
    assert 16 * 2**20 < n < 2**30


def test_no_cpu_fallback():
    """Without a GPU the engine refuses to construct; it never routes to the oracle."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from kangaroo_amd._lib import KangarooHipError
    from kangaroo_amd.config import get_shape
    from kangaroo_amd.kangaroo_model import KangarooModel
    from kangaroo_amd.weights import synth_state_dicts
    shape = get_shape("tiny")
    base, adapter = synth_state_dicts(shape, 2)
    with pytest.raises(KangarooHipError):
        KangarooModel.from_state_dicts(shape, base, adapter)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "kangaroo_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.replace("the oracle", ""), f"{f} mentions oracle"


def test_packing_roundtrip_and_orders():
    from kangaroo_amd.packing import pack_weight, unpack_weight, pack_qkv, pack_gate_up, rope_pair_order, rope_tables
    w = torch.randn(100, 96).half()
    p = pack_weight(w)
    assert p.numel() == 112 * 96
    assert torch.equal(unpack_weight(p, 100, 96), w)
    rg, kt, g, r, j = 3, 2, 1, 7, 5
    assert p[(rg * 3 + kt) * 512 + g * 128 + r * 8 + j] == w[16 * rg + r, 32 * kt + 8 * g + j]
    with pytest.raises(ValueError):
        pack_weight(torch.zeros(16, 48))
    o = rope_pair_order(2, 8).tolist()
    assert o[:4] == [0, 4, 1, 5] and o[8:12] == [8, 12, 9, 13]
    H, Hkv, hd, d = 4, 2, 16, 64
    wq, wk, wv = torch.randn(H * hd, d), torch.randn(Hkv * hd, d), torch.randn(Hkv * hd, d)
    u = unpack_weight(pack_qkv(wq, wk, wv, H, Hkv, hd), (H + 2 * Hkv) * hd, d)
    assert torch.equal(u[0], wq[0]) and torch.equal(u[1], wq[8]) and torch.equal(u[H * hd + 1], wk[8])
    wg, wu = torch.randn(40, 32), torch.randn(40, 32)
    u = unpack_weight(pack_gate_up(wg, wu), 80, 32)
    assert torch.equal(u[0::2], wg) and torch.equal(u[1::2], wu)
    cos, sin = rope_tables(16, 64)
    from oracle.kangaroo_oracle import rope_tables as ref_tables
    rc, rs = ref_tables(16, 64)
    assert torch.equal(cos, rc[:, :8]) and torch.equal(sin, rs[:, :8]) and torch.equal(rc[:, :8], rc[:, 8:])


def test_checkpoint_formats(tmp_path):
    """HF sharded checkpoint + adapter directory (kangaroo/kangaroo_model.py:22-44)."""
    from kangaroo_amd.config import get_shape, LlamaShape
    from kangaroo_amd.weights import synth_state_dicts, save_synthetic_checkpoint, load_base_state_dict, load_adapter_state_dict
    shape = get_shape("tiny")
    base, adapter = synth_state_dicts(shape, 2, seed=5)
    save_synthetic_checkpoint(str(tmp_path / "base"), str(tmp_path / "adapter"), shape, base, adapter, n_shards=3)
    b2 = load_base_state_dict(str(tmp_path / "base"))
    a2 = load_adapter_state_dict(str(tmp_path / "adapter"))
    assert set(b2) == set(base) and all(torch.equal(b2[k], base[k]) for k in base)
    assert all(torch.equal(a2[k], adapter[k]) for k in adapter)
    s2 = LlamaShape.from_hf_config(str(tmp_path / "base"))
    assert s2.hidden_size == shape.hidden_size and s2.num_key_value_heads == shape.num_key_value_heads
    assert LlamaShape.from_hf_config(str(tmp_path / "adapter" / "config.json")).num_hidden_layers == 1


def test_shapes_and_byte_accounting():
    from kangaroo_amd.config import get_shape
    s = get_shape("vicuna-7b")
    assert s.layer_weight_bytes() == 404_750_336 and s.head_weight_bytes() == 262_144_000
    assert s.attn_only_weight_bytes() == 134_217_728 and s.kv_bytes_per_position() == 16384
    assert get_shape("vicuna-13b").layer_weight_bytes() == (4 * 5120 * 5120 + 3 * 5120 * 13824) * 2


def test_prompts_and_sharding():
    from kangaroo_amd.harness import spec_bench_prompts, shard_prompts, SPEC_BENCH_GROUPS
    ps = spec_bench_prompts(32000)
    assert len(ps) == 560 == sum(c for _, c, _ in SPEC_BENCH_GROUPS)
    assert {p.category for p in ps[:6]} == {g[0] for g in SPEC_BENCH_GROUPS}
    assert all(3 <= int(p.input_ids.min()) and int(p.input_ids.max()) < 32000 for p in ps[:20])
    again = spec_bench_prompts(32000, 10)
    assert all(torch.equal(a.input_ids, b.input_ids) for a, b in zip(again, ps))
    for world in (1, 2, 4, 8, 7):
        shards = [shard_prompts(ps, r, world) for r in range(world)]
        flat = [p.question_id for sh in shards for p in sh]
        assert flat == [p.question_id for p in ps]          # disjoint, ordered, complete
        assert max(len(s) for s in shards) - min(len(s) for s in shards) < world


def test_speed_report_formulas():
    """evaluation/speed.py:24-30,58-60."""
    from kangaroo_amd.harness import EvalRecord, speed_report
    recs = [EvalRecord(0, "qa", 100, 0.5, 40, [1, 3, 2]), EvalRecord(1, "qa", 50, 0.5, 30, [1, 1])]
    base = [EvalRecord(0, "qa", 100, 1.0, 100, []), EvalRecord(1, "qa", 50, 1.0, 50, [])]
    r = speed_report(recs, base)
    assert r["tokens_per_second"] == pytest.approx(150.0) and r["tokens_per_second_baseline"] == pytest.approx(75.0)
    assert r["speedup"] == pytest.approx(2.0) and r["mean_accept"] == pytest.approx(8 / 5)


_WORKER = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from kangaroo_amd.harness import spec_bench_prompts, shard_prompts
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
ps = spec_bench_prompts(1000, 11)
mine = shard_prompts(ps, rank, world)
# stand-in for decoding: new tokens = prompt length (no GPU here); aggregate as bench.py does
t = torch.tensor([float(sum(p.input_ids.shape[1] for p in mine)), 1.0 + rank], dtype=torch.float64)
tmax = t.clone()
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
ids = [None] * world
dist.all_gather_object(ids, [p.question_id for p in mine])
if rank == 0:
    print(json.dumps({"tokens": float(t[0]), "tmax": float(tmax[1]), "ids": ids}))
dist.barrier()
dist.destroy_process_group()
"""


def test_prompt_sharding_two_ranks_gloo(tmp_path):
    """World size 2 on CPU: disjoint shards, SUM of tokens / MAX of time as bench.py aggregates."""
    import json
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    from kangaroo_amd.harness import spec_bench_prompts
    ps = spec_bench_prompts(1000, 11)
    assert j["tokens"] == float(sum(p.input_ids.shape[1] for p in ps))
    assert j["tmax"] == 2.0
    assert j["ids"][0] + j["ids"][1] == [p.question_id for p in ps] and len(j["ids"][0]) == 5


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (WORLD_SIZE unset) starts two rank processes itself, as the
    reference's harness spawns its workers (evaluation/eval.py:38-68); --launch-check stops short of the decode, so
    this runs without a GPU: process group, SUM / MAX aggregation, rank 0's line relayed by the parent."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["KG_BENCH_BACKEND"] = "gloo"
    for extra, tp in ((["--gpus", "2"], 1), (["--tp", "2"], 2)):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check"] + extra,
                             capture_output=True, text=True, env=env, timeout=240)
        assert out.returncode == 0, out.stderr[-2000:]
        j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert j["n_gpus"] == 2 and j["tp"] == tp and j["ranks"] == [[0, 0], [1, 1]]
        assert j["tokens"] == 201.0 and j["t_job"] == 2.0          # SUM over ranks, MAX over ranks
    # a rank count that does not fill whole tensor-parallel groups is refused before anything starts
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch-check", "--gpus", "3", "--tp", "2"],
                         capture_output=True, text=True, env=env, timeout=60)
    assert out.returncode != 0 and "multiple" in out.stderr


def test_question_file_conversation_and_answer_file(tmp_path):
    """Host-side pieces of the question-file evaluation loop (no GPU): Vicuna turn format, question
    slicing, answer-file re-ordering."""
    from kangaroo_amd.eval import VicunaConversation, load_questions, reorg_answer_file, VICUNA_SYSTEM
    conv = VicunaConversation()
    conv.append_message(conv.roles[0], "Hi")
    conv.append_message(conv.roles[1], None)
    assert conv.get_prompt() == VICUNA_SYSTEM + " USER: Hi ASSISTANT:"
    conv.messages[-1][-1] = "Hello!"
    conv.append_message(conv.roles[0], "Again")
    conv.append_message(conv.roles[1], None)
    assert conv.get_prompt() == VICUNA_SYSTEM + " USER: Hi ASSISTANT: Hello!</s>USER: Again ASSISTANT:"
    qf = tmp_path / "question.jsonl"
    qf.write_text("".join(json.dumps({"question_id": i, "category": "qa", "turns": [f"q{i}"]}) + "\n" for i in range(5)))
    assert [q["question_id"] for q in load_questions(str(qf), 1, 4)] == [1, 2, 3]
    assert len(load_questions(str(qf))) == 5
    af = tmp_path / "answers.jsonl"
    af.write_text("".join(json.dumps({"question_id": i, "v": v}) + "\n" for i, v in [(3, 0), (1, 0), (3, 1)]))
    reorg_answer_file(str(af))
    rows = [json.loads(l) for l in af.read_text().splitlines()]
    assert [(r["question_id"], r["v"]) for r in rows] == [(1, 0), (3, 1)]


def test_speed_report_from_answer_files(tmp_path):
    from kangaroo_amd.eval import speed
    def row(qid, cat, nt, wt, acc):
        return json.dumps({"question_id": qid, "category": cat, "choices": [{"index": 0, "turns": ["a b c"], "idxs": [1], "new_tokens": nt, "wall_time": wt, "accept_lengths": acc}]}) + "\n"
    k = tmp_path / "k.jsonl"; b = tmp_path / "b.jsonl"
    k.write_text(row(1, "qa", [100, 50], [0.5, 0.25], [1, 2, 3]) + row(2, "math", [80], [0.2], [2, 2]))
    b.write_text(row(1, "qa", [100, 50], [1.0, 0.5], [1]) + row(2, "math", [80], [0.8], [1]))
    r = speed(str(k), str(b))
    assert abs(r["tokens_per_second"] - (200 + 400) / 2) < 1e-9 and abs(r["tokens_per_second_baseline"] - (100 + 100) / 2) < 1e-9
    assert abs(r["speedup"] - 3.0) < 1e-9 and abs(r["mean_accept"] - 2.0) < 1e-9
    assert speed(str(k), str(b), task="qa")["questions"] == 1


def test_constants_of_the_header_match_the_bindings():
    """Values the Python side passes by number are the header's."""
    src = open(os.path.join(ROOT, "include", "kangaroo_hip.h")).read()
    from kangaroo_amd import tp
    m = re.search(r"#define\s+KG_POS_IN_SCRATCH\s+\((-?\d+)\)", src)
    assert m and int(m.group(1)) == tp.POS_IN_SCRATCH
