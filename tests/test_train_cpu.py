"""Host logic of the adapter training path (no GPU): sample preparation and collation as train.py:101-185 does them,
the warm-up schedule, the data-parallel gradient mean on two gloo ranks, and the absence of a CPU path."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from kangaroo_amd import train as T


def _sample(n, d, g, exit_layer=2):
    return {"hidden_state": torch.randn(n, d, generator=g), "input_ids": torch.randint(3, 90, (n,), generator=g),
            "loss_mask": (torch.rand(n, generator=g) > 0.4).long(), f"hidden_state_layer{exit_layer}": torch.randn(n, d, generator=g)}


def test_prepare_sample_shifts_and_clears_the_last_position():
    g = torch.Generator().manual_seed(1)
    raw = _sample(9, 8, g)
    s = T.prepare_sample(raw, 2, max_len=7)
    assert s["hidden_state_big"].shape == (1, 7, 8) and s["target"].shape == (1, 7, 8)
    assert torch.equal(s["target"][0, :6], raw["hidden_state"][1:7]) and float(s["target"][0, 6].abs().sum()) == 0.0
    assert torch.equal(s["hidden_state_early"][0, :6], raw["hidden_state_layer2"][1:7]) and float(s["hidden_state_early"][0, 6].abs().sum()) == 0.0
    assert torch.equal(s["input_ids"][0, :6], raw["input_ids"][1:7]) and int(s["input_ids"][0, 6]) == 0
    assert s["loss_mask"][-1] == 0 and s["loss_mask"][:-1] == raw["loss_mask"][:6].tolist()
    assert s["attention_mask"] == [1] * 7


def test_collator_right_pads_with_zeros():
    g = torch.Generator().manual_seed(2)
    feats = [T.prepare_sample(_sample(n, 8, g), 2) for n in (5, 9, 3)]
    b = T.DataCollatorWithPadding()(feats)
    assert b["hidden_states_early"].shape == (3, 9, 8) and b["target"].shape == (3, 9, 8) and b["input_ids"].shape == (3, 9)
    assert b["attention_mask"].tolist()[0] == [1] * 5 + [0] * 4 and b["attention_mask"].tolist()[2] == [1] * 3 + [0] * 6
    assert b["loss_mask"][0, 4:].sum() == 0 and float(b["hidden_states_early"][2, 3:].abs().sum()) == 0.0
    assert torch.equal(b["hidden_states"][1], feats[1]["hidden_state_big"][0])


def test_warmup_schedule():
    assert T.linear_warmup_lr(5e-5, 0, 2000, 800000) == 0.0
    assert abs(T.linear_warmup_lr(5e-5, 1000, 2000, 800000) - 2.5e-5) < 1e-15
    assert T.linear_warmup_lr(5e-5, 2000, 2000, 800000) == 5e-5
    assert abs(T.linear_warmup_lr(5e-5, 401000, 2000, 800000) - 2.5e-5) < 1e-12
    assert T.linear_warmup_lr(5e-5, 900000, 2000, 800000) == 0.0


def test_every_rank_runs_the_same_number_of_steps():
    """954 files, 8 ranks, batches of 4: 239 batches are not a multiple of 8.  Each step ends in a gradient all-reduce,
    so a rank with one batch more would wait for its peers until the collective times out.  The order wraps round
    instead (accelerate's even_batches), every file is still seen, and one rank keeps the reference's plain batching
    with a short last batch (DataLoader(drop_last=False), train.py:217)."""
    for n, bs, world in ((954, 4, 8), (10, 4, 2), (7, 2, 4), (3, 4, 2), (16, 4, 4)):
        per_rank = [T.epoch_batches(n, bs, 5, r, world) for r in range(world)]
        assert len({len(b) for b in per_rank}) == 1, (n, bs, world)
        assert all(len(x) == bs for b in per_rank for x in b)
        seen = [i for b in per_rank for x in b for i in x]
        assert set(seen) == set(range(n))
        assert len(seen) - n < world * bs + bs                 # wrapped indices: less than one global step plus the short batch's fill
        # interleaved back, the ranks' batches are the epoch's order (wrapped at the end)
        order = torch.randperm(n, generator=torch.Generator().manual_seed(5)).tolist()
        flat = [i for step in zip(*per_rank) for x in step for i in x]
        assert flat[:n] == order and all(flat[n + j] == order[j % n] for j in range(len(flat) - n))
    one = T.epoch_batches(10, 4, 5, 0, 1)
    assert [len(x) for x in one] == [4, 4, 2] and sorted(i for x in one for i in x) == list(range(10))
    assert T.epoch_batches(0, 4, 0, 0, 2) == []
    assert T.epoch_batches(10, 4, 5, 0, 2) != T.epoch_batches(10, 4, 6, 0, 2)      # the seed (epoch index) reshuffles


def test_no_cpu_training_path():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from kangaroo_amd.config import LlamaShape
    from kangaroo_amd._lib import KangarooHipError
    shape = LlamaShape(hidden_size=16, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2, intermediate_size=32, vocab_size=11)
    with pytest.raises(KangarooHipError):
        T.AdapterTrainer(shape, {}, torch.zeros(11, 16))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_main(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sync = T.GradSync()
    g = torch.Generator().manual_seed(100 + rank)
    a, b, c = torch.randn(7, 5, generator=g).half(), torch.randn(11, generator=g), torch.randn(3, 4, generator=g).half()
    (a32, b32) = sync.start([a, b])           # two buckets in flight at once
    (c32,) = sync.start([c])
    sync.finish()
    out[rank] = (a32.clone(), b32.clone(), c32.clone())
    dist.destroy_process_group()


def test_gradient_mean_over_two_gloo_ranks():
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_main, args=(world, port, out), nprocs=world, join=True)
    want = []
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        want.append((torch.randn(7, 5, generator=g).half().float(), torch.randn(11, generator=g), torch.randn(3, 4, generator=g).half().float()))
    mean = [sum(w[i] for w in want) / world for i in range(3)]
    for rank in range(world):
        for i in range(3):
            assert torch.allclose(out[rank][i], mean[i], atol=1e-6), (rank, i)
    # every rank holds the same bits, so the replicas' weights stay identical step after step
    for i in range(3):
        assert torch.equal(out[0][i], out[1][i])


def _epoch_rank_main(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=30))
    sync = T.GradSync()
    seen = []
    for batch in T.epoch_batches(11, 2, 3, rank, world):      # 11 files, 2 ranks x 2 per batch: 6 batches, no multiple of 2 x 2 files
        (g32,) = sync.start([torch.full((4,), float(len(seen) + 1 + rank))])       # what a step does: one collective per bucket
        sync.finish()
        seen.append((batch, float(g32[0])))
    out[rank] = seen
    dist.destroy_process_group()


def test_an_epoch_on_two_gloo_ranks_ends_together():
    """Two ranks walk one epoch's batches with a gradient all-reduce per step, as train_epoch does: with unequal step counts
    the rank with the extra batch would sit in its collective until the 30 s timeout and the spawn would fail."""
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_epoch_rank_main, args=(world, port, out), nprocs=world, join=True)
    assert len(out[0]) == len(out[1]) == 3
    assert {i for r in range(world) for b, _ in out[r] for i in b} == set(range(11))
    assert [m for _, m in out[0]] == [m for _, m in out[1]] == [1.5, 2.5, 3.5]        # the mean of step + rank over both ranks


def test_head_weight_alone_is_read_from_its_shard(tmp_path):
    # train.py:69-86 opens only the shard that holds lm_head.weight
    import json
    a = {"model.embed_tokens.weight": torch.randn(5, 4)}
    b = {"lm_head.weight": torch.randn(5, 4), "model.norm.weight": torch.ones(4)}
    torch.save(a, tmp_path / "pytorch_model-00001-of-00002.bin")
    torch.save(b, tmp_path / "pytorch_model-00002-of-00002.bin")
    wm = {k: "pytorch_model-00001-of-00002.bin" for k in a}
    wm.update({k: "pytorch_model-00002-of-00002.bin" for k in b})
    (tmp_path / "pytorch_model.bin.index.json").write_text(json.dumps({"weight_map": wm}))
    (tmp_path / "pytorch_model-00001-of-00002.bin").write_bytes(b"not a checkpoint")      # must not be opened
    from kangaroo_amd.weights import load_base_state_dict
    got = load_base_state_dict(str(tmp_path), only=("lm_head.weight",))
    assert list(got) == ["lm_head.weight"] and torch.equal(got["lm_head.weight"], b["lm_head.weight"])
