"""Benchmark of the Kangaroo decode path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one whole kangaroo_forward call (prefill + draft/verify rounds) on one
synthetic Spec-Bench-shaped prompt.  At N=1 the workload is the reference's
headline configuration: Vicuna-7B shape, fp16, exitlayer=2, steps=6, threshold=0.6, greedy,
max_new_tokens=1024, random-init weights with a recipe calibrated so the mean accept
length is near the reference's recorded 2.05.  For N>1 (launched by
torch.distributed.run, one rank per GPU) every rank holds a full replica and decodes
its own K prompts: independent bsz=1 streams, no data-path collective ("weak" scaling).
Run without a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) the script starts the N
ranks itself, one fresh child process per GPU, relays rank 0's line and exits non-zero if any rank
did; this mirrors the reference's harness, which spawns its own workers (evaluation/eval.py:38-68).

Rank 0 prints ONE JSON line: whole-job decode tokens/s (prefill inside the timed
region, as the reference's harness times it), speed-up over plain greedy decoding on
the same kernels, achieved weight-streaming GB/s, the roofline record of the dominant
kernel (timed live with HIP events on its launch stream) and, at N=1, the CPU
baseline: the oracle's restatement of the reference loop on the host cores over a
bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from kangaroo_amd.config import get_shape, DEFAULT_EXIT_LAYER
from kangaroo_amd.harness import spec_bench_prompts
from kangaroo_amd.kangaroo_model import KangarooModel
from kangaroo_amd.weights import Recipe, synth_state_dicts

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak
CAL_RECIPE = (0.08, 10.0)      # (deep_scale, head_scale): mean accept ~2 at the 7B shape


def round_bytes(shape, L, d):
    """Algorithmic weight bytes of one verify round with d drafted tokens (fp16):
    (d+1) shallow passes + d (adapter + head) + verify of the remaining layers + head."""
    lay, ad, hd = shape.layer_weight_bytes(), shape.attn_only_weight_bytes(), shape.head_weight_bytes()
    return (d + 1) * L * lay + d * (ad + hd) + (shape.num_hidden_layers - L) * lay + hd


def round_kv_bytes(shape, L, d, T):
    """Algorithmic KV-cache bytes of the same round at context length T: every pass over a layer
    reads that layer's K and V rows [0, T) once (2 * Hkv * hd * 2 B per position) and appends its own."""
    per_pos = 2 * shape.num_key_value_heads * shape.head_dim * 2
    passes = (d + 1) * L + d + (shape.num_hidden_layers - L)        # shallow layers, adapter, deep layers
    return passes * per_pos * T + ((d + 1) * L + d + (d + 1) * (shape.num_hidden_layers - L)) * per_pos


def cpu_baseline(shape, L, base_sd_gpu, adapter_sd_gpu, steps, thr, model, n_prompts=4, budget_new=32, prompt_len=32):
    """The oracle (CPU restatement of the reference loop, fp32 math on the same 16-bit weights) timed on the
    host cores over a bounded sample: n_prompts prompts of prompt_len tokens, budget_new new tokens each.
    The same prompts then run on the device, and whatever differs first per prompt (a token, an accept or a
    draft length) is traced by oracle/parity.py: the oracle is teacher-forced on the device's prefix and the
    device's token and logits row must sit within 1e-2 of the row's max-abs (a 16-bit argmax near-tie)."""
    from oracle import kangaroo_oracle as ko
    from oracle import parity
    from kangaroo_amd.inference import kangaroo_forward
    t0 = time.time()
    base = {k: v.to("cpu") for k, v in base_sd_gpu.items()}
    adapter = {k: v.to("cpu") for k, v in adapter_sd_gpu.items()}
    om = ko.OracleKangarooModel(ko.Shape(**shape.to_dict()), base, adapter, L)
    del base, adapter
    # the same base model (same fp32 copies of the weights) with every stored tensor rounded to the device's type: what the
    # reference's own arithmetic in that type would give; only used to size the logits-row check of a traced difference
    import copy
    rounding = copy.copy(om.base_model)
    rounding.round_to = torch.bfloat16 if str(model.engine.dtype).endswith("bfloat16") else torch.float16
    rounding.past_key_values = None
    setup = time.time() - t0
    tot_new, tot_dt, accepts, diffs, same_tokens, cmp_tokens = 0, 0.0, [], [], 0, 0
    for i in range(n_prompts):
        ids = torch.randint(3, shape.vocab_size, (1, prompt_len), generator=torch.Generator().manual_seed(7 + i))
        tr = ko.Trace()
        t0 = time.time()
        ref = ko.kangaroo_forward(ids, om, -1, budget_new, EARLY_STOP_LAYER=L, SPECULATIVE_DECODING_STEPS=steps, threshold=thr,
                                  trace=tr, keep_logits=True)
        tot_dt += time.time() - t0
        tot_new += ref[1]
        accepts += ref[3]
        # the same prompt on the device (untimed): token identity with the CPU path, or a traced near-tie
        out, new_tok, idx, accept, st = kangaroo_forward(ids.cuda(), model, -1, budget_new, EARLY_STOP_LAYER=L,
                                                         SPECULATIVE_DECODING_STEPS=steps, threshold=thr, return_stats=True)
        rec = parity.first_difference(out[0], accept, st.draft_lengths, ref[0][0], ref[3], tr.draft_lengths, prompt_len, om,
                                      tr.p_max, tr.draft_logits, thr,
                                      dev_logits_row=lambda pre: model.base_model(torch.tensor([pre]).cuda(), logits_rows="last").logits[0, -1],
                                      rounding_model=rounding)
        rec["prompt_seed"] = 7 + i
        diffs.append(rec)
        n = min(len(out[0]), len(ref[0][0]))
        same_tokens += next((j for j in range(n) if out[0][j] != ref[0][0][j]), n) - prompt_len
        cmp_tokens += n - prompt_len
    return {"value": tot_new / tot_dt, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_prompts} prompts of {prompt_len} tokens, {tot_new} new tokens in {tot_dt:.1f} s (prefill included), fp32 math; "
                      f"weights upcast in {setup:.0f} s (untimed)",
            "mean_accept": sum(accepts) / len(accepts),
            "identical_tokens_vs_gpu": f"{same_tokens}/{cmp_tokens} new tokens before the first difference, summed over the prompts",
            "first_difference": diffs,
            "all_differences_within_tol": all(d["kind"] == "none" or d["within_tol"] for d in diffs)}


def self_launch(n_ranks, argv):
    """Start n_ranks copies of this script, rank r on GPU r, and wait for them (the parent never touches a GPU and is
    not replaced by a child: fresh processes, rendezvous on 127.0.0.1 at a free port).  Rank 0's stdout is relayed;
    the exit status is the first non-zero status of a rank, whose peers are then stopped instead of left waiting in
    a collective."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            for p in procs:                 # exactly the children started above
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    if out0 and out0[0]:
        sys.stdout.write(out0[0].decode(errors="replace"))
        sys.stdout.flush()
    return rc


def launch_check(a, rank, world):
    """--launch-check: what a rank does around the decode (process group, SUM of tokens / MAX of time over the ranks,
    one line from rank 0) with stand-in numbers instead of a decode; needs no GPU (gloo)."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group(os.environ.get("KG_BENCH_BACKEND", "gloo"), rank=rank, world_size=world)
    tot = torch.tensor([100.0 + rank, 1.0 + rank], dtype=torch.float64)
    ranks = [None] * world
    if world > 1:
        tmax = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "0"))))
        tot[1] = tmax[1]
    else:
        ranks = [(0, 0)]
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "tp": a.tp, "ranks": ranks, "tokens": float(tot[0]), "t_job": float(tot[1])}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main_tp(a, shape, L, ds, hs, tdt, dev, dist, rank, world):
    """--tp N: every group of N ranks holds ONE model whose remaining layers are split N ways (kangaroo_amd/tp.py);
    the draft stack is replicated.  Groups decode disjoint prompts; the job's tokens are counted once per group.
    --tp-exchange p2p (default): the library holds the shards and the whole loop runs on the device as in the
    single-GPU case, the partial sums exchanged by the one-shot peer-to-peer kernel inside the captured round.
    --tp-exchange rccl: the loop is driven pass by pass from the host; a pass over the split layers (its kernels and
    torch.distributed all_gathers between them) is one captured graph per row count, replayed at changing positions
    (--tp-no-graphs launches it kernel by kernel)."""
    from kangaroo_amd.inference import kangaroo_forward, kangaroo_forward_stepwise, vanilla_forward
    from kangaroo_amd.tp import DistGroup, LocalGroup, NativeTPKangarooModel, TPKangarooModel, TPPlan
    native = a.tp_exchange == "p2p"
    pg = None
    if a.tp_local:
        n_groups, gidx = world, rank
    else:
        if dist is None or world % a.tp:
            sys.exit(f"bench.py --tp {a.tp}: launch a multiple of {a.tp} ranks (WORLD_SIZE is {world}), or rehearse on one GPU with --tp-local")
        n_groups, gidx = world // a.tp, rank // a.tp
        pgs = [dist.new_group(list(range(g * a.tp, (g + 1) * a.tp))) for g in range(n_groups)]
        pg = pgs[gidx]
    base_sd, adapter_sd = synth_state_dicts(shape, L, seed=0, dtype=tdt, recipe=Recipe.agreeable(ds, hs), device=dev)
    if native:
        model = NativeTPKangarooModel(shape, base_sd, adapter_sd, tdt, L, TPPlan.local(a.tp) if a.tp_local else TPPlan.distributed(pg), device=dev)
    else:
        model = TPKangarooModel(shape, base_sd, adapter_sd, tdt, L, LocalGroup(a.tp) if a.tp_local else DistGroup(pg), device=dev, graphs=not a.tp_no_graphs)
    del base_sd, adapter_sd
    torch.cuda.empty_cache()
    K, W = a.steps, a.warmup
    allp = [p for p in spec_bench_prompts(shape.vocab_size, 4 * (K + W) * n_groups) if p.input_ids.shape[1] <= 256][:(K + W) * n_groups]
    mine = allp[gidx * (K + W):(gidx + 1) * (K + W)]
    warm, timed = mine[:W], mine[W:]
    new = min(a.max_new_tokens, 128)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def run(p, steps):
        if native:
            # plain decoding = kg_generate's vanilla mode over the same sharded kernels (its remaining layers in a verify row's arithmetic)
            if steps == 0:
                return vanilla_forward(p.input_ids.to(dev), model, -1, new)
            return kangaroo_forward(p.input_ids.to(dev), model, -1, new, EARLY_STOP_LAYER=L, SPECULATIVE_DECODING_STEPS=steps, threshold=a.threshold)
        model.head_model.single_rows_in_verify_arithmetic = (steps == 0)      # the comparator's lone rows in a verify row's arithmetic
        return kangaroo_forward_stepwise(p.input_ids.to(dev), model, -1, new, EARLY_STOP_LAYER=L, SPECULATIVE_DECODING_STEPS=steps, threshold=a.threshold)

    for p in warm:
        run(p, a.spec_steps)
    barrier()
    t0 = time.time()
    outs = [run(p, a.spec_steps) for p in timed]
    barrier()
    elapsed = time.time() - t0
    new_tokens = sum(o[1] for o in outs)
    # plain greedy decoding over the same kernels: the same loop with no draft steps (one shallow pass + one verify row per token)
    torch.cuda.synchronize()
    tv0 = time.time()
    if native:
        run(timed[0], 0)                                   # graph capture of the plain-decoding step outside its timed region
        torch.cuda.synchronize()
        tv0 = time.time()
    van = [run(p, 0) for p in timed]
    torch.cuda.synchronize()
    tv = time.time() - tv0
    agree = sum(int(v[0][0][:min(len(v[0][0]), len(o[0][0]))] == o[0][0][:min(len(v[0][0]), len(o[0][0]))]) for v, o in zip(van, outs))
    leader = a.tp_local or (rank % a.tp == 0)
    tot = torch.tensor([float(new_tokens) if leader else 0.0, elapsed], dtype=torch.float64, device=dev if dist is None or dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        total_tokens, t_job = float(tot[0]), float(tmax[1])
    else:
        total_tokens, t_job = float(tot[0]), elapsed
    acc = [x for o in outs for x in o[3]]
    eng = model.engine
    us, wbytes = eng.bench_op("gateup", 1, iters=32)
    ach = wbytes / (us * 1e-6) / 1e9
    res = {"metric": "decode tokens/s at bsz=1 (Kangaroo, prefill included)", "value": total_tokens / t_job, "unit": "tokens/s",
           "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * t_job / max(1, K), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": f"{a.model} shape {'fp16' if a.dtype == 'f16' else 'bf16'} Kangaroo exitlayer={L} steps={a.spec_steps} thr={a.threshold} greedy, "
                                  f"synthetic prompts <= 256 tokens, max_new_tokens={new}, remaining layers tensor-parallel x{a.tp} "
                                  f"({'all ranks rehearsed on one GPU' if a.tp_local else ('one-shot peer-to-peer exchange kernel' if native else 'RCCL all_gather') + ' + rank-order sum'}), "
                                  + ("device-side loop, one captured graph per round" if native else f"pass-by-pass loop, {'one captured graph per pass' if not a.tp_no_graphs else 'kernel-by-kernel launches'}"),
                      "prompts_per_group": K, "parallelism": f"tp{a.tp} x {n_groups} group(s)"},
           "mean_accept": sum(acc) / max(1, len(acc)),
           "vanilla_tokens_per_s_rank0": sum(v[1] for v in van) / tv, "speedup_vs_vanilla": (new_tokens / elapsed) / (sum(v[1] for v in van) / tv),
           "lossless": f"{agree}/{len(timed)} prompts token-identical to plain greedy decoding over the same tensor-parallel kernels",
           "roofline": {"bound": "hbm", "kernel": "kg_linear_kernel<GEMV, RMSNorm, SiLU*mul> (gate/up projection of a replicated draft layer)",
                        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                        "us_per_launch": us, "algorithmic_bytes_per_launch": wbytes},
           "cpu_baseline": None}
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6, help="prompts decoded per rank inside the timed region")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="vicuna-7b")
    ap.add_argument("--max-new-tokens", type=int, default=1024)
    ap.add_argument("--spec-steps", type=int, default=6)
    ap.add_argument("--threshold", type=float, default=0.6)
    ap.add_argument("--recipe", default=f"{CAL_RECIPE[0]}:{CAL_RECIPE[1]}")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"], help="storage / MFMA input type of weights, activations and caches")
    ap.add_argument("--tp", type=int, default=1, help="tensor-parallel degree of the remaining-layer verify pass (the >= 33B-class configuration): "
                    "ranks form groups of this many GPUs, one replica per group")
    ap.add_argument("--tp-local", action="store_true", help="rehearse all --tp ranks of a group on ONE GPU, shard after shard (no process group)")
    ap.add_argument("--tp-exchange", default="p2p", choices=["p2p", "rccl"], help="p2p: shards held by the library, exchange kernel inside the device-side round "
                    "(ranks map each other's exchange buffers); rccl: host-driven passes with torch.distributed all_gather")
    ap.add_argument("--tp-no-graphs", action="store_true", help="launch the tensor-parallel pass kernel by kernel instead of replaying one captured graph per row count")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-prompts", type=int, default=4, help="prompts of the bounded CPU-baseline sample (32 tokens in, 32 out each)")
    ap.add_argument("--no-vanilla", action="store_true")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, form the process group, aggregate stand-in numbers and print one line: "
                    "the launch path without a decode (runs without a GPU under KG_BENCH_BACKEND=gloo)")
    a = ap.parse_args()

    # ranks this job needs: --gpus N replicas, or whole tensor-parallel groups
    want = a.gpus if (a.tp <= 1 or a.tp_local) else max(a.gpus, a.tp)
    if a.tp > 1 and not a.tp_local and want % a.tp:
        sys.exit(f"bench.py: --gpus {a.gpus} is not a multiple of --tp {a.tp}")
    if "WORLD_SIZE" not in os.environ and want > 1:
        sys.exit(self_launch(want, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank == 0:
        # a line on stderr every 45 s: building the weights of the larger shapes takes minutes without any output,
        # and job runners take a silent process for a hung one
        import threading
        t_start = time.time()
        def _heartbeat():
            while True:
                time.sleep(45)
                print(f"[bench] {a.model}: running, {time.time() - t_start:.0f} s", file=sys.stderr, flush=True)
        threading.Thread(target=_heartbeat, daemon=True).start()
    if want != world:
        sys.exit(f"bench.py: --gpus {a.gpus} --tp {a.tp} needs {want} rank(s) but WORLD_SIZE is {world}: run it without a launcher "
                 f"(it starts its own ranks) or with `python -m torch.distributed.run --nnodes=1 --nproc-per-node {want} --master-addr 127.0.0.1 bench.py ...`")
    if a.launch_check:
        return launch_check(a, rank, world)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal knobs for a box with fewer GPUs than ranks: KG_BENCH_BACKEND=gloo KG_BENCH_DEVICE=0
        backend = os.environ.get("KG_BENCH_BACKEND", "nccl")
        if "KG_BENCH_DEVICE" in os.environ:
            local_rank = int(os.environ["KG_BENCH_DEVICE"])
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    shape = get_shape(a.model)
    L = DEFAULT_EXIT_LAYER.get(a.model, 2)
    ds, hs = [float(x) for x in a.recipe.split(":")]
    tdt = torch.float16 if a.dtype == "f16" else torch.bfloat16
    if a.tp > 1:
        return main_tp(a, shape, L, ds, hs, tdt, dev, dist, rank, world)
    base_sd, adapter_sd = synth_state_dicts(shape, L, seed=0, dtype=tdt, recipe=Recipe.agreeable(ds, hs), device=dev)
    model = KangarooModel.from_state_dicts(shape, base_sd, adapter_sd, tdt, L, device=dev)
    eng = model.engine

    K, W = a.steps, a.warmup
    allp = spec_bench_prompts(shape.vocab_size, (K + W) * world)
    mine = allp[rank * (K + W):(rank + 1) * (K + W)]     # contiguous shard per rank
    warm, timed = mine[:W], mine[W:]
    eos = -1                                              # random weights have no meaningful eos: decode the full budget
    kw = dict(steps=a.spec_steps, threshold=a.threshold, eos_token_id=eos)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for p in warm:
        eng.generate(p.input_ids[0], a.max_new_tokens, mode="kangaroo", **kw)
    if not warm:
        eng.generate(timed[0].input_ids[0], 16, mode="kangaroo", **kw)       # graph capture outside the timed region
    barrier()
    t0 = time.time()
    stats, outs = [], []
    for p in timed:
        out, st = eng.generate(p.input_ids[0], a.max_new_tokens, mode="kangaroo", **kw)
        stats.append(st)
        outs.append(out)
    barrier()
    elapsed = time.time() - t0
    new_tokens = sum(s.new_tokens for s in stats)
    tot = torch.tensor([float(new_tokens), elapsed], dtype=torch.float64, device=dev if dist is None or dist.get_backend() == "nccl" else "cpu")
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        total_tokens, t_job = float(tot[0]), float(tmax[1])
    else:
        total_tokens, t_job = float(tot[0]), elapsed

    # ---- own vanilla decode on the same kernels, same prompts (not inside the timed region) ----
    van_tps = speedup = van_gemv_tps = None
    lossless = None
    if not a.no_vanilla:
        eng.generate(timed[0].input_ids[0], 16, mode="vanilla", eos_token_id=eos)
        torch.cuda.synchronize()
        tv0 = time.time()
        vtok = 0
        agree = 0
        for p, st, out in zip(timed, stats, outs):
            vout, vst = eng.generate(p.input_ids[0], st.new_tokens, mode="vanilla", eos_token_id=eos)
            vtok += vst.new_tokens
            agree += int(vout == out)
        torch.cuda.synchronize()
        van_tps = vtok / (time.time() - tv0)
        speedup = (new_tokens / elapsed) / van_tps
        lossless = f"{agree}/{len(timed)} prompts token-identical to vanilla greedy"
        # plain decoding with every layer in the 1-row GEMV kernels: the fastest plain decoding these kernels
        # offer (the comparator above runs the remaining layers in a verify row's arithmetic, which is what
        # makes token identity exact and costs it a few per cent)
        eng.set_tuning(vanilla_gemv=2)
        eng.generate(timed[0].input_ids[0], 16, mode="vanilla", eos_token_id=eos)
        torch.cuda.synchronize()
        tg0 = time.time()
        gtok = 0
        for p, st in zip(timed, stats):
            gtok += eng.generate(p.input_ids[0], st.new_tokens, mode="vanilla", eos_token_id=eos)[1].new_tokens
        torch.cuda.synchronize()
        van_gemv_tps = gtok / (time.time() - tg0)
        eng.set_tuning(vanilla_gemv=1)

    # ---- achieved bytes / roofline ----
    dec_ms = sum(s.decode_ms for s in stats)
    bytes_dec = sum(round_bytes(shape, L, d) for s in stats for d in s.draft_lengths)
    kv_dec = 0
    for p, s in zip(timed, stats):
        T = p.input_ids.shape[1]
        for d, acc in zip(s.draft_lengths, s.accept_lengths):
            kv_dec += round_kv_bytes(shape, L, d, T)
            T += acc
    path_gbs = bytes_dec / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
    path_kv_gbs = (bytes_dec + kv_dec) / (dec_ms * 1e-3) / 1e9 if dec_ms > 0 else 0.0
    rows = a.spec_steps + 1
    us, wbytes = eng.bench_op("gateup", rows, iters=4 * shape.num_hidden_layers)
    ach = wbytes / (us * 1e-6) / 1e9
    # HBM bytes per launch from the PMC passes over this same kernel and shape (counters need runs of their own: the
    # file records command, raw counters, the gfx950 correction and the commit of the library they were taken on)
    traffic = traffic_src = None
    pmc_file = os.path.join(ROOT, "profiles", "r03_pmc_gateup_rows7.json")
    if a.model == "vicuna-7b" and rows == 7 and os.path.exists(pmc_file):
        with open(pmc_file) as f:
            pj = json.load(f)
        traffic, traffic_src = pj.get("hbm_bytes_per_launch"), f"profiles/r03_pmc_gateup_rows7.json (library built from commit {pj.get('built_from_commit')})"
    roofline = {"bound": "hbm", "kernel": "kg_linear_kernel<f16, MFMA, RMSNorm, SiLU*mul> (gate/up projection of the verify pass)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "us_per_launch": us, "algorithmic_bytes_per_launch": wbytes}
    per_kernel = {}
    for op in ("qkv", "o", "gateup", "down", "head"):
        for r in (1, rows):
            if op == "gateup" and r == rows:          # the dominant kernel is timed once: the roofline record's figure
                per_kernel[f"{op}_rows{r}"] = {"us": round(us, 2), "GB/s": round(ach, 1)}
                continue
            u, b = eng.bench_op(op, r, iters=2 * shape.num_hidden_layers)
            per_kernel[f"{op}_rows{r}"] = {"us": round(u, 2), "GB/s": round(b / (u * 1e-6) / 1e9, 1)}

    acc_all = [x for s in stats for x in ([1] + s.accept_lengths)]
    d_all = [x for s in stats for x in s.draft_lengths]
    res = {
        "metric": "decode tokens/s at bsz=1 (Kangaroo, prefill included)", "value": total_tokens / t_job, "unit": "tokens/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * t_job / max(1, K), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {"workload": f"{a.model} shape {'fp16' if a.dtype == 'f16' else 'bf16'} Kangaroo exitlayer={L} steps={a.spec_steps} thr={a.threshold} greedy, "
                               f"Spec-Bench-shaped synthetic prompts, max_new_tokens={a.max_new_tokens}, random-init weights "
                               f"(deep_scale={ds}, head_scale={hs})",
                   "prompts_per_gpu": K, "parallelism": f"prompt-sharded x{world} (replicas, no collectives)"},
        "mean_accept": sum(acc_all) / max(1, len(acc_all)), "mean_draft": sum(d_all) / max(1, len(d_all)),
        "decode_only_tokens_per_s_rank0": (new_tokens - len(stats)) / (dec_ms * 1e-3) if dec_ms > 0 else None,
        "vanilla_tokens_per_s_rank0": van_tps, "speedup_vs_vanilla": speedup, "lossless": lossless,
        "vanilla_gemv_tokens_per_s_rank0": van_gemv_tps,
        "speedup_vs_vanilla_gemv": (new_tokens / elapsed) / van_gemv_tps if van_gemv_tps else None,
        "hbm_GBps_decode_loop_rank0": path_gbs, "hbm_frac_decode_loop": path_gbs / HBM_PEAK_GBS,
        "hbm_GBps_decode_loop_with_kv_rank0": path_kv_gbs, "hbm_frac_decode_loop_with_kv": path_kv_gbs / HBM_PEAK_GBS,
        "roofline": roofline, "kernels": per_kernel,
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(shape, L, base_sd, adapter_sd, a.spec_steps, a.threshold, model,
                                           n_prompts=a.cpu_prompts)
    elif rank == 0:
        res["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
