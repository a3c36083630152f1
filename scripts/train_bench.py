"""Adapter training step at a model shape: synthetic hidden states, `--steps` timed steps after `--warmup`.
One process per GPU (torchrun) makes it data parallel; prints one JSON line on rank 0.
usage: python scripts/train_bench.py [--model vicuna-7b] [--bs 4] [--seq 2048] [--steps 10] [--warmup 3] [--dtype float16]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd.config import get_shape
from kangaroo_amd.train import AdapterTrainer, PARAM_KEYS

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="vicuna-7b"); ap.add_argument("--bs", type=int, default=4); ap.add_argument("--seq", type=int, default=2048)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--warmup", type=int, default=3); ap.add_argument("--dtype", default="float16")
a = ap.parse_args()
world, rank, local = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist = None
if world > 1:
    import torch.distributed as dist
    dist.init_process_group("nccl")
shape = get_shape(a.model)
d, V, dt = shape.hidden_size, shape.vocab_size, getattr(torch, a.dtype)
g = torch.Generator().manual_seed(0)          # same initial adapter on every rank
lin = lambda n, k: torch.randn(n, k, generator=g) * 0.02
sd = {PARAM_KEYS["q_proj"]: lin(shape.q_dim, d), PARAM_KEYS["k_proj"]: lin(shape.kv_dim, d), PARAM_KEYS["v_proj"]: lin(shape.kv_dim, d),
      PARAM_KEYS["o_proj"]: lin(d, shape.q_dim), PARAM_KEYS["input_layernorm"]: torch.ones(d), PARAM_KEYS["norm"]: torch.ones(d)}
head = lin(V, d)
tr = AdapterTrainer(shape, sd, head, dtype=dt, lr=5e-5, num_warmup_steps=2000, total_steps=800000)
gb = torch.Generator(device="cuda").manual_seed(1 + rank)
def batch():
    lens = torch.randint(a.seq // 2, a.seq + 1, (a.bs,), generator=g).tolist()
    lens[0] = a.seq
    am = torch.zeros(a.bs, a.seq, dtype=torch.int32); lm = torch.zeros(a.bs, a.seq, dtype=torch.int32)
    for b, n in enumerate(lens):
        am[b, :n] = 1; lm[b, :n - 1] = 1
    return {"hidden_states_early": torch.randn(a.bs, a.seq, d, generator=gb, device="cuda", dtype=torch.float32).to(dt),
            "target": torch.randn(a.bs, a.seq, d, generator=gb, device="cuda", dtype=torch.float32).to(dt), "attention_mask": am, "loss_mask": lm}
data = [batch() for _ in range(2)]
for i in range(a.warmup):
    out = tr.step(data[i % 2])
torch.cuda.synchronize()
if dist: dist.barrier()
t0 = time.time()
for i in range(a.steps):
    out = tr.step(data[i % 2])
torch.cuda.synchronize()
if dist: dist.barrier()
dtm = time.time() - t0
R = a.bs * a.seq
flop = 3 * 2 * R * V * d + 3 * 2 * R * d * (2 * shape.q_dim + 2 * shape.kv_dim) + 6 * 2 * a.bs * shape.num_attention_heads * a.seq * a.seq * shape.head_dim
if rank == 0:
    print(json.dumps({"metric": "adapter training tokens/s (padded positions included)", "value": world * R * a.steps / dtm, "ms_per_step": dtm / a.steps * 1e3,
                      "n_gpus": world, "model": a.model, "bs_per_gpu": a.bs, "seq": a.seq, "dtype": a.dtype, "loss": out["loss"], "prob_acc": out["prob_acc"],
                      "gemm_tflop_per_step": flop / 1e12, "gemm_tflops_per_gpu": flop / (dtm / a.steps) / 1e12,
                      "peak_mem_gb": torch.cuda.max_memory_allocated() / 2 ** 30}))
