"""Worker of tests/test_gpu_multiproc.py: one data-parallel rank of a tiny fp32-tier training run.  All ranks share GPU 0
(the GPU box has one device) and exchange gradients through gloo, so the whole multi-process path — rank-0 broadcast,
phased backward with per-slice async all-reduce, 1/world folded into the optimizer — runs exactly as under RCCL."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd.diffusion import create_diffusion  # noqa: E402
from osu_diffusion_amd.models import DiT  # noqa: E402
from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows  # noqa: E402
from osu_diffusion_amd.training import NativeTrainer  # noqa: E402


def build(seed):
    torch.manual_seed(seed)
    m = DiT(depth=2, hidden_size=128, num_heads=2, context_size=144, num_classes=10, class_dropout_prob=0.0, precision="fp32")
    return randomize_zero_init(m.to(f"cuda:{torch.cuda.current_device()}"), seed=seed).train()


def batch():
    (x, o, c), y = synthetic_windows(8, 64, 10, seed=11)
    t = torch.tensor([0, 1, 17, 250, 500, 731, 998, 999])
    noise = torch.randn(8, 2, 64, generator=torch.Generator().manual_seed(12))
    return x, o, c, y, t, noise


def trainer_kw():
    """Schedule switches of the worker's trainer, set by the test that launches it (the library itself reads no such variables)."""
    return dict(force_phased=os.environ.get("OSUD_TEST_FORCE_PHASED", "0") == "1",
                native_comm=os.environ.get("OSUD_TEST_NATIVE_COMM", "0") == "1",
                stub_exchange=os.environ.get("OSUD_TEST_STUB", "0") == "1")


def run(rank, world, steps=2, zero1=False, wire=None, full_state=False):
    # every rank starts from DIFFERENT weights: the constructor's rank-0 broadcast must make them equal
    model = build(100 + rank)
    tr = NativeTrainer(model, create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True), lr=1e-3,
                       shard_optimizer=zero1, wire_dtype=wire, **trainer_kw())
    x, o, c, y, t, noise = batch()
    per = 8 // world
    sl = slice(rank * per, (rank + 1) * per)
    for _ in range(steps):
        tr.step(x[sl], o[sl], c[sl], y[sl], t=t[sl], noise=noise[sl])
    torch.cuda.synchronize()
    if full_state:  # checkpoint() gathers the sharded moments / EMA first
        tr.sync_sharded_state()
        return (tr.arena.flat.detach().cpu().clone(), tr.ema_arena.flat.detach().cpu().clone(), tr.exp_avg.detach().cpu().clone(),
                tr.exp_avg_sq.detach().cpu().clone())
    return tr.arena.flat.detach().cpu().clone(), tr.ema_arena.flat.detach().cpu().clone()


if __name__ == "__main__":
    out_dir = sys.argv[1]
    # (OSUD_TEST_PER_RANK_DEVICE=1: one GPU per rank -- the only placement RCCL accepts for more than one rank; multi-GPU nodes only)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("OSUD_TEST_PER_RANK_DEVICE", "0") == "1" else 0)
    dist.init_process_group(os.environ.get("OSUD_DIST_BACKEND", "gloo"))
    rank, world = dist.get_rank(), dist.get_world_size()
    mode = os.environ.get("OSUD_TEST_MODE", "allreduce")  # allreduce | zero1 | zero1_bf16
    if mode == "zero1_ckpt":
        # train.py's save path with the sharded optimizer: checkpoint() alone on the saving rank must refuse (it would enter an
        # all-gather nobody else joins); after the collective sync on EVERY rank, rank 0 alone writes the file, then the barrier
        model = build(100 + rank)
        tr = NativeTrainer(model, create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True), lr=1e-3, shard_optimizer=True,
                           **trainer_kw())
        x, o, c, y, t, noise = batch()
        per = 8 // world
        sl = slice(rank * per, (rank + 1) * per)
        for _ in range(2):
            tr.step(x[sl], o[sl], c[sl], y[sl], t=t[sl], noise=noise[sl])
        refused = False
        if rank == 0:
            try:
                tr.checkpoint()
            except RuntimeError as e:
                refused = "sync_sharded_state" in str(e)
        tr.sync_sharded_state()
        if rank == 0:
            ck = tr.checkpoint({"note": "zero1"})
            ck["refused"] = refused
            torch.save(ck, os.path.join(out_dir, "ckpt.pt"))
        dist.barrier()
        torch.save({"exp_avg": tr.exp_avg.cpu(), "ema": tr.ema_arena.flat.cpu()}, os.path.join(out_dir, f"rank{rank}.pt"))
        dist.barrier()
        dist.destroy_process_group()
        sys.exit(0)
    if mode == "allreduce":
        flat, ema = run(rank, world)
        torch.save({"flat": flat, "ema": ema}, os.path.join(out_dir, f"rank{rank}.pt"))
    else:
        flat, ema, m1, m2 = run(rank, world, zero1=True, wire=torch.bfloat16 if mode.endswith("bf16") else None, full_state=True)
        torch.save({"flat": flat, "ema": ema, "exp_avg": m1, "exp_avg_sq": m2}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
