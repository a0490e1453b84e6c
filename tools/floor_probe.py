#!/usr/bin/env python3
"""Where the multi-GPU schedule's compute floor differs from the single-GPU step, on one GPU (bench.py: multi_gpu_schedule_compute_floor):
the same DiT-B training step as bench.py in four set-ups, alternated so that the box's drift cancels:
  plain            the single-GPU step (one backward call, weight gradients on the side stream)
  phased           force_phased + stub_exchange: a backward call per block, early AdamW on finished slices, collectives left out
  phased+queues    ... with osud_set_gemm_dynamic_tiles(1): GEMM tiles / weight-gradient K-chunks / attention heads from ticket queues
  plain+queues     the single call with the queues on
usage: floor_probe.py [--steps 30] [--rounds 2]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from osu_diffusion_amd import _lib  # noqa: E402
from osu_diffusion_amd.diffusion import create_diffusion  # noqa: E402
from osu_diffusion_amd.models import DiT_models  # noqa: E402
from osu_diffusion_amd.synthetic import randomize_zero_init, synthetic_windows  # noqa: E402
from osu_diffusion_amd.training import NativeTrainer  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--only", default=None)
args = ap.parse_args()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = DiT_models["DiT-B"](num_classes=52670, context_size=144, class_dropout_prob=0.2, precision="bf16")
model = randomize_zero_init(model.to(dev), seed=0).train()
diffusion = create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True)
batches = []
for i in range(4):
    (x, o, c), y = synthetic_windows(256, 128, 52670, seed=i, train_offsets=True)
    batches.append(((x.to(dev), o.to(dev), c.to(dev)), y.to(dev)))
L = _lib.lib()
SETUPS = {"plain": (False, 0), "phased": (True, 0), "phased+queues": (True, 1), "plain+queues": (False, 1)}
for r in range(args.rounds):
    for name, (phased, dyn) in SETUPS.items():
        if args.only and name != args.only:
            continue
        _lib.check(L.osud_set_gemm_dynamic_tiles(dyn))
        tr = NativeTrainer(model, diffusion, lr=1e-4, force_phased=phased, stub_exchange=phased)
        for i in range(6):
            (x, o, c), y = batches[i % 4]
            tr.step(x, o, c, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            (x, o, c), y = batches[i % 4]
            tr.step(x, o, c, y)
        tr.finish_exchange()
        torch.cuda.synchronize()
        print(f"round {r} {name:15s} {(time.perf_counter() - t0) / args.steps * 1e3:.3f} ms/step", flush=True)
        del tr
        torch.cuda.empty_cache()
_lib.check(L.osud_set_gemm_dynamic_tiles(-1))
