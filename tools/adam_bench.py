#!/usr/bin/env python3
"""AdamW + EMA over a DiT-B-sized arena next to a device copy of the same bytes (HIP events).  OSUD_LIB=ab/libosud_x.so selects a variant.

  python tools/adam_bench.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
n = int(os.environ.get("N", "130000000"))
p = torch.randn(n, device=dev) * 0.02; g = torch.randn(n, device=dev) * 1e-3
m1 = torch.zeros(n, device=dev); m2 = torch.zeros(n, device=dev); ema = p.clone()
step = [0]


def adam():
    step[0] += 1
    _lib.check(L.osud_adamw_ema_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(ema), n, 1e-4, 0.9, 0.999, 1e-8, 0.0,
                                     step[0], 0.9999, 0, 0, 1.0, None))


def timeit(name, go, nbytes, iters=20):
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{os.path.basename(_lib.LIB_PATH):22s} {name:28s} {us:8.1f} us   {nbytes / us / 1e6:5.2f} TB/s", flush=True)


timeit("adamw_ema", adam, n * 36)
if os.environ.get("COPY"):
    a = torch.empty(n * 36 // 8, device=dev); b = torch.empty_like(a)
    timeit("torch copy_ (same bytes)", lambda: b.copy_(a), a.numel() * 8)
