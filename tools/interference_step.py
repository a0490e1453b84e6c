#!/usr/bin/env python3
"""A whole DiT-B training step while another kernel holds a few compute units for the entire time (the occupier of
tools/probes/occupy.hip, standing in for collectives that overlap the backward): fixed-stride vs queued GEMM tiles.
    python tools/interference_step.py [held_cus]"""
import ctypes
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

occ = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "liboccupy.so"))
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
held = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)
torch.manual_seed(0)
model = randomize_zero_init(DiT_models["DiT-B"](num_classes=52670, context_size=144, class_dropout_prob=0.2).to(dev), seed=0).train()
tr = NativeTrainer(model, create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True))
(x, o, c), y = synthetic_windows(256, 128, 52670, seed=1, train_offsets=True)
x, o, c, y = x.to(dev), o.to(dev), c.to(dev), y.to(dev)
L = _lib.lib()


def run(k, drain=True):
    if drain:
        torch.cuda.synchronize()
    main = torch.cuda.current_stream(dev)
    t0 = time.perf_counter()
    for _ in range(k):
        tr.step(x, o, c, y)
    main.synchronize()  # not the device: the occupier on the side stream is meant to outlive this
    return (time.perf_counter() - t0) / k * 1e3


for dyn in (0, 1):
    _lib.check(L.osud_set_gemm_dynamic_tiles(dyn))
    run(5)
    alone = run(15)
    assert occ.occupy(held, 1500000, 96, ctypes.c_void_p(side.cuda_stream)) == 0  # 1.5 s
    time.sleep(0.01)
    beside = run(15, drain=False)
    t_left = time.perf_counter()
    side.synchronize()
    t_left = time.perf_counter() - t_left
    print(f"queued tiles = {dyn}: {alone:.2f} ms per step alone, {beside:.2f} ms with {held} CUs held "
          f"(the occupier outlived the timed steps by {t_left * 1e3:.0f} ms)", flush=True)
