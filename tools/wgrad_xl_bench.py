#!/usr/bin/env python3
"""Weight-gradient launches at DiT-XL's shapes (D = 1152, 32768 tokens): time per shape with the geometry the launcher picks (256 x 192 /
192 x 256 where a side is a multiple of 192) and with the padded 256 x 256 geometry (osud_set_gemm_dynamic_tiles(1) keeps that one)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
D = 1152
shapes = [("in_proj", 3 * D, D), ("out_proj", D, D), ("fc1", 4 * D, D), ("fc2", D, 4 * D)]
for dyn in (0, 1):
    _lib.check(L.osud_set_gemm_dynamic_tiles(dyn))
    tot_us = tot_fl = 0.0
    for name, Ny, Nx in shapes:
        # (one spare token row: the padded 256-wide geometry stages up to 128 features past a row's end -- the library's own buffers carry that
        #  slack, dit.h: dev_alloc -- and the LAST row's over-read must stay inside the allocation)
        P = (torch.randn(M + 1, Ny, device=dev) * 0.05).to(torch.bfloat16)[:M]
        Q = torch.randn(M + 1, Nx, device=dev).to(torch.bfloat16)[:M]
        out = torch.empty(Ny, Nx, device=dev)
        ws = torch.empty(32 * Ny * Nx, device=dev)
        for _ in range(3):
            _lib.check(L.osud_op_wgrad(_lib.ptr(P), Ny, _lib.ptr(Q), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            _lib.check(L.osud_op_wgrad(_lib.ptr(P), Ny, _lib.ptr(Q), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        fl = 2.0 * M * Ny * Nx
        tot_us += us
        tot_fl += fl
        print(f"dyn={dyn} {name:9s} {Ny}x{Nx}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
        del P, Q, out, ws
    print(f"dyn={dyn} all four: {tot_us:8.1f} us  {tot_fl / tot_us / 1e6:7.1f} TFLOP/s", flush=True)
_lib.check(L.osud_set_gemm_dynamic_tiles(-1))
