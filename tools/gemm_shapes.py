#!/usr/bin/env python3
"""Time every GEMM shape of a DiT-B training step (M = 32768) per tile geometry (option gemm_tile)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
D, M = 768, int(os.environ.get('M', '32768'))
SHAPES = [("qkv fwd", _lib.EPI_BIAS_TE, M, 3 * D, D, False), ("proj fwd", _lib.EPI_BIAS_TE, M, D, D, False),
          ("fc1 fwd", _lib.EPI_BIAS_GELU_TE, M, 4 * D, D, False), ("fc2 fwd", _lib.EPI_BIAS_TE, M, D, 4 * D, False),
          ("fc2 dgrad", _lib.EPI_NONE_TE, M, 4 * D, D, False), ("fc1 dgrad", _lib.EPI_NONE_TE, M, D, 4 * D, False),
          ("qkv dgrad", _lib.EPI_NONE_TE, M, D, 3 * D, False), ("proj dgrad", _lib.EPI_NONE_TE, M, D, D, False)]
NBUF = int(os.environ.get("NBUF", "1"))  # > 1: rotate over NBUF operand/output sets so that nothing is Infinity-Cache warm
def bench(epi, My, Nx, K, f32out, iters=24):
    Ys = [torch.randn(My, K, device=dev).to(torch.bfloat16) for _ in range(NBUF)]
    X = (torch.randn(Nx, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    outs = [torch.zeros(My, Nx, dtype=torch.float32 if f32out else torch.bfloat16, device=dev) for _ in range(NBUF)]
    bias = torch.randn(max(My, Nx), device=dev) * 0.02
    def go(i):
        _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Ys[i % NBUF]), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(outs[i % NBUF]), Nx, _lib.ptr(bias), None, 0, 0, 0, None))
    for i in range(3): go(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters): go(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters
for name, epi, My, Nx, K, f in SHAPES:
    row = []
    for tile in ("", "256", "192", "128"):
        _lib.set_option("gemm_tile", int(tile) if tile else 0)
        us = bench(epi, My, Nx, K, f)
        row.append(f"{tile or 'auto'}:{us:6.1f}us/{2.0 * My * Nx * K / us / 1e6:5.0f}TF")
    print(f"{name:11s} {My}x{Nx}x{K}  " + "  ".join(row), flush=True)
