#!/usr/bin/env python3
"""Time the DiT GEMM shapes through osud_op_gemm (HIP events, random bf16 operands)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")


def bench(name, epi, My, Nx, K, iters=30, f32out=False):
    Yf = torch.randn(My, K, device=dev); Xf = torch.randn(Nx, K, device=dev) / K ** 0.5
    Y = torch.empty(My, K, dtype=torch.bfloat16, device=dev); X = torch.empty(Nx, K, dtype=torch.bfloat16, device=dev)
    L.osud_op_convert(0, _lib.ptr(Yf), _lib.ptr(Y), Yf.numel(), None); L.osud_op_convert(0, _lib.ptr(Xf), _lib.ptr(X), Xf.numel(), None)
    out = torch.zeros(My, Nx, dtype=torch.float32 if f32out else torch.bfloat16, device=dev)
    bias = torch.randn(max(My, Nx), device=dev) * 0.02
    ns = max(1, My // 128)
    gate = torch.randn(ns, Nx, device=dev)
    def go():
        _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias),
                                  _lib.ptr(gate), Nx, 128, ns, None))
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    tf = 2.0 * My * Nx * K / us / 1e6
    print(f"{name:28s} {My:6d}x{Nx:5d}x{K:5d}  {us:8.1f} us  {tf:7.1f} TFLOP/s  ({100 * tf / 2500:.1f}% of bf16 peak)", flush=True)

D = 768
for M in (16384, 32768):
    bench("embed (bias f32)", _lib.EPI_BIAS_F32, M, D, 576, f32out=True)
    bench("qk proj (bias)", _lib.EPI_BIAS_TE, M, 2 * D, D)
    bench("v^T proj (rowbias)", _lib.EPI_ROWBIAS_TE, D, M, D)
    bench("attn out (gate+res)", _lib.EPI_GATE_RES, M, D, D, f32out=True)
    bench("fc1 (bias+gelu)", _lib.EPI_BIAS_GELU_TE, M, 4 * D, D)
    bench("fc2 (gate+res)", _lib.EPI_GATE_RES, M, D, 4 * D, f32out=True)
bench("adaLN all (bias f32)", _lib.EPI_BIAS_F32, 128, 6 * D * 12 + 2 * D, D, f32out=True)
bench("square 4096", _lib.EPI_NONE_TE, 4096, 4096, 4096)
bench("square 8192", _lib.EPI_NONE_TE, 8192, 8192, 8192, iters=10)
