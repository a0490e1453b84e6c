#!/usr/bin/env python3
"""Is a persistent MFMA GEMM slower on fewer compute units?  Under a power cap it should not be, by much: fc1 (32768 x 3072 x 768, 1536
tiles) on OSUD_GEMM_CUS = 256 / 192 / 128 workgroups (6 / 8 / 12 tiles each), and the 8192^3 GEMM.  Run once per setting."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
def bench(name, epi, My, Nx, K, iters=40):
    Y = (torch.randn(My, K, device=dev)).to(torch.bfloat16); X = (torch.randn(Nx, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(My, Nx, dtype=torch.bfloat16, device=dev); bias = torch.randn(max(My, Nx), device=dev) * 0.02
    go = lambda: _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias), None, 0, 0, 0, None))
    for _ in range(5): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"[CUS={os.environ.get('OSUD_GEMM_CUS', 'all')}] {name:18s} {us:8.1f} us  {2.0 * My * Nx * K / us / 1e6:7.1f} TFLOP/s", flush=True)
bench("fc1 bias+gelu", _lib.EPI_BIAS_GELU_TE, 32768, 3072, 768)
bench("fc1 plain", _lib.EPI_BIAS_TE, 32768, 3072, 768)
bench("8192^3", _lib.EPI_BIAS_TE, 8192, 8192, 8192, iters=10)
