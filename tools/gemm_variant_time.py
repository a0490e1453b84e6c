#!/usr/bin/env python3
"""Time the phased loop of ONE library build (OSUD_LIB=ab/libosud_<variant>.so) on four training shapes: one line, us per launch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
loop = int(os.environ.get("LOOP", "1")); M = int(os.environ.get("M", "32768")); D = 768
_lib.set_option("gemm_loop", loop)
SH = [("fc2f", _lib.EPI_BIAS_TE, M, D, 4 * D), ("fc1f", _lib.EPI_BIAS_GELU_TE, M, 4 * D, D), ("qkvd", _lib.EPI_NONE_TE, M, D, 3 * D), ("fc2d", _lib.EPI_NONE_TE, M, 4 * D, D)]
out_s = []
for name, epi, My, Nx, K in SH:
    Y = torch.randn(My, K, device=dev).to(torch.bfloat16); X = (torch.randn(Nx, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(My, Nx, dtype=torch.bfloat16, device=dev); bias = torch.randn(max(My, Nx), device=dev) * 0.02
    go = lambda: _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias), None, 0, 0, 0, None))
    for _ in range(5): go()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20): go()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    out_s.append(f"{name} {sorted(ts)[2]:6.1f}")
print(os.path.basename(os.environ.get("OSUD_LIB", "libosud.so")), f"loop={loop}", "  ".join(out_s), flush=True)
