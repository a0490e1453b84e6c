#!/usr/bin/env python3
"""Multi-round GEMM launches (more tiles than workgroups: the dynamic tile queue) against torch.mm on the same bf16 operands,
alone and beside an occupier kernel; also repeated launches (the counters must re-arm)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (My, Nx, K) in [(32768, 3072, 768), (32768, 768, 3072), (16384, 2304, 768), (8192, 1536, 256), (65536, 768, 768)]:
    Y = torch.randn(My, K, device=dev).to(torch.bfloat16)
    X = (torch.randn(Nx, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(Nx, device=dev)
    ref = (Y.float() @ X.float().t() + bias).to(torch.bfloat16).float()
    worst = 0.0
    for rep in range(5):
        out = torch.zeros(My, Nx, dtype=torch.bfloat16, device=dev)
        _lib.check(L.osud_op_gemm(0, _lib.EPI_BIAS_TE, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias), None, 0, 0, 0, None))
        torch.cuda.synchronize()
        worst = max(worst, float((out.float() - ref).abs().max()))
    print(f"{My}x{Nx}x{K}: max|d| over 5 launches = {worst:.4f} (bf16 output, |ref| max {float(ref.abs().max()):.1f})", flush=True)
    assert worst <= 0.05 * float(ref.abs().max()) + 0.05
print("ok")
