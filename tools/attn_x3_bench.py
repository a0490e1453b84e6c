#!/usr/bin/env python3
"""Time the split-bf16 attention core at the sampling shape (128 windows x 12 heads x T = 128; HIP events): plane-pair operands in,
plane pairs (PREC 3) or fp16 + e4m3 rows (PREC 4: the tolerance tier) out.  Option attn_fwd_kernel = 1 selects the general kernel.

  python tools/attn_x3_bench.py        # OSUD_LIB=ab/libosud_x.so selects a variant build
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
N, T, H, hd = int(os.environ.get("N", "128")), 128, 12, 64
D, M = H * hd, N * T
qkv = (torch.randn(M, 6 * D, device=dev)).to(torch.bfloat16)  # [hi | lo] planes of Q | K | V (timing only: lo is not a residual here)
out = torch.zeros(M, 4 * D, dtype=torch.uint8, device=dev)
for prec, name in ((_lib.PREC_BF16X3, "bf16x3"), (_lib.PREC_F16F8, "fp16f8")):
    for general in (1, 0):
        _lib.set_option("attn_fwd_kernel", general)
        go = lambda: _lib.check(L.osud_op_attention(prec, _lib.ptr(qkv), 3 * D, None, _lib.ptr(out), N, T, T, M, H, hd, None))
        for _ in range(5):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        nbytes = M * D * (3 * 4 + 4)
        print(f"{os.path.basename(_lib.LIB_PATH):18s} {name} {'general ' if general else 'streamed'} {us:7.1f} us  {nbytes / us / 1e6:5.2f} TB/s", flush=True)
