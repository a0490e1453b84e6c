#!/usr/bin/env python3
"""Time the attention core of a DiT-B training step in isolation (HIP events): forward (osud_op_attention) and backward
(osud_op_attention_bwd) at N = 256 windows x 12 heads x T = 128, next to the bytes each has to move.

  python tools/attn_bench.py [N] [T]        # OSUD_OPTIONS=attn_bwd_kernel=1 selects the one-workgroup-per-head backward kernel
  H=16 HD=72 python tools/attn_bench.py 128 256     # DiT-XL's shape
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H, hd = int(os.environ.get("H", "12")), int(os.environ.get("HD", "64"))
D, M = H * hd, N * T
bf = torch.bfloat16
qkv = torch.randn(M, 3 * D, device=dev).to(bf)
dout = torch.randn(M, D, device=dev).to(bf)
out = torch.zeros(M, D, device=dev, dtype=bf)
dqkv = torch.zeros(M, 3 * D, device=dev, dtype=bf)
lse = torch.zeros(N, H, T, device=dev)
ws = torch.zeros(N, H, T, device=dev)


def timeit(name, go, nbytes, iters=30):
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:34s} {us:8.1f} us   {nbytes / 1e6:7.1f} MB   {nbytes / us / 1e6:5.2f} TB/s", flush=True)


fwd = lambda: _lib.check(L.osud_op_attention(0, _lib.ptr(qkv), 3 * D, None, _lib.ptr(out), N, T, T, M, H, hd, None))
bwd = lambda: _lib.check(L.osud_op_attention_bwd(0, _lib.ptr(qkv), _lib.ptr(dout), _lib.ptr(out), _lib.ptr(lse), _lib.ptr(dqkv), N, T, H, hd,
                                                 _lib.ptr(ws), None))
timeit("attention forward", fwd, M * D * 2 * 4)
# realistic statistics for the backward: lse of the actual scores (log2 domain)
q, k, _ = (qkv[:, i * D:(i + 1) * D].float().reshape(N, T, H, hd).transpose(1, 2) for i in range(3))
lse.copy_(torch.logsumexp(q @ k.transpose(-1, -2) / hd ** 0.5, -1) * 1.4426950408889634)
timeit("attention backward", bwd, M * D * 2 * 8)
