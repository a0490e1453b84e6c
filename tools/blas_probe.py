#!/usr/bin/env python3
"""Calibration only (not a product path): what the vendor GEMM (torch.mm -> hipBLASLt / rocBLAS) reaches on the path's shapes,
next to the library's own kernels (tools/gemm_shapes.py)."""
import time

import torch

dev = "cuda:0"
shapes = [(32768, 3072, 768), (32768, 768, 3072), (32768, 2304, 768), (32768, 768, 768), (16384, 3072, 768), (16384, 768, 3072),
          (16384, 2304, 768), (16384, 768, 768), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    it = 30
    for _ in range(it):
        c = a @ w.t()
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / it * 1e3
    print(f"torch.mm bf16 {M}x{N}x{K}: {us:8.1f} us  {2 * M * N * K / us / 1e6:8.1f} TFLOP/s")

print("weight-gradient shapes: dW (N x K) = dY^T (N x M) @ X (M x K), operands token-major as the backward has them")
for M, N, K in [(32768, 3072, 768), (32768, 768, 3072), (32768, 2304, 768), (32768, 768, 768)]:
    dy = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    for _ in range(5):
        c = dy.t() @ x
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30):
        c = dy.t() @ x
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 30 * 1e3
    print(f"torch.mm bf16 wgrad {N}x{K} over M={M}: {us:8.1f} us  {2 * M * N * K / us / 1e6:8.1f} TFLOP/s (bf16 output)")
