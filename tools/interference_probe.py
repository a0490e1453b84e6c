#!/usr/bin/env python3
"""How much does a GEMM slow down when another kernel (a collective, an optimizer update) holds a few compute units?
Builds tools/probes/liboccupy.so on first use (hipcc), then times the library's GEMMs alone and beside the occupier.
    python tools/interference_probe.py [wgs ...]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from osu_diffusion_amd import _lib  # noqa: E402

so = os.path.join(ROOT, "tools", "probes", "liboccupy.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", "-o", so,
                           os.path.join(ROOT, "tools", "probes", "occupy.hip")])
occ = ctypes.CDLL(so)
occ.occupy.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
L = _lib.lib()
dev = torch.device("cuda:0")
side = torch.cuda.Stream(device=dev)
D, M = 768, 32768
SHAPES = [("fc1 fwd (6 rounds)", _lib.EPI_BIAS_GELU_TE, M, 4 * D, D), ("fc2 fwd (2 rounds)", _lib.EPI_BIAS_TE, M, D, 4 * D),
          ("qkv fwd (6 rounds)", _lib.EPI_BIAS_TE, M, 3 * D, D), ("proj fwd (2 rounds)", _lib.EPI_BIAS_TE, M, D, D)]


def bench(epi, My, Nx, K, wgs, iters=20):
    Y = torch.randn(My, K, device=dev).to(torch.bfloat16)
    X = (torch.randn(Nx, K, device=dev) / K ** 0.5).to(torch.bfloat16)
    out = torch.zeros(My, Nx, dtype=torch.bfloat16, device=dev)
    bias = torch.zeros(max(My, Nx), device=dev)

    def go():
        _lib.check(L.osud_op_gemm(0, epi, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias), None, 0, 0, 0, None))
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    if wgs:
        assert occ.occupy(wgs, 40000, 96, ctypes.c_void_p(side.cuda_stream)) == 0  # 40 ms: outlives the timed region
        import time
        time.sleep(0.005)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    torch.cuda.synchronize()
    return us


def bench_wgrad(Ny, Nx, wgs, iters=20):
    P = (torch.randn(M + 1, Ny, device=dev) * 0.05).to(torch.bfloat16)[:M]
    Q = torch.randn(M + 1, Nx, device=dev).to(torch.bfloat16)[:M]
    out = torch.empty(Ny, Nx, device=dev)
    ws = torch.empty(32 * Ny * Nx, device=dev)

    def go():
        _lib.check(L.osud_op_wgrad(_lib.ptr(P), Ny, _lib.ptr(Q), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None))
    for _ in range(3):
        go()
    torch.cuda.synchronize()
    if wgs:
        assert occ.occupy(wgs, 40000, 96, ctypes.c_void_p(side.cuda_stream)) == 0
        import time
        time.sleep(0.005)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        go()
    e1.record()
    e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    torch.cuda.synchronize()
    return us


if __name__ == "__main__":
    # every shape with the static tile / unit order and with the ticket queues (osud_set_gemm_dynamic_tiles(1): what a data-parallel trainer
    # switches on); both run the phased loops (option gemm_loop = 1) since round 6
    counts = [int(a) for a in sys.argv[1:]] or [0, 8, 16, 32]
    for mode, label in ((0, "static"), (1, "queued")):
        _lib.check(L.osud_set_gemm_dynamic_tiles(mode))
        for name, epi, My, Nx, K in SHAPES:
            row = [f"{w:3d} CUs held: {bench(epi, My, Nx, K, w):7.1f} us" for w in counts]
            print(f"{label} {name:20s} " + "   ".join(row), flush=True)
        for name, Ny, Nx in (("fc1 wgrad", 4 * D, D), ("qkv wgrad", 3 * D, D)):
            row = [f"{w:3d} CUs held: {bench_wgrad(Ny, Nx, w):7.1f} us" for w in counts]
            print(f"{label} {name:20s} " + "   ".join(row), flush=True)
    _lib.check(L.osud_set_gemm_dynamic_tiles(-1))
