#!/usr/bin/env python3
"""The phased weight-gradient kernel (csrc/wgrad.hip: wgrad_phased_kernel, option gemm_loop = 1) against wgrad_kernel (gemm_loop = 0)
through osud_op_wgrad: bit equality of the combined result on DiT-B / small / ragged-split shapes, then interleaved timings."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(7)


def make(Ny, Nx, M):
    P = torch.randn(M, Ny, device=dev, generator=g).to(torch.bfloat16); Q = torch.randn(M, Nx, device=dev, generator=g).to(torch.bfloat16)
    tiles = ((Ny + 255) // 256) * ((Nx + 255) // 256)
    S = max(1, min(32, 256 // tiles))
    ws = torch.empty(S * Ny * Nx, device=dev); out = torch.empty(Ny, Nx, device=dev)
    return P, Q, ws, out


def run(P, Q, ws, out):
    M, Ny = P.shape; Nx = Q.shape[1]
    _lib.check(L.osud_op_wgrad(_lib.ptr(P), Ny, _lib.ptr(Q), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None))


bad = 0
for Ny, Nx, M in [(256, 256, 128), (256, 256, 192), (512, 256, 1024), (768, 768, 4096), (768, 3072, 8192), (2304, 768, 32768), (768, 768, 32768),
                  (3072, 768, 32768), (768, 3072, 32768), (1152, 1152, 8192), (384, 256, 2048)]:
    for rep in range(2):
        P, Q, ws, out = make(Ny, Nx, M)
        res = []
        for loop in (0, 1):
            _lib.set_option("gemm_loop", loop)
            out.fill_(float("nan")); ws.fill_(float("nan"))
            run(P, Q, ws, out); torch.cuda.synchronize()
            res.append(out.clone())
        same = bool(torch.equal(res[0], res[1]))
        ref = P.float().t() @ Q.float()
        err = float((res[1] - ref).abs().max()) / max(1.0, float(ref.abs().max()))
        bad += (not same) or not (err < 1e-2)
        print(f"wgrad {Ny:5d} x {Nx:5d} over {M:6d} tokens  phased == slab: {same}  rel err vs fp32 {err:.2e}", flush=True)
print("MISMATCHES:", bad, flush=True)
for name, Ny, Nx in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    P, Q, ws, out = make(Ny, Nx, 32768)
    r = {0: [], 1: []}
    for rnd in range(5):
        for loop in (0, 1):
            _lib.set_option("gemm_loop", loop)
            for _ in range(3): run(P, Q, ws, out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): run(P, Q, ws, out)
            e1.record(); torch.cuda.synchronize(); r[loop].append(e0.elapsed_time(e1) * 1e3 / 20)
    m = {k: sorted(v)[2] for k, v in r.items()}
    print(f"{name:5s} wgrad + combine: slab {m[0]:6.1f} us {2.0 * Ny * Nx * 32768 / m[0] / 1e6:5.0f} TF   phased {m[1]:6.1f} us {2.0 * Ny * Nx * 32768 / m[1] / 1e6:5.0f} TF   ratio {m[1] / m[0]:.3f}", flush=True)
sys.exit(1 if bad else 0)
