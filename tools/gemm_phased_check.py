#!/usr/bin/env python3
"""The phased GEMM main loop (csrc/gemm_phased.h, option gemm_loop = 1) against the slab loop (gemm_loop = 0) through osud_op_gemm:
bit equality on every shape / epilogue listed, then interleaved timings of the DiT-B training and sampling shapes.
    python tools/gemm_phased_check.py [--no-time] [--rounds N] [--m 32768]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--no-time", action="store_true")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--m", type=int, default=32768)
ap.add_argument("--prec", type=int, default=0, help="0 bf16, 5 fp16 (OSUD_PREC_F16)")
args = ap.parse_args()
L = _lib.lib(); dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(11)
EPI = {"bias_f32": _lib.EPI_BIAS_F32, "bias": _lib.EPI_BIAS_TE, "silu": _lib.EPI_BIAS_SILU_TE, "rowbias": _lib.EPI_ROWBIAS_TE,
       "gelu": _lib.EPI_BIAS_GELU_TE, "gate": _lib.EPI_GATE_RES, "none_f32": _lib.EPI_NONE_F32, "none": _lib.EPI_NONE_TE,
       "accum": _lib.EPI_ACCUM_F32}
F32OUT = {"bias_f32", "gate", "none_f32", "accum"}
tdt = torch.bfloat16 if args.prec == 0 else torch.float16


def operands(M, N, K):
    Y = torch.randn(M, K, device=dev, generator=g).to(tdt)
    X = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(tdt)
    return Y, X


def run(epi, Y, X, out, bias, gate, rps, ns):
    M, K = Y.shape; N = X.shape[0]
    _lib.check(L.osud_op_gemm(args.prec, EPI[epi], _lib.ptr(Y), K, _lib.ptr(X), K, M, N, K, _lib.ptr(out), N, _lib.ptr(bias),
                              _lib.ptr(gate) if gate is not None else None, N if gate is not None else 0, rps, ns, None))


def both(epi, M, N, K):
    Y, X = operands(M, N, K)
    bias = torch.randn(max(M, N), device=dev, generator=g)
    gate, rps, ns = None, 0, 0
    if epi == "gate":
        rps, ns = 128, M // 128
        gate = torch.randn(ns, N, device=dev, generator=g)
    outs = []
    for loop in (0, 1):
        _lib.set_option("gemm_loop", loop)
        init = torch.randn(M, N, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
        out = init.clone() if epi in F32OUT else init.to(tdt)
        run(epi, Y, X, out, bias, gate, rps, ns)
        torch.cuda.synchronize()
        outs.append(out)
    same = bool(torch.equal(outs[0], outs[1]))
    ref = Y.float() @ X.float().t()
    sane = True
    if epi in ("none", "none_f32"):
        err = float((outs[1].float() - ref).abs().max()); sane = err < 0.1 * max(1.0, float(ref.abs().max()))
    return same, sane


CASES = []
for epi in EPI:
    CASES += [(epi, 512, 512, 768), (epi, 512, 768, 768)]          # 256x256 / 256x192 tiles, 12 slabs
CASES += [("bias", 256, 256, 128), ("bias", 256, 768, 128),          # two slabs: the shortest stream
          ("bias", 256, 512, 192), ("bias", 512, 384, 192),          # odd slab counts: the buffer parity alternates between tiles
          ("none", 1024, 256, 320), ("none", 768, 384, 448),
          ("gelu", 2048, 3072, 768), ("bias", 2048, 768, 3072), ("bias", 2048, 2304, 768),
          ("none", 16384, 768, 768), ("none", 16384, 3072, 768), ("bias", 16384, 2304, 768),
          ("gelu", 32768, 3072, 768), ("none", 32768, 768, 3072), ("gate", 32768, 768, 768)]
bad = 0
for epi, M, N, K in CASES:
    for tile in (256, 192):  # force each 256-row geometry where it divides (small shapes would otherwise get the 128-row tiles)
        if N % tile:
            continue
        _lib.set_option("gemm_tile", tile)
        for rep in range(2):  # twice: a race would not repeat itself
            same, sane = both(epi, M, N, K)
            if not (same and sane):
                bad += 1
            print(f"{epi:9s} {M:6d} x {N:5d} x {K:5d} tile 256x{tile}  phased == slab: {same}  sane: {sane}", flush=True)
_lib.set_option("gemm_tile", 0)
print("MISMATCHES:", bad, flush=True)
if args.no_time:
    sys.exit(1 if bad else 0)

D, M = 768, args.m
SHAPES = [("qkv fwd", "bias", M, 3 * D, D), ("proj fwd", "bias", M, D, D), ("fc1 fwd", "gelu", M, 4 * D, D), ("fc2 fwd", "bias", M, D, 4 * D),
          ("fc2 dgrad", "none", M, 4 * D, D), ("fc1 dgrad", "none", M, D, 4 * D), ("qkv dgrad", "none", M, D, 3 * D), ("proj dgrad", "none", M, D, D)]


def bench(epi, Mm, N, K, iters=20):
    Y, X = operands(Mm, N, K)
    out = torch.zeros(Mm, N, dtype=torch.float32 if epi in F32OUT else tdt, device=dev)
    bias = torch.randn(max(Mm, N), device=dev) * 0.02
    res = {0: [], 1: []}
    for rnd in range(args.rounds):
        for loop in (0, 1):
            _lib.set_option("gemm_loop", loop)
            for _ in range(3):
                run(epi, Y, X, out, bias, None, 0, 0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(iters):
                run(epi, Y, X, out, bias, None, 0, 0)
            e1.record(); torch.cuda.synchronize()
            res[loop].append(e0.elapsed_time(e1) * 1e3 / iters)
    return res


for name, epi, Mm, N, K in SHAPES:
    r = bench(epi, Mm, N, K)
    med = {k: sorted(v)[len(v) // 2] for k, v in r.items()}
    tf = {k: 2.0 * Mm * N * K / med[k] / 1e6 for k in med}
    print(f"{name:11s} {Mm}x{N}x{K}  slab {med[0]:6.1f} us {tf[0]:5.0f} TF (min {min(r[0]):6.1f})   phased {med[1]:6.1f} us {tf[1]:5.0f} TF (min {min(r[1]):6.1f})"
          f"   phased/slab {med[1] / med[0]:.3f}", flush=True)
sys.exit(1 if bad else 0)
