#!/usr/bin/env python3
"""Time the HBM-bound row-wise kernels of a DiT-B training step in isolation (HIP events, M = 32768 rows, D = 768): LayerNorm+modulate
forward and backward, the gate step, AdamW+EMA -- next to a plain device copy of the same byte count (what this chip's memory system
gives a streaming kernel).  The launchers are C++ symbols of libosud.so (not ABI): this is a probe, looked up by mangled name.

  python tools/rowwise_bench.py            # OSUD_LIB=ab/libosud_x.so selects a variant build
"""
import ctypes, os, subprocess, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")
path = _lib.LIB_PATH
syms = {}
for line in subprocess.run(["nm", "-D", path], capture_output=True, text=True).stdout.splitlines():
    parts = line.split()
    if len(parts) == 3 and parts[1] == "T" and parts[2].startswith("_ZN4osud"):
        name = parts[2][8:]
        n = int("".join(c for c in name[:3] if c.isdigit()))
        syms[name[len(str(n)):len(str(n)) + n]] = parts[2]
raw = ctypes.CDLL(path)


def fn(name):
    f = getattr(raw, syms[name])
    f.restype = ctypes.c_int
    return f


P = lambda t: ctypes.c_void_p(0 if t is None else t.data_ptr())
I = ctypes.c_int


def timeit(name, go, nbytes, iters=40):
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:46s} {us:8.1f} us   {nbytes / 1e6:7.1f} MB   {nbytes / us / 1e6:5.2f} TB/s", flush=True)
    return us


def main():
    M, D, T = 32768, int(os.environ.get("D", "768")), 128
    N = M // T
    AC = 6 * D * 12 + 2 * D
    bf = torch.bfloat16
    h = torch.randn(M, D, device=dev); stats = torch.stack([h.mean(1), h.var(1, unbiased=False).add(1e-6).rsqrt()], 1).contiguous()
    du = torch.randn(M, D, device=dev).to(bf); ada = torch.randn(N, AC, device=dev) * 0.1
    dh = torch.randn(M, D, device=dev); dh2 = torch.empty_like(dh); dada = torch.zeros(N, AC, device=dev)
    br = torch.randn(M, D, device=dev).to(bf); dbr = torch.empty_like(br); db = torch.zeros(D, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    lnb = fn("launch_ln_mod_bwd")

    part = torch.empty(M // 64, 6 * D + 64, device=dev)  # per-workgroup partial rows (summed by row_reduce_kernel, not timed here)

    def ln_bwd(with_gate):
        lnb(I(0), P(h), P(stats), P(du), P(ada), I(AC), I(0), I(D), P(dh), P(dh2), P(part), I(M), I(T), I(D), st,
            P(br if with_gate else None), I(2 * D), P(dbr if with_gate else None),
            P(None), P(None), P(None))  # (no e4m3 twin)
    row = D * (4 + 2 + 4 + 4)
    timeit("ln_mod_bwd (no gate step)", lambda: ln_bwd(False), M * row)
    timeit("ln_mod_bwd + gate step of the next branch", lambda: ln_bwd(True), M * (row + D * 4))

    gb = fn("launch_gate_bwd")
    timeit("gate_bwd", lambda: gb(I(0), P(dh), P(br), P(ada), I(AC), P(dbr), P(part), I(M), I(T), I(D), st), M * D * (4 + 2 + 2))

    u = torch.empty(M, D, device=dev, dtype=bf); stats_o = torch.empty(M, 2, device=dev)
    ln = fn("launch_ln_mod")
    h_out = torch.empty_like(h)

    def ln_fwd(with_res):  # launch_ln_mod(prec, h, ada, ld_ada, off_shift, off_scale, out, stats, M, Tp, N, D, st, br, off_gate, h_out, fp8_scale)
        ln(I(0), P(h), P(ada), I(AC), I(0), I(D), P(u), P(stats_o), I(M), I(T), I(N), I(D), st, P(br if with_res else None), I(2 * D),
           P(h_out if with_res else None), ctypes.c_float(0.0))
    timeit("ln_mod forward", lambda: ln_fwd(False), M * D * (4 + 2))
    timeit("ln_mod forward + gated residual in front", lambda: ln_fwd(True), M * D * (4 + 2 + 2 + 4))

    # AdamW + EMA over a DiT-B-sized arena
    n = 130_000_000
    p = torch.randn(n, device=dev) * 0.02; g = torch.randn(n, device=dev) * 1e-3
    m1 = torch.zeros(n, device=dev); m2 = torch.zeros(n, device=dev); ema = p.clone()
    step = [0]

    def adam():
        step[0] += 1
        _lib.check(L.osud_adamw_ema_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m1), _lib.ptr(m2), _lib.ptr(ema), n, 1e-4, 0.9, 0.999, 1e-8, 0.0,
                                         step[0], 0.9999, 0, 0, 1.0, None))
    timeit("adamw_ema 130 M parameters", adam, n * 36, iters=10)

    # yardsticks: plain copies
    a = torch.empty(M * row // 8, device=dev); b = torch.empty_like(a)
    timeit("torch copy_ (same bytes as ln_mod_bwd)", lambda: b.copy_(a), a.numel() * 8)
    a = torch.empty(n * 36 // 8, device=dev); b = torch.empty_like(a)
    timeit("torch copy_ (same bytes as adamw_ema)", lambda: b.copy_(a), a.numel() * 8, iters=10)


if __name__ == "__main__":
    main()
