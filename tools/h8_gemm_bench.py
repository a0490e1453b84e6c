#!/usr/bin/env python3
"""fp16 + e4m3 operand GEMM (OSUD_PREC_F16F8) next to the split-bf16 one (OSUD_PREC_BF16X3): the device pack kernel against a torch
restatement of the format, accuracy against an fp64 product of the original fp32 operands, and time per launch at the four GEMM
shapes of a DiT-B sampling step (16384 rows).

  python tools/h8_gemm_bench.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib

L = _lib.lib()
dev = torch.device("cuda:0")


def pack_h8_torch(t, weight):
    """(R, C) fp32 (C % 32 == 0) -> (R, 4C) uint8: groups of 32 columns = [64 B fp16 hi | 32 B plane P | 32 B plane Q]."""
    t = t.cpu()
    R, C = t.shape
    hi = t.to(torch.float16)
    lo8 = ((t - hi.float()) * 4096.0).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    hi8 = t.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    P, Q = (hi8, lo8) if weight else (lo8, hi8)
    g = torch.cat([hi.view(torch.uint8).view(R, C // 32, 64), P.view(R, C // 32, 32), Q.view(R, C // 32, 32)], dim=2)
    return g.reshape(R, 4 * C).contiguous()


def pack_h8(t, weight):
    R, C = t.shape
    out = torch.empty(R, 4 * C, dtype=torch.uint8, device=dev)
    _lib.check(L.osud_op_pack_h8(_lib.ptr(t), C, C, _lib.ptr(out), C, R, 1 if weight else 0, None))
    return out


def to_x3(t):
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=1).contiguous()


def timeit(go, iters=30):
    for _ in range(3):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    torch.manual_seed(0)
    a = torch.randn(256, 96, device=dev) * 3
    for w in (False, True):
        d, r = pack_h8(a, w).cpu(), pack_h8_torch(a, w)
        print(f"pack kernel vs torch restatement (weight={w}): {int((d != r).sum())} differing bytes of {d.numel()}")
    for (My, Nx, K) in [(256, 256, 64), (16384, 3072, 768), (16384, 768, 3072), (16384, 2304, 768), (16384, 768, 768)]:
        Y = torch.randn(My, K, device=dev) * 2.0
        X = torch.randn(Nx, K, device=dev) / K ** 0.5
        bias = torch.randn(Nx, device=dev)
        rows = torch.arange(0, My, max(1, My // 256), device=dev)[:256]
        ref = (Y[rows].double() @ X.double().T + bias.double()).float()
        out = torch.zeros(My, Nx, device=dev)
        res = {}
        for name, prec, Yc, Xc in (("x3", _lib.PREC_BF16X3, to_x3(Y), to_x3(X)), ("h8", _lib.PREC_F16F8, pack_h8(Y, False), pack_h8(X, True))):
            go = lambda: _lib.check(L.osud_op_gemm(prec, _lib.EPI_BIAS_F32, _lib.ptr(Yc), K, _lib.ptr(Xc), K, My, Nx, K, _lib.ptr(out), Nx,
                                                   _lib.ptr(bias), None, 0, 0, 0, None))
            out.zero_()
            go()
            torch.cuda.synchronize()
            d = (out[rows] - ref)
            res[name] = (float(d.abs().max()), float(d.pow(2).mean().sqrt()), timeit(go))
        f16 = (Y[rows].half().double() @ X.half().double().T + bias.double()).float() - ref
        print(f"{My}x{Nx}x{K}: x3 max {res['x3'][0]:.2e} rms {res['x3'][1]:.2e} {res['x3'][2]:7.1f} us | h8 max {res['h8'][0]:.2e} rms {res['h8'][1]:.2e} "
              f"{res['h8'][2]:7.1f} us  ({res['x3'][2] / res['h8'][2]:.2f}x) | plain fp16 operands: max {float(f16.abs().max()):.2e}", flush=True)


if __name__ == "__main__":
    main()
