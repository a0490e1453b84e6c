"""Experimental fp8 (OCP e4m3, unit block scales) operand type of the GEMM: correctness against an fp32 product of the same
fp8 values, and speed next to bf16 on the DiT-B shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")

def run(prec, Y, X, My, Nx, K, iters=20):
    out = torch.zeros(My, Nx, dtype=torch.float32, device=dev)
    def go():
        _lib.check(L.osud_op_gemm(prec, _lib.EPI_NONE_F32, _lib.ptr(Y), K, _lib.ptr(X), K, My, Nx, K, _lib.ptr(out), Nx, None, None, 0, 0, 0, None))
    for _ in range(3): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): go()
    e1.record(); torch.cuda.synchronize()
    return out, e0.elapsed_time(e1) * 1e3 / iters

torch.manual_seed(0)
# correctness on a small case
My, Nx, K = 256, 384, 512
Yf = (torch.randn(My, K, device=dev) * 0.5); Xf = (torch.randn(Nx, K, device=dev) * 0.5)
Y8, X8 = Yf.to(torch.float8_e4m3fn), Xf.to(torch.float8_e4m3fn)
got, _ = run(2, Y8, X8, My, Nx, K, iters=1)
ref = Y8.float() @ X8.float().T
print("fp8 gemm max|d| vs fp32 product of the same fp8 values:", float((got - ref).abs().max()), " max|ref|", float(ref.abs().max()))
D = 768
for name, My, Nx, K in (("fc1", 32768, 4 * D, D), ("fc2", 32768, D, 4 * D), ("qkv", 32768, 3 * D, D), ("square 8192", 8192, 8192, 8192)):
    Yf = torch.randn(My, K, device=dev) * 0.5; Xf = torch.randn(Nx, K, device=dev) * 0.5
    _, us8 = run(2, Yf.to(torch.float8_e4m3fn), Xf.to(torch.float8_e4m3fn), My, Nx, K)
    _, us16 = run(0, Yf.to(torch.bfloat16), Xf.to(torch.bfloat16), My, Nx, K)
    fl = 2.0 * My * Nx * K
    print(f"{name:12s} {My}x{Nx}x{K}: fp8 {us8:7.1f} us = {fl / us8 / 1e6:6.0f} TF/s | bf16 {us16:7.1f} us = {fl / us16 / 1e6:6.0f} TF/s (fp32 output)")
