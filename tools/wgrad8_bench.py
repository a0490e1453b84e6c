"""Weight-gradient product on e4m3 operands (osud_op_wgrad8) next to the bf16 kernel (osud_op_wgrad) at the shapes of a DiT-XL
(T = 256, 128 windows: M = 32768) and a DiT-B training step; HIP events, random operands."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from osu_diffusion_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0")
M = 32768
def timed(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, Ny, Nx in (("XL qkv", 3456, 1152), ("XL proj", 1152, 1152), ("XL fc1", 4608, 1152), ("XL fc2", 1152, 4608),
                     ("B qkv", 2304, 768), ("B fc1", 3072, 768), ("B fc2", 768, 3072)):
    # (one extra row: the 256-wide edge tiles of widths like 1152 / 3456 stage up to 256 bytes past the last row's end -- the library's
    #  own buffers carry that slack, csrc/dit.h dev_alloc)
    P = (torch.randn(M + 1, Ny, device=dev) * 0.02)[:M]; Q = torch.randn(M + 1, Nx, device=dev)[:M]
    pad = lambda t, dt: torch.cat([t, t[:1]]).to(dt)[:M]  # noqa: E731  (same slack behind the converted copies)
    Pb, Qb = pad(P, torch.bfloat16), pad(Q, torch.bfloat16)
    P8, Q8 = pad(P * 4000, torch.float8_e4m3fn), pad(Q * 50, torch.float8_e4m3fn)
    inv = torch.tensor([1.0], device=dev)
    out = torch.empty(Ny, Nx, device=dev); ws = torch.empty(32 * Ny * Nx, device=dev)
    t16 = timed(lambda: _lib.check(L.osud_op_wgrad(_lib.ptr(Pb), Ny, _lib.ptr(Qb), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), None)))
    t8 = timed(lambda: _lib.check(L.osud_op_wgrad8(_lib.ptr(P8), Ny, _lib.ptr(Q8), Nx, Ny, Nx, M, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.ptr(inv), _lib.ptr(inv), None)))
    gf = 2.0 * M * Ny * Nx / 1e9
    print(f"{name:8s} {Ny:5d} x {Nx:5d}: bf16 {t16:7.1f} us ({gf / t16 * 1e-3:6.0f} TF/s)   e4m3 {t8:7.1f} us ({gf / t8 * 1e-3:6.0f} TF/s)   x{t16 / t8:.2f}")
