#!/usr/bin/env python3
"""GPU bring-up probe: prints measured errors of each native op against torch / the oracle.
Not a test (tests/ hold the assertions); used to size tolerances and debug on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from osu_diffusion_amd import _lib  # noqa: E402
from osu_diffusion_amd.models import DiT  # noqa: E402
from osu_diffusion_amd.diffusion import create_diffusion  # noqa: E402
from oracle import dit_oracle as mo, diffusion_oracle as do  # noqa: E402
from tests.helpers import load, T, weights_for  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
print("lib", _lib.LIB_PATH, "arch", L.osud_build_arch().decode(), "device", torch.cuda.get_device_name(0), flush=True)


def conv(prec, t):
    out = torch.empty(t.numel() * (2 if prec == 0 else 4), dtype=torch.uint8, device=dev)
    _lib.check(L.osud_op_convert(prec, _lib.ptr(t.contiguous()), _lib.ptr(out), t.numel(), None))
    return out


def back(prec, buf, shape):
    if prec == 0:
        return buf.view(torch.bfloat16).view(shape).float()
    return buf.view(torch.float32).view(shape).clone()


def probe_gemm():
    torch.manual_seed(0)
    for prec in (0, 1):
        for (My, Nx, K) in [(128, 128, 64), (256, 384, 576), (384, 256, 768), (128, 3072, 768), (256, 768, 3072)]:
            Y = torch.randn(My, K, device=dev)
            X = torch.randn(Nx, K, device=dev) / K ** 0.5
            bias = torch.randn(Nx, device=dev)
            Yc, Xc = conv(prec, Y), conv(prec, X)
            Yr, Xr = back(prec, Yc, (My, K)), back(prec, Xc, (Nx, K))
            ref = (Yr.double() @ Xr.double().T + bias.double()).float()
            out = torch.zeros(My, Nx, device=dev)
            _lib.check(L.osud_op_gemm(prec, _lib.EPI_BIAS_F32, _lib.ptr(Yc), K, _lib.ptr(Xc), K, My, Nx, K, _lib.ptr(out),
                                      Nx, _lib.ptr(bias), None, 0, 0, 0, None))
            torch.cuda.synchronize()
            print(f"gemm prec={prec} {My}x{Nx}x{K}: max|d|={float((out - ref).abs().max()):.3e} (|ref|max {float(ref.abs().max()):.2f})",
                  flush=True)


def probe_forward():
    for tag in ["tiny_T64", "tiny_T128", "tiny_T200_band", "tiny_T128_allfalse", "small_T128", "tiny_T128_rough"]:
        fx = load("g3_forward_" + tag)
        shape, sd = weights_for(fx)
        for precision in ("fp32", "bf16"):
            m = DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads, context_size=144,
                    num_classes=shape.num_classes, class_dropout_prob=0.2, precision=precision)
            m.load_state_dict(sd)
            m = m.to(dev).eval()
            mask = T(fx["attn_mask"]).to(dev) if "attn_mask" in fx else None
            args = [T(fx[k]).to(dev) for k in ("x", "t", "o", "c", "y")]
            with torch.no_grad():
                out = m(*args, attn_mask=mask)
                cfg4 = m.forward_with_cfg(*args, 4.0, attn_mask=mask)
            torch.cuda.synchronize()
            e1 = float((out.cpu() - T(fx["out"])).abs().max())
            e2 = float((cfg4.cpu() - T(fx["out_cfg4"])).abs().max())
            print(f"forward {tag} {precision}: max|d| out={e1:.3e} cfg4={e2:.3e} (|ref|max {float(np.abs(fx['out']).max()):.3f})",
                  flush=True)


def probe_steps():
    for tag in ("1000", "250"):
        fx = load("g5_step_" + tag)
        d = create_diffusion(tag, noise_schedule="squaredcos_cap_v2")
        x, t, mout = (T(fx[k]).to(dev) for k in ("x", "t", "model_out"))
        N, _, TT = x.shape
        for mode, eta, key in [(0, 0.0, "p"), (1, 0.0, "ddim0"), (1, 1.0, "ddim1")]:
            out, x0 = torch.empty_like(x), torch.empty_like(x)
            nz = T(fx[key + "_noise"]).to(dev)
            _lib.check(L.osud_sampler_step(d._sched.handle, mode, eta, _lib.ptr(mout), _lib.ptr(x), _lib.ptr(t), _lib.ptr(nz),
                                           N, TT, -1.0, 1, _lib.ptr(out), _lib.ptr(x0), None))
            torch.cuda.synchronize()
            print(f"step {tag} {key}: sample max|d|={float((out.cpu() - T(fx[key + '_sample'])).abs().max()):.3e} "
                  f"x0 max|d|={float((x0.cpu() - T(fx[key + '_x0'])).abs().max()):.3e}", flush=True)


def probe_loop():
    for tag in ("p20", "ddim20_eta1", "ddim20_eta05"):
        fx = load("g6_loop_" + tag)
        shape, sd = weights_for(fx)
        for precision in ("fp32", "bf16"):
            m = DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads, context_size=144,
                    num_classes=shape.num_classes, class_dropout_prob=0.2, precision=precision)
            m.load_state_dict(sd)
            m = m.to(dev).eval()
            d = create_diffusion(str(fx["respacing"]), noise_schedule="squaredcos_cap_v2")
            z = T(fx["z"]).to(dev)
            kw = dict(o=T(fx["o"]).to(dev), c=T(fx["c"]).to(dev), y=T(fx["y"]).to(dev), cfg_scale=4.0, attn_mask=None)
            eta = float(fx["eta"])
            for graph in ("1", "0"):
                os.environ["OSUD_NO_GRAPH"] = "0" if graph == "1" else "1"
                if eta < 0:
                    fin = d.p_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=T(fx["noises"]))
                else:
                    fin = d.ddim_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, eta=eta,
                                             step_noise=T(fx["noises"]))
                torch.cuda.synchronize()
                print(f"loop {tag} {precision} graph={graph}: final max|d|={float((fin.cpu() - T(fx['final'])).abs().max()):.3e}",
                      flush=True)
    os.environ["OSUD_NO_GRAPH"] = "0"


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "forward", "steps", "loop"]
    for w in which:
        t0 = time.time()
        globals()["probe_" + w]()
        print(f"[{w}] done in {time.time() - t0:.1f}s", flush=True)
