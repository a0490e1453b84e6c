"""GPU (-m gpu): the split-bf16 tier (precision="bf16x3", OSUD_PREC_BF16X3) through the C ABI.

Every GEMM / attention operand travels as hi = bf16(v), lo = bf16(v - hi) (rows [hi plane | lo plane]); a product is the three
bf16 MFMAs hi*hi + lo*hi + hi*lo with fp32 accumulation: 16 significand bits per operand -- finer than the TF32 matmuls of the
reference's own sampling path (sample.py:25-26) -- at a third of the bf16 tier's MFMA rate.  This is the tier that meets the
north star's 1e-3 tolerance on final coordinates at MFMA speed; the exact-f32 tier (tests/test_gpu_forward.py) stays the anchor.

Tolerances: operators against an fp64 evaluation of the ORIGINAL fp32 operands (so the hi/lo rounding is inside the bound);
model outputs and loops against the fixtures frozen from the reference; each bound is <= 3x what was measured on MI355X
(measured values in the comments, printed by the tests).
"""
import numpy as np
import pytest
import torch

from osu_diffusion_amd import _lib
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.synthetic import banded_attn_mask
from tests.helpers import T, load, maxdiff, weights_for
from tests.test_gpu_forward import FWD_TAGS, native_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
X3 = _lib.PREC_BF16X3


def to_x3(t):
    """(R, C) fp32 -> (R, 2C) bf16 plane pair [hi | lo]."""
    hi = t.to(torch.bfloat16)
    lo = (t - hi.float()).to(torch.bfloat16)
    return torch.cat([hi, lo], dim=1).contiguous()


def from_x3(buf, rows, cols):
    p = buf.view(torch.bfloat16).view(rows, 2 * cols).float()
    return p[:, :cols] + p[:, cols:]


@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 384, 576), (128, 3072, 768), (256, 768, 3072), (16384, 768, 768)])
def test_gemm_split_bf16_is_an_fp32_class_product(shape):
    My, Nx, K = shape
    torch.manual_seed(My + Nx + K)
    Y = torch.randn(My, K, device=DEV)
    X = torch.randn(Nx, K, device=DEV) / K ** 0.5  # asymmetric operands: a transposed result cannot pass
    bias = torch.randn(Nx, device=DEV)
    ref = (Y.double() @ X.double().T + bias.double()).float()
    Yc, Xc = to_x3(Y), to_x3(X)
    out = torch.zeros(My, Nx, device=DEV)
    _lib.check(_lib.lib().osud_op_gemm(X3, _lib.EPI_BIAS_F32, _lib.ptr(Yc), K, _lib.ptr(Xc), K, My, Nx, K,
                                       _lib.ptr(out), Nx, _lib.ptr(bias), None, 0, 0, 0, None))
    err = maxdiff(out.cpu(), ref.cpu())
    bf = maxdiff((Y.to(torch.bfloat16).double() @ X.to(torch.bfloat16).double().T + bias.double()).float().cpu(), ref.cpu())
    print(f"split-bf16 gemm {shape}: max|d| = {err:.3e} (plain bf16 operands: {bf:.3e})")
    assert err < 6e-5 and err < bf / 100  # measured 2.0-2.5e-5; plain bf16 operands 1.0-1.3e-2


def test_gemm_split_bf16_fused_epilogues():
    My, Nx, K, Tp, NS = 256, 256, 128, 64, 3
    torch.manual_seed(5)
    Y = torch.randn(My, K, device=DEV)
    X = torch.randn(Nx, K, device=DEV) / K ** 0.5
    bias = torch.randn(Nx, device=DEV)
    gate = torch.randn(NS, Nx, device=DEV)
    acc = Y.double() @ X.double().T
    Yc, Xc = to_x3(Y), to_x3(X)
    L = _lib.lib()

    def run(epi, out, g=None):
        _lib.check(L.osud_op_gemm(X3, epi, _lib.ptr(Yc), K, _lib.ptr(Xc), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias),
                                  _lib.ptr(g), Nx if g is not None else 0, Tp, NS, None))
        torch.cuda.synchronize()

    o = torch.zeros(My, 2 * Nx, dtype=torch.bfloat16, device=DEV)  # plane-pair output
    errs = {}
    run(_lib.EPI_BIAS_TE, o)
    errs["bias"] = maxdiff(from_x3(o, My, Nx).cpu(), (acc + bias.double()).cpu())
    run(_lib.EPI_BIAS_SILU_TE, o)
    errs["silu"] = maxdiff(from_x3(o, My, Nx).cpu(), torch.nn.functional.silu(acc + bias.double()).cpu())
    run(_lib.EPI_BIAS_GELU_TE, o)
    errs["gelu"] = maxdiff(from_x3(o, My, Nx).cpu(), torch.nn.functional.gelu(acc + bias.double(), approximate="tanh").cpu())
    res = torch.randn(My, Nx, device=DEV)
    want = res.double() + gate.double()[torch.clamp(torch.arange(My, device=DEV) // Tp, max=NS - 1)] * (acc + bias.double())
    run(_lib.EPI_GATE_RES, res, gate)
    errs["gate_res"] = maxdiff(res.cpu(), want.cpu())
    print("split-bf16 epilogues:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < 1.5e-4, errs  # measured 3.3-5.6e-5 (outputs up to |5|: 2^-17 relative of the stored pair)


def test_gemm_split_bf16_rejects_what_it_does_not_build():
    a = torch.zeros(128, 2 * 96, dtype=torch.bfloat16, device=DEV)
    o = torch.zeros(128, 128, device=DEV)
    rc = _lib.lib().osud_op_gemm(X3, _lib.EPI_BIAS_F32, _lib.ptr(a), 96, _lib.ptr(a), 96, 128, 128, 96, _lib.ptr(o), 128, _lib.ptr(o),
                                 None, 0, 0, 0, None)
    assert rc == _lib.ERR_ARG and "split-bf16" in _lib.last_error()
    a = torch.zeros(128, 2 * 64, dtype=torch.bfloat16, device=DEV)
    rc = _lib.lib().osud_op_gemm(X3, _lib.EPI_GELUGRAD_TE, _lib.ptr(a), 64, _lib.ptr(a), 64, 128, 128, 64, _lib.ptr(o), 128, _lib.ptr(o),
                                 None, 0, 0, 0, None)
    assert rc in (_lib.ERR_UNSUPPORTED, _lib.ERR_ARG)


@pytest.mark.parametrize("T_,masked,N,H,hd", [(64, False, 2, 2, 64), (128, False, 2, 2, 64), (200, True, 2, 2, 64), (77, False, 2, 2, 64),
                                               (128, False, 41, 13, 64), (128, True, 3, 2, 64), (256, False, 2, 3, 72), (200, True, 2, 2, 72)])
def test_attention_core_split_bf16(T_, masked, N, H, hd):
    D = H * hd
    Tp = (T_ + 63) // 64 * 64
    Mp = (N * Tp + 127) // 128 * 128
    torch.manual_seed(T_)
    qkv = torch.randn(Mp, 3 * D, device=DEV)  # packed in_proj output: Q | K | V
    mask = banded_attn_mask(T_, 128).to(DEV) if masked else None
    qkc = to_x3(qkv)
    out = torch.zeros(Mp, 2 * D, dtype=torch.bfloat16, device=DEV)
    m8 = None if mask is None else mask.to(torch.uint8).contiguous()
    _lib.check(_lib.lib().osud_op_attention(X3, _lib.ptr(qkc), 3 * D, _lib.ptr(m8), _lib.ptr(out), N, T_, Tp, Mp, H, hd, None))
    got = from_x3(out, Mp, D)
    worst = 0.0
    for n in range(N):
        rows = slice(n * Tp, n * Tp + T_)
        q = qkv[rows, :D].reshape(T_, H, hd).transpose(0, 1).double()
        k = qkv[rows, D:2 * D].reshape(T_, H, hd).transpose(0, 1).double()
        vv = qkv[rows, 2 * D:].reshape(T_, H, hd).transpose(0, 1).double()
        s = q @ k.transpose(-1, -2) / hd ** 0.5
        if mask is not None:
            s = s.masked_fill(mask, float("-inf"))
        ref = (torch.softmax(s, -1) @ vv).transpose(0, 1).reshape(T_, D)
        worst = max(worst, maxdiff(got[rows].cpu(), ref.cpu()))
    print(f"split-bf16 attention T={T_} masked={masked} hd={hd}: max|d| = {worst:.3e}")
    assert worst < 7e-5  # measured 0.6-2.3e-5 (the bf16 tier's bound: 2e-2)


@pytest.mark.parametrize("tag", FWD_TAGS)
def test_forward_matches_reference_golden_split_bf16(tag):
    fx = load(f"g3_forward_{tag}")
    shape, sd = weights_for(fx)
    m = native_model(shape, sd, "bf16x3")
    x, t, o, c, y = (T(fx[k]).to(DEV) for k in ("x", "t", "o", "c", "y"))
    mask = T(fx["attn_mask"]).to(DEV) if "attn_mask" in fx else None
    scale = float(np.abs(fx["out"]).max())
    with torch.no_grad():
        errs = {"out": maxdiff(m(x, t, o, c, y, attn_mask=mask).cpu(), fx["out"]),
                "cfg4": maxdiff(m.forward_with_cfg(x, t, o, c, y, 4.0, attn_mask=mask).cpu(), fx["out_cfg4"]),
                "cfg1": maxdiff(m.forward_with_cfg(x, t, o, c, y, 1.0, attn_mask=mask).cpu(), fx["out_cfg1"])}
    print(f"split-bf16 forward {tag}: scale {scale:.2f}, errors {({k: f'{v:.2e}' for k, v in errs.items()})}")
    # measured / max(scale, 1): plain forward 1.3-2.6e-5, guided (cfg 4 amplifies cond - uncond differences) 0.5-1.2e-4
    assert max(errs["out"], errs["cfg1"]) <= 7e-5 * max(scale, 1.0) and errs["cfg4"] <= 3.5e-4 * max(scale, 1.0), errs


@pytest.mark.parametrize("tag", ["p20", "ddim20_eta1", "p250"])
def test_chained_loop_final_coordinates_split_bf16(tag):
    """North star: final (x, y) within 1e-3 of the reference for identical (seed, window, steps) -- met by the MFMA-speed tier."""
    fx = load(f"g6_loop_{tag}")
    shape, sd = weights_for(fx)
    m = native_model(shape, sd, "bf16x3")
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    d = create_diffusion(str(fx["respacing"]), noise_schedule="squaredcos_cap_v2")
    z = T(fx["z"]).to(DEV)
    eta = float(fx["eta"])
    if eta >= 0:
        got = d.ddim_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, eta=eta, step_noise=T(fx["noises"]))
    else:
        got = d.p_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=T(fx["noises"]))
    err = maxdiff(got.cpu(), fx["final"])
    print(f"split-bf16 loop {tag}: final max|d| = {err:.3e}")
    assert err < 2e-4  # measured 1.8e-5 (p20), 6.2e-5 (ddim20), 2.2e-5 (p250); the north star's bound is 1e-3


# ------------------------------------------------------------------ BASELINE configs[3] end to end against the reference
def _p1000_inputs():
    fx = load("g6_loop_p1000_dit_b")
    shape, sd = weights_for(fx)
    z = T(fx["z"])
    # the per-step noise is not stored (4 MB): redrawn from the reference run's seed in its order (one randn_like per step,
    # gaussian_diffusion.py:454) and pinned by the fixture's checksum and first / last values
    torch.manual_seed(int(fx["noise_seed"]))
    noises = torch.stack([torch.randn_like(z) for _ in range(1000)])
    assert abs(float(noises.double().sum()) - float(fx["noise_sum"])) < 1e-6 and torch.equal(noises[0, 0, 0, :8], T(fx["noise_head"]))
    assert torch.equal(noises[-1, -1, -1, -8:], T(fx["noise_tail"]))
    return fx, shape, sd, z, noises


@pytest.mark.parametrize("precision,bound", [("fp32", 1e-3), ("bf16x3", 1e-3), ("bf16", None)])
def test_dit_b_1000_step_cfg4_loop_matches_the_reference(precision, bound):
    """sample.py's headline shape end to end (sample.py:174-182, gaussian_diffusion.py:469-561): DiT-B, 12 blocks, "1000" steps,
    cfg 4.0, N = 4 rows, T = 128, the reference's own p_sample_loop on CPU with recorded seeds (fixture g6_loop_p1000_dit_b).  The
    state after 250 / 500 / 750 executed steps and the final coordinates must be within 1e-3 of the reference in the exact-f32 tier
    AND in the split-bf16 tier (the reference's own fp32 result is 3.5e-4 from an fp64 evaluation of the same loop: fixture field
    final_fp64); the bf16 tier is measured and bounded at 3x what MI355X showed."""
    fx, shape, sd, z, noises = _p1000_inputs()
    m = native_model(shape, sd, precision)
    d = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    x = z.to(DEV).clone()
    nz = noises.to(DEV)
    errs, done = {}, 0
    for k in (250, 500, 750, 1000):  # steps 999 - done .. 1000 - k
        d.run_steps(m.forward_with_cfg, x, kw, first_step=999 - done, last_step=1000 - k, step_noise=nz[done:k])
        errs[k] = maxdiff(x.cpu(), fx["final"] if k == 1000 else fx[f"after_{k}"])
        done = k
    ref64 = maxdiff(fx["final"], fx["final_fp64"])
    print(f"MEASURED p1000_dit_b[{precision}]: max|d| vs reference after 250/500/750/1000 steps = "
          + " / ".join(f"{errs[k]:.3e}" for k in (250, 500, 750, 1000)) + f" (reference fp32 vs fp64 evaluation: {ref64:.3e})")
    if bound is not None:
        assert max(errs.values()) < bound, errs
    else:
        assert max(errs.values()) < BF16_P1000_BOUND, errs


BF16_P1000_BOUND = 2.6e-2  # bf16 tier, 1000 steps: 3x the measured 8.75e-3 (fp32 tier 9.8e-5, split-bf16 tier 1.18e-4, round 3)


@pytest.mark.parametrize("prec,N,H", [("bf16x3", 41, 13), ("bf16x3", 128, 12), ("fp16f8", 7, 3), ("fp16w8", 6, 12)])
def test_streamed_window_kernel_equals_the_general_kernel_bit_for_bit(osud_option, prec, N, H):
    """T = Tp = 128, head_dim 64, no mask -- the shape of window sampling -- runs `attn_fwd_stream_x3_kernel` (pairs of heads, 64-key stages
    by LDS-DMA into a double buffer) instead of the general kernel; the arithmetic is the same instruction for instruction, so the
    outputs are equal BIT FOR BIT in all three output forms (plane pairs, fp16 + e4m3 rows, fp16 + e4m3(v) rows), odd head counts (a
    half-empty last pair) and more pairs than compute units included."""
    hd, T_ = 64, 128
    D, Mp = H * hd, N * T_
    torch.manual_seed(N * 100 + H)
    qkc = to_x3(torch.randn(Mp, 3 * D, device=DEV) * 1.7)
    code = {"bf16x3": _lib.PREC_BF16X3, "fp16f8": _lib.PREC_F16F8, "fp16w8": _lib.PREC_F16W8}[prec]
    outs = []
    for general in (1, 0):
        osud_option("attn_fwd_kernel", general)
        out = torch.zeros(Mp, 4 * D, dtype=torch.uint8, device=DEV)  # (4 bytes per element in the plane-pair and fp16 + e4m3 forms, 3 in fp16w8)
        _lib.check(_lib.lib().osud_op_attention(code, _lib.ptr(qkc), 3 * D, None, _lib.ptr(out), N, T_, T_, Mp, H, hd, None))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert int((outs[1] != 0).sum()) > Mp * D  # (something was written)
