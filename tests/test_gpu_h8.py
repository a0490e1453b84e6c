"""GPU (-m gpu): the fp16 + e4m3-residual operand form (precision="fp16f8", OSUD_PREC_F16F8) through the C ABI.

The tolerance tier's faster form: the split-bf16 tier (tests/test_gpu_x3.py) with the four big GEMMs of every block -- in_proj,
out_proj, fc1, fc2 -- on operands v = hi + 2^-12 lo8 with hi = fp16(v) (11 significand bits) and lo8 = e4m3((v - hi) 2^12) (4 more):
a product over 32 k is hi.hi on two v_mfma_f32_32x32x16_f16 plus BOTH cross terms (lo8_a.hi8_w + hi8_a.lo8_w) in ONE block-scaled
v_mfma_scale_f32_32x32x64_f8f6f4 -- 32 matrix-pipe passes where the split-bf16 form issues 48 -- with fp32 accumulation.  Rows are
K-blocked: 128-byte groups of 32 logical columns [64 B fp16 hi | 32 B plane P | 32 B plane Q] (activations: P = lo8, Q = hi8; weights:
P = hi8, Q = lo8), which is the GEMM's LDS stage image.

Tolerances: operators against an fp64 evaluation of the ORIGINAL fp32 operands; model outputs and loops against the fixtures frozen
from the reference; each bound <= 3x what was measured on MI355X (values in the comments, printed by the tests).
"""
import numpy as np
import pytest
import torch

from osu_diffusion_amd import _lib
from osu_diffusion_amd.diffusion import create_diffusion
from tests.helpers import T, load, maxdiff, weights_for
from tests.test_gpu_forward import FWD_TAGS, native_model
from tests.test_gpu_x3 import _p1000_inputs, from_x3

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
H8 = _lib.PREC_F16F8


def pack_h8_torch(t, weight):
    """(R, C) fp32, C % 32 == 0 -> (R, 4C) uint8 in the K-blocked layout (a torch restatement of csrc/common.h: store4_h8)."""
    t = t.detach().cpu().float()
    R, C = t.shape
    hi = t.to(torch.float16)
    lo8 = ((t - hi.float()) * 4096.0).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    hi8 = t.clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    P, Q = (hi8, lo8) if weight else (lo8, hi8)
    return torch.cat([hi.view(torch.uint8).view(R, C // 32, 64), P.view(R, C // 32, 32), Q.view(R, C // 32, 32)], dim=2).reshape(R, 4 * C).contiguous()


def pack_h8(t, weight):
    R, C = t.shape
    out = torch.empty(R, 4 * C, dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib().osud_op_pack_h8(_lib.ptr(t), C, C, _lib.ptr(out), C, R, 1 if weight else 0, None))
    return out


def from_h8(buf, rows, cols):
    """Decode an ACTIVATION-flavour buffer: hi + lo8 / 4096 (what the next GEMM's arithmetic sees, up to the hi8 partner plane)."""
    g = buf.cpu().view(torch.uint8).view(rows, cols // 32, 128)
    hi = g[:, :, :64].contiguous().view(torch.float16).float().view(rows, cols)
    lo = g[:, :, 64:96].contiguous().view(torch.float8_e4m3fn).float().view(rows, cols)
    return hi + lo / 4096.0


@pytest.mark.parametrize("weight", [False, True])
def test_pack_kernel_matches_the_format_restatement(weight):
    torch.manual_seed(3)
    a = torch.randn(192, 96, device=DEV) * torch.logspace(-4, 2, 96, device=DEV)  # columns from 1e-4 to 1e2: fp16 / e4m3 subnormals too
    a[0, :4] = torch.tensor([0.0, -0.0, 500.0, -1000.0])  # beyond e4m3's 448: the 8-bit planes saturate, hi does not
    d, r = pack_h8(a, weight).cpu(), pack_h8_torch(a, weight)
    assert int((d != r).sum()) == 0
    # zero padding of a narrower source
    out = torch.empty(8, 4 * 64, dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib().osud_op_pack_h8(_lib.ptr(a), 96, 40, _lib.ptr(out), 64, 8, int(weight), None))
    ref = torch.zeros(8, 64)
    ref[:, :40] = a[:8, :40].cpu()
    assert torch.equal(out.cpu(), pack_h8_torch(ref, weight))


@pytest.mark.parametrize("shape", [(128, 128, 32), (256, 384, 576), (128, 3072, 768), (256, 768, 3072), (16384, 768, 768)])
def test_gemm_fp16_e4m3_is_a_15_bit_product(shape):
    My, Nx, K = shape
    torch.manual_seed(My + Nx + K)
    Y = torch.randn(My, K, device=DEV) * 2.0
    X = torch.randn(Nx, K, device=DEV) / K ** 0.5  # asymmetric operands: a transposed result cannot pass
    bias = torch.randn(Nx, device=DEV)
    ref = (Y.double() @ X.double().T + bias.double()).float()
    Yc, Xc = pack_h8(Y, False), pack_h8(X, True)
    out = torch.zeros(My, Nx, device=DEV)
    _lib.check(_lib.lib().osud_op_gemm(H8, _lib.EPI_BIAS_F32, _lib.ptr(Yc), K, _lib.ptr(Xc), K, My, Nx, K,
                                       _lib.ptr(out), Nx, _lib.ptr(bias), None, 0, 0, 0, None))
    err = maxdiff(out.cpu(), ref.cpu())
    f16 = maxdiff((Y.half().double() @ X.half().double().T + bias.double()).float().cpu(), ref.cpu())
    print(f"MEASURED h8_gemm{shape}: max|d| = {err:.3e} at output scale {float(ref.abs().max()):.1f} (plain fp16 operands: {f16:.3e})")
    assert err < 3.5e-4 and err < f16 / 8  # measured 1.1-1.2e-4 (the split-bf16 form: 4-5e-5; plain fp16 operands 2.7-3.1e-3)


def test_gemm_fp16_e4m3_fused_epilogues():
    """in_proj's epilogue writes split-bf16 planes (the attention kernel's input), fc1's writes the next GEMM's h8 rows, out_proj /
    fc2 update the fp32 residual through the gate."""
    My, Nx, K, Tp, NS = 256, 256, 128, 64, 3
    torch.manual_seed(5)
    Y = torch.randn(My, K, device=DEV)
    X = torch.randn(Nx, K, device=DEV) / K ** 0.5
    bias = torch.randn(Nx, device=DEV)
    gate = torch.randn(NS + 1, Nx, device=DEV)
    res = torch.randn(My, Nx, device=DEV)
    Yc, Xc = pack_h8(Y, False), pack_h8(X, True)
    z = (Y.double() @ X.double().T + bias.double())
    L = _lib.lib()

    def run(epi, out, g=None):
        _lib.check(L.osud_op_gemm(H8, epi, _lib.ptr(Yc), K, _lib.ptr(Xc), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(bias),
                                  None if g is None else _lib.ptr(g), Nx, Tp, NS + 1, None))
        return out

    o1 = run(_lib.EPI_BIAS_TE, torch.zeros(My, 2 * Nx, dtype=torch.bfloat16, device=DEV))
    e1 = maxdiff(from_x3(o1, My, Nx).cpu(), z.float().cpu())
    o4 = run(_lib.EPI_BIAS_GELU_TE, torch.zeros(My, 4 * Nx, dtype=torch.uint8, device=DEV))
    gelu = torch.nn.functional.gelu(z, approximate="tanh").float()
    e4 = maxdiff(from_h8(o4, My, Nx), gelu.cpu())
    o5 = run(_lib.EPI_GATE_RES, res.clone(), gate)
    want5 = res.double() + gate.double().repeat_interleave(Tp, 0)[:My] * z
    e5 = maxdiff(o5.cpu(), want5.float().cpu())
    # the GELU output as the next GEMM's operand: (gelu) . X2^T against fp64
    X2 = torch.randn(128, Nx, device=DEV) / Nx ** 0.5
    out2 = torch.zeros(My, 128, device=DEV)
    _lib.check(L.osud_op_gemm(H8, _lib.EPI_NONE_F32, _lib.ptr(o4), Nx, _lib.ptr(pack_h8(X2, True)), Nx, My, 128, Nx, _lib.ptr(out2), 128,
                              None, None, 0, 0, 0, None))
    e6 = maxdiff(out2.cpu(), (gelu.double().to(DEV) @ X2.double().T).float().cpu())
    print(f"MEASURED h8_epilogues: bias -> split-bf16 planes {e1:.2e}, gelu -> h8 rows {e4:.2e}, gated residual {e5:.2e}, chained product {e6:.2e}")
    assert e1 < 1.5e-4 and e4 < 1.5e-4 and e5 < 3e-4 and e6 < 1.5e-4  # measured: see the printed line


def test_gemm_fp16_e4m3_rejects_what_it_does_not_build():
    Y = torch.zeros(128, 4 * 64, dtype=torch.uint8, device=DEV)
    out = torch.zeros(128, 128, device=DEV)
    L = _lib.lib()
    rc = L.osud_op_gemm(H8, _lib.EPI_NONE_F32, _lib.ptr(Y), 64, _lib.ptr(Y), 64, 128, 128, 48, _lib.ptr(out), 128, None, None, 0, 0, 0, None)
    assert rc == _lib.ERR_ARG  # K % 32
    rc = L.osud_op_gemm(H8, 7, _lib.ptr(Y), 64, _lib.ptr(Y), 64, 128, 128, 64, _lib.ptr(out), 128, None, None, 0, 0, 0, None)
    assert rc == _lib.ERR_UNSUPPORTED  # a training epilogue


@pytest.mark.parametrize("tag", FWD_TAGS)
def test_forward_matches_reference_golden_fp16_e4m3(tag):
    fx = load(f"g3_forward_{tag}")
    shape, sd = weights_for(fx)
    m = native_model(shape, sd, "fp16f8")
    x, t, o, c, y = (T(fx[k]).to(DEV) for k in ("x", "t", "o", "c", "y"))
    mask = T(fx["attn_mask"]).to(DEV) if "attn_mask" in fx else None
    scale = float(np.abs(fx["out"]).max())
    with torch.no_grad():
        errs = {"out": maxdiff(m(x, t, o, c, y, attn_mask=mask).cpu(), fx["out"]),
                "cfg4": maxdiff(m.forward_with_cfg(x, t, o, c, y, 4.0, attn_mask=mask).cpu(), fx["out_cfg4"]),
                "cfg1": maxdiff(m.forward_with_cfg(x, t, o, c, y, 1.0, attn_mask=mask).cpu(), fx["out_cfg1"])}
    print(f"MEASURED h8_forward[{tag}]: scale {scale:.2f}, errors {({k: f'{v:.2e}' for k, v in errs.items()})}")
    assert max(errs["out"], errs["cfg1"]) <= H8_FWD * max(scale, 1.0) and errs["cfg4"] <= 5 * H8_FWD * max(scale, 1.0), errs


H8_FWD = 2e-4  # plain forward / max(scale, 1): 3x the measured value (split-bf16 tier: 7e-5 for a measured 1.3-2.6e-5)


@pytest.mark.parametrize("tag", ["p20", "ddim20_eta1", "p250"])
def test_chained_loop_final_coordinates_fp16_e4m3(tag):
    fx = load(f"g6_loop_{tag}")
    shape, sd = weights_for(fx)
    m = native_model(shape, sd, "fp16f8")
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    d = create_diffusion(str(fx["respacing"]), noise_schedule="squaredcos_cap_v2")
    z = T(fx["z"]).to(DEV)
    eta = float(fx["eta"])
    if eta >= 0:
        got = d.ddim_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, eta=eta, step_noise=T(fx["noises"]))
    else:
        got = d.p_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=T(fx["noises"]))
    err = maxdiff(got.cpu(), fx["final"])
    print(f"MEASURED h8_loop[{tag}]: final max|d| = {err:.3e}")
    assert err < 5e-4  # the north star's bound is 1e-3


def test_dit_b_1000_step_cfg4_loop_matches_the_reference_fp16_e4m3():
    """BASELINE configs[3] end to end (as tests/test_gpu_x3.py::test_dit_b_1000_step_cfg4_loop_matches_the_reference): within 1e-3 of the
    reference's own 1000-step CFG-4 DiT-B loop at every quarter."""
    fx, shape, sd, z, noises = _p1000_inputs()
    m = native_model(shape, sd, "fp16f8")
    d = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    x = z.to(DEV).clone()
    nz = noises.to(DEV)
    errs, done = {}, 0
    for k in (250, 500, 750, 1000):
        d.run_steps(m.forward_with_cfg, x, kw, first_step=999 - done, last_step=1000 - k, step_noise=nz[done:k])
        errs[k] = maxdiff(x.cpu(), fx["final"] if k == 1000 else fx[f"after_{k}"])
        done = k
    print("MEASURED p1000_dit_b[fp16f8]: max|d| vs reference after 250/500/750/1000 steps = " + " / ".join(f"{errs[k]:.3e}" for k in (250, 500, 750, 1000)))
    assert max(errs.values()) < 1e-3, errs


@pytest.mark.parametrize("hidden,heads,T_", [(1152, 16, 256), (1152, 16, 200), (1024, 16, 128)])
def test_forward_on_other_geometries_against_the_fp32_oracle(hidden, heads, T_):
    """DiT-XL's geometry (1152 = 18 columns per lane: the 2-wide row stores; heads of 72 columns, which straddle the 32-column groups of
    the operand rows) and DiT-L's, masked and not: both tolerance tiers against the fp32 oracle (no reference fixture at these widths)."""
    from oracle import dit_oracle as mo
    from osu_diffusion_amd.synthetic import banded_attn_mask, synthetic_windows

    shape = mo.DitShape(depth=2, hidden=hidden, heads=heads, num_classes=10)
    sd = mo.seeded_state_dict(shape, 78)
    (x, o, c), y = synthetic_windows(3, T_, 10, seed=6)
    t = torch.tensor([999, 400, 0])
    mask = banded_attn_mask(T_, 128) if T_ == 200 else None
    ref = mo.forward(sd, shape, x, t, o, c, y, attn_mask=mask)
    scale = max(1.0, float(ref.abs().max()))
    errs = {}
    for prec in ("bf16x3", "fp16f8"):
        with torch.no_grad():
            got = native_model(shape, sd, prec)(x.to(DEV), t.to(DEV), o.to(DEV), c.to(DEV), y.to(DEV),
                                                attn_mask=None if mask is None else mask.to(DEV)).cpu()
        errs[prec] = maxdiff(got, ref)
    print(f"MEASURED h8_forward_geometry[{hidden},{heads},{T_}]: max|d| vs the fp32 oracle at scale {scale:.2f}: "
          + ", ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["bf16x3"] < 1.5e-5 * scale and errs["fp16f8"] < 1.8e-5 * scale, errs  # 3x measured (4.2-4.9e-6 / 4.6-5.6e-6 per unit of scale)
