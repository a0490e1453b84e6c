"""GPU (-m gpu): parity of the native gfx950 path (through the C ABI) against the CPU oracle and
the golden vectors frozen from the reference.

Tolerances (stated per north_star):
  * parity tier (fp32, exact-f32 MFMA): |out - ref| <= 2e-4 on model outputs of O(1); final
    sampled coordinates within 1e-3 of the reference after a chained CFG-4 loop.
  * fast tier (bf16 MFMA operands, fp32 accumulate/residual/statistics): teacher-forced (same
    inputs) |out - ref| <= 1e-2 * max|ref|; end-to-end drift is reported, not asserted to 1e-3.
  * sampler update given the same model output: <= 1e-6 (differs only by expf's last ulp).
"""
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as do
from oracle import dit_oracle as mo
from osu_diffusion_amd import _lib
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT
from osu_diffusion_amd.synthetic import banded_attn_mask, synthetic_windows
from tests.helpers import T, load, maxdiff, weights_for

pytestmark = pytest.mark.gpu
# bf16 tier bounds: 3x what MI355X measured (the tests print MEASURED ... lines).  Forward: max|d| / max|ref| per fixture for the plain
# and the cfg-4 outputs (round 3, gpurun_out/r3b: deeper / wider / rougher models round more)
BF16_FWD_MEASURED = {"tiny_T64": (2.74e-4, 7.87e-4), "tiny_T128": (2.60e-4, 8.73e-4), "tiny_T200_band": (2.45e-4, 9.86e-4),
                     "tiny_T128_allfalse": (2.29e-4, 7.86e-4), "small_T128": (6.81e-4, 2.70e-3), "tiny_T128_rough": (5.32e-4, 1.89e-3),
                     "dit_b_T128": (1.33e-3, 3.91e-3), "dit_b_T128_rough": (4.32e-3, 1.75e-2)}
BF16_P20_DRIFT = 5e-3  # measured 1.53e-3

DEV = "cuda:0"


def native_model(shape, sd, precision):
    m = DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads, context_size=shape.context,
            num_classes=shape.num_classes, class_dropout_prob=0.2, precision=precision)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def to_elem(prec, t):
    out = torch.empty(t.numel() * (2 if prec == 0 else 4), dtype=torch.uint8, device=DEV)
    _lib.check(_lib.lib().osud_op_convert(prec, _lib.ptr(t.contiguous()), _lib.ptr(out), t.numel(), None))
    return out


def from_elem(prec, buf, shape):
    return buf.view(torch.bfloat16).view(shape).float() if prec == 0 else buf.view(torch.float32).view(shape).clone()


# ------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("prec", [_lib.PREC_BF16, _lib.PREC_F32])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 384, 576), (128, 3072, 768), (256, 768, 3072)])
def test_gemm_bias(prec, shape):
    My, Nx, K = shape
    torch.manual_seed(My + Nx + K)
    Y = torch.randn(My, K, device=DEV)
    X = torch.randn(Nx, K, device=DEV) / K ** 0.5  # asymmetric operands: a transposed result cannot pass
    bias = torch.randn(Nx, device=DEV)
    Yc, Xc = to_elem(prec, Y), to_elem(prec, X)
    ref = (from_elem(prec, Yc, (My, K)).double() @ from_elem(prec, Xc, (Nx, K)).double().T + bias.double()).float()
    out = torch.zeros(My, Nx, device=DEV)
    _lib.check(_lib.lib().osud_op_gemm(prec, _lib.EPI_BIAS_F32, _lib.ptr(Yc), K, _lib.ptr(Xc), K, My, Nx, K,
                                       _lib.ptr(out), Nx, _lib.ptr(bias), None, 0, 0, 0, None))
    assert maxdiff(out.cpu(), ref.cpu()) < 1e-4


@pytest.mark.parametrize("prec", [_lib.PREC_BF16, _lib.PREC_F32])
def test_gemm_fused_epilogues(prec):
    My, Nx, K, Tp, NS = 256, 256, 128, 64, 3  # 3 real samples of 64 rows + 64 padding rows
    torch.manual_seed(5)
    Y = torch.randn(My, K, device=DEV)
    X = torch.randn(Nx, K, device=DEV) / K ** 0.5
    bias, rbias = torch.randn(Nx, device=DEV), torch.randn(My, device=DEV)
    gate = torch.randn(NS, Nx, device=DEV)
    Yc, Xc = to_elem(prec, Y), to_elem(prec, X)
    acc = from_elem(prec, Yc, (My, K)).double() @ from_elem(prec, Xc, (Nx, K)).double().T
    L = _lib.lib()
    tol = 2e-2 if prec == _lib.PREC_BF16 else 1e-4  # bf16 outputs are rounded to bf16

    def run(epi, out, b, g=None):
        _lib.check(L.osud_op_gemm(prec, epi, _lib.ptr(Yc), K, _lib.ptr(Xc), K, My, Nx, K, _lib.ptr(out), Nx, _lib.ptr(b),
                                  _lib.ptr(g), Nx if g is not None else 0, Tp, NS, None))
        torch.cuda.synchronize()

    esz = 2 if prec == 0 else 4
    o = torch.zeros(My * Nx * esz, dtype=torch.uint8, device=DEV)
    run(_lib.EPI_BIAS_TE, o, bias)
    assert maxdiff(from_elem(prec, o, (My, Nx)).cpu(), (acc + bias.double()).cpu()) < tol * 4
    run(_lib.EPI_BIAS_SILU_TE, o, bias)
    assert maxdiff(from_elem(prec, o, (My, Nx)).cpu(), torch.nn.functional.silu(acc + bias.double()).cpu()) < tol * 4
    run(_lib.EPI_BIAS_GELU_TE, o, bias)
    ref = torch.nn.functional.gelu(acc + bias.double(), approximate="tanh")
    assert maxdiff(from_elem(prec, o, (My, Nx)).cpu(), ref.cpu()) < tol * 4
    run(_lib.EPI_ROWBIAS_TE, o, rbias)
    assert maxdiff(from_elem(prec, o, (My, Nx)).cpu(), (acc + rbias.double()[:, None]).cpu()) < tol * 4
    res = torch.randn(My, Nx, device=DEV)
    want = res.double() + gate.double()[torch.clamp(torch.arange(My, device=DEV) // Tp, max=NS - 1)] * (acc + bias.double())
    run(_lib.EPI_GATE_RES, res, bias, gate)
    assert maxdiff(res.cpu(), want.cpu()) < 1e-4


def test_gemm_rejects_bad_shapes():
    a = torch.zeros(128 * 64 * 4, dtype=torch.uint8, device=DEV)
    o = torch.zeros(128, 128, device=DEV)
    rc = _lib.lib().osud_op_gemm(0, 0, _lib.ptr(a), 64, _lib.ptr(a), 64, 100, 128, 64, _lib.ptr(o), 128, _lib.ptr(o), None,
                                 0, 0, 0, None)
    assert rc == _lib.ERR_ARG and "multiples of 128" in _lib.last_error()


# ------------------------------------------------------------------------------------ attention
@pytest.mark.parametrize("prec,tol", [(_lib.PREC_BF16, 2e-2), (_lib.PREC_F32, 2e-5)])
@pytest.mark.parametrize("T_,masked,N,H,hd", [(64, False, 2, 2, 64), (128, False, 2, 2, 64), (200, True, 2, 2, 64), (77, False, 2, 2, 64),
                                               (128, False, 41, 13, 64), (128, True, 3, 2, 64), (256, False, 2, 3, 72),
                                               (256, False, 34, 16, 72), (200, True, 2, 2, 72)])
def test_attention_core(prec, tol, T_, masked, N, H, hd):
    """(N, H) = (41, 13): 533 heads at T = 128 without a mask -- the persistent streamed kernel of the bf16 tier with more head pairs
    than compute units (a second loop iteration on some workgroups) and an odd head count (a half-empty last pair).  head_dim 72 at
    T = 256 is DiT-XL's shape and has its own streamed kernel (544 heads: up to three per workgroup)."""
    D = H * hd
    Tp = (T_ + 63) // 64 * 64
    Mp = (N * Tp + 127) // 128 * 128
    torch.manual_seed(T_)
    qkv = torch.randn(Mp, 3 * D, device=DEV)  # packed in_proj output: Q | K | V
    mask = banded_attn_mask(T_, 128).to(DEV) if masked else None
    qkc = to_elem(prec, qkv)
    qkr = from_elem(prec, qkc, (Mp, 3 * D))
    vr = qkr[:, 2 * D:]
    out = torch.zeros(Mp * D * (2 if prec == 0 else 4), dtype=torch.uint8, device=DEV)
    m8 = None if mask is None else mask.to(torch.uint8).contiguous()
    _lib.check(_lib.lib().osud_op_attention(prec, _lib.ptr(qkc), 3 * D, _lib.ptr(m8), _lib.ptr(out), N, T_, Tp, Mp, H,
                                            hd, None))
    got = from_elem(prec, out, (Mp, D))
    for n in range(N):
        rows = slice(n * Tp, n * Tp + T_)
        q = qkr[rows, :D].reshape(T_, H, hd).transpose(0, 1).double()
        k = qkr[rows, D:2 * D].reshape(T_, H, hd).transpose(0, 1).double()
        vv = vr[rows].reshape(T_, H, hd).transpose(0, 1).double()
        s = q @ k.transpose(-1, -2) / hd ** 0.5
        if mask is not None:
            s = s.masked_fill(mask, float("-inf"))
        ref = (torch.softmax(s, -1) @ vv).transpose(0, 1).reshape(T_, D)
        assert maxdiff(got[rows].cpu(), ref.cpu()) < tol, (n, T_)


@pytest.mark.parametrize("prec,tol", [(_lib.PREC_BF16, 3e-2), (_lib.PREC_F32, 5e-5)])
@pytest.mark.parametrize("N,H,T_,stream,hd", [(3, 2, 64, "1", 64), (24, 12, 128, "1", 64), (24, 12, 128, "0", 64), (2, 2, 256, "1", 64),
                                              (2, 2, 320, "1", 64), (2, 3, 256, "1", 72), (2, 3, 256, "0", 72), (34, 16, 256, "1", 72),
                                              (2, 2, 128, "1", 72)])
def test_attention_core_backward(prec, tol, N, H, T_, stream, hd, osud_option):
    """osud_op_attention_bwd against torch autograd of softmax(q k^T / sqrt(hd)) v on the same (rounded) operands.  N*H = 288 heads
    at T = 128 is more than one per compute unit: the persistent streamed kernel runs its double-buffered loop (a second head on
    32 of the workgroups); option attn_bwd_kernel = 1 selects the one-workgroup-per-head kernel; T = 320 the tiled one.  head_dim 72 at
    T = 256 is DiT-XL's shape: its own streamed kernel (34 x 16 = 544 heads: up to three per workgroup), or the tiled one with "0"."""
    osud_option("attn_bwd_kernel", 0 if stream == "1" else 1)
    D = H * hd
    M = N * T_
    torch.manual_seed(N * 1000 + T_)
    qkv = torch.randn(M, 3 * D, device=DEV)
    dout = torch.randn(M, D, device=DEV)
    qkc, doc = to_elem(prec, qkv), to_elem(prec, dout)
    qkr = from_elem(prec, qkc, (M, 3 * D)).double().requires_grad_(True)
    dor = from_elem(prec, doc, (M, D)).double()
    q, k, v = (qkr[:, i * D:(i + 1) * D].reshape(N, T_, H, hd).transpose(1, 2) for i in range(3))
    s = q @ k.transpose(-1, -2) / hd ** 0.5
    o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(M, D)
    o.backward(dor)
    lse = torch.logsumexp(s, -1)  # (N, H, T)
    if prec == _lib.PREC_BF16:
        lse = lse * 1.4426950408889634  # the bf16 tier keeps it in the log2 domain
    lse = lse.float().contiguous()
    oc = to_elem(prec, o.detach().float())
    got = torch.zeros(M * 3 * D * (2 if prec == 0 else 4), dtype=torch.uint8, device=DEV)
    ws = torch.zeros(N * H * T_, device=DEV)
    _lib.check(_lib.lib().osud_op_attention_bwd(prec, _lib.ptr(qkc), _lib.ptr(doc), _lib.ptr(oc), _lib.ptr(lse), _lib.ptr(got), N, T_,
                                                H, hd, _lib.ptr(ws), None))
    g = from_elem(prec, got, (M, 3 * D))
    ref = qkr.grad.float()
    assert maxdiff(g.cpu(), ref.cpu()) < tol * max(1.0, float(ref.abs().max())), (N, H, T_)


@pytest.mark.selfcheck
@pytest.mark.parametrize("N,H,T_,hd", [(96, 12, 128, 64), (34, 16, 256, 72)])
def test_attention_head_queue_equals_fixed_stride(N, H, T_, hd):
    """Shared-GPU mode (osud_set_gemm_dynamic_tiles(1), what data-parallel trainers switch on): the persistent attention kernels draw
    their heads from a ticket queue instead of a fixed stride.  1152 heads (576 pairs) at T = 128 / 544 heads of DiT-XL's shape are
    more than two per compute unit, so the queue is live; which workgroup computes a head must not change a bit of the result."""
    D, M = H * hd, N * T_
    torch.manual_seed(5)
    qkc = to_elem(0, torch.randn(M, 3 * D, device=DEV))
    doc = to_elem(0, torch.randn(M, D, device=DEV))
    lse = torch.randn(N, H, T_, device=DEV) + 6.0
    L = _lib.lib()
    res = []
    try:
        for mode in (0, 1, 1):  # the second queued launch runs on counters the first one re-armed
            L.osud_set_gemm_dynamic_tiles(mode)
            out = torch.zeros(M * D * 2, dtype=torch.uint8, device=DEV)
            dq = torch.zeros(M * 3 * D * 2, dtype=torch.uint8, device=DEV)
            _lib.check(L.osud_op_attention(0, _lib.ptr(qkc), 3 * D, None, _lib.ptr(out), N, T_, T_, M, H, hd, None))
            ws = torch.zeros(N * H * T_, device=DEV)
            _lib.check(L.osud_op_attention_bwd(0, _lib.ptr(qkc), _lib.ptr(doc), _lib.ptr(out), _lib.ptr(lse), _lib.ptr(dq), N, T_, H, hd,
                                               _lib.ptr(ws), None))
            torch.cuda.synchronize()
            res.append((out.clone(), dq.clone()))
    finally:
        L.osud_set_gemm_dynamic_tiles(-1)
    for o, g in res[1:]:
        assert torch.equal(o, res[0][0]) and torch.equal(g, res[0][1])


# ------------------------------------------------------------------------------------ forward
FWD_TAGS = ["tiny_T64", "tiny_T128", "tiny_T200_band", "tiny_T128_allfalse", "small_T128", "tiny_T128_rough",
            "dit_b_T128", "dit_b_T128_rough"]  # dit_b: D=768, 12 heads, 12 blocks -- the geometry bench.py times


@pytest.mark.parametrize("tag", FWD_TAGS)
@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_forward_matches_reference_golden(tag, precision):
    fx = load("g3_forward_" + tag)
    shape, sd = weights_for(fx)
    m = native_model(shape, sd, precision)
    mask = T(fx["attn_mask"]).to(DEV) if "attn_mask" in fx else None
    args = [T(fx[k]).to(DEV) for k in ("x", "t", "o", "c", "y")]
    with torch.no_grad():
        out = m(*args, attn_mask=mask)
        cfg4 = m.forward_with_cfg(*args, 4.0, attn_mask=mask)
        cfg1 = m.forward_with_cfg(*args, 1.0, attn_mask=mask)
    scale = float(np.abs(fx["out"]).max())
    # fp32: a fixed bound.  bf16: 3 x the error this implementation measured on the fixture (BF16_FWD_MEASURED) -- a REGRESSION bound fitted to
    # the implementation, not an accuracy claim: the bf16 tier is outside the 1e-3 end-to-end tolerance and says so.
    tol = 2e-4 * max(scale, 1.0) if precision == "fp32" else 3 * BF16_FWD_MEASURED[tag][0] * scale
    assert out.shape == (len(args[0]), 4, args[0].shape[2]) and out.dtype == torch.float32
    e_out, e_cfg4, e_cfg1 = maxdiff(out.cpu(), fx["out"]), maxdiff(cfg4.cpu(), fx["out_cfg4"]), maxdiff(cfg1.cpu(), fx["out_cfg1"])
    print(f"MEASURED forward[{tag},{precision}]: scale {scale:.3f}, out {e_out:.3e} ({e_out / scale:.2e} x scale), cfg4 {e_cfg4:.3e} "
          f"({e_cfg4 / scale:.2e} x scale), cfg1 {e_cfg1:.3e}")
    assert e_out < tol
    assert e_cfg4 < (7 * tol if precision == "fp32" else 3 * BF16_FWD_MEASURED[tag][1] * scale)  # guidance amplifies cond-uncond differences by 4 (+3)
    assert e_cfg1 < tol
    n = len(cfg4) // 2
    assert torch.equal(cfg4[:n, :2], cfg4[n:, :2])  # same guided eps in both halves (models.py:342)


def test_forward_validates_inputs_like_the_reference():
    shape = mo.DitShape(depth=1, hidden=128, heads=2, num_classes=4)
    m = native_model(shape, mo.seeded_state_dict(shape, 1), "bf16")
    (x, o, c), y = synthetic_windows(2, 64, 4, seed=0)
    t = torch.zeros(2, dtype=torch.long)
    with torch.no_grad():
        with pytest.raises(AssertionError):
            m(x[:, :1], t, o, c, y)
        with pytest.raises(AssertionError):
            m(x, t, o[:, :10], c, y)
        with pytest.raises(AssertionError):
            m.forward_with_cfg(x[:1], t[:1], o[:1], c[:1], y[:1], 4.0)
        out = m(x, t, o, c, y)  # CPU inputs are moved to the module's device
    assert out.is_cuda and torch.isfinite(out).all()


def test_parameter_updates_are_picked_up():
    shape = mo.DitShape(depth=1, hidden=128, heads=2, num_classes=4)
    sd = mo.seeded_state_dict(shape, 2)
    m = native_model(shape, sd, "fp32")
    (x, o, c), y = synthetic_windows(2, 64, 4, seed=1)
    t = torch.tensor([3, 700])
    with torch.no_grad():
        a = m(x, t, o, c, y)
        m.final_layer.linear.bias.add_(1.0)
        b = m(x, t, o, c, y)
    assert maxdiff((b - a).cpu(), torch.ones_like(a).cpu()) < 1e-5
    with torch.no_grad():
        ref = mo.forward({k: v.detach().cpu() for k, v in m.state_dict().items()}, shape, x, t, o, c, y)
    assert maxdiff(b.cpu(), ref) < 2e-4


# ------------------------------------------------------------------------------------ sampler
@pytest.mark.parametrize("tag", ["1000", "250"])
def test_sampler_step_against_reference_golden(tag):
    fx = load("g5_step_" + tag)
    d = create_diffusion(tag, noise_schedule="squaredcos_cap_v2")
    x, t, mout = (T(fx[k]).to(DEV) for k in ("x", "t", "model_out"))
    N, _, TT = x.shape
    L = _lib.lib()
    for mode, eta, key, clip in [(0, 0.0, "p", 1), (1, 0.0, "ddim0", 1), (1, 1.0, "ddim1", 1), (0, 0.0, "p_noclip", 0)]:
        out, x0 = torch.empty_like(x), torch.empty_like(x)
        nz = T(fx[("p" if key == "p_noclip" else key) + "_noise"]).to(DEV)
        _lib.check(L.osud_sampler_step(d._sched.handle, mode, eta, _lib.ptr(mout), _lib.ptr(x), _lib.ptr(t), _lib.ptr(nz), N, TT,
                                       -1.0, clip, _lib.ptr(out), _lib.ptr(x0), None))
        assert maxdiff(out.cpu(), fx[key + "_sample"]) < 1e-6, key
        if clip:
            assert maxdiff(x0.cpu(), fx[key + "_x0"]) == 0.0, key


def test_sampler_step_fuses_cfg():
    """cfg inside the update == forward_with_cfg's combine followed by the plain update."""
    sch = do.create_schedule("250", "squaredcos_cap_v2")
    d = create_diffusion("250", noise_schedule="squaredcos_cap_v2")
    g = torch.Generator().manual_seed(9)
    N, TT = 6, 96
    raw = torch.randn(N, 4, TT, generator=g)
    x, nz = torch.randn(N, 2, TT, generator=g), torch.randn(N, 2, TT, generator=g)
    t = torch.tensor([0, 5, 249, 0, 5, 249])
    ce, ue = raw[:3, :2], raw[3:, :2]
    he = ue + 4.0 * (ce - ue)
    combined = torch.cat([torch.cat([he, he]), raw[:, 2:]], dim=1)
    want = do.p_sample_step(sch, combined, x, t, nz)["sample"]
    out = torch.empty(N, 2, TT, device=DEV)
    a = [v.to(DEV) for v in (raw, x, t, nz)]
    _lib.check(_lib.lib().osud_sampler_step(d._sched.handle, 0, 0.0, _lib.ptr(a[0]), _lib.ptr(a[1]), _lib.ptr(a[2]), _lib.ptr(a[3]),
                                            N, TT, 4.0, 1, _lib.ptr(out), None, None))
    assert maxdiff(out.cpu(), want) < 1e-6


def test_p_sample_api_native_step_matches_oracle():
    shape = mo.DitShape(depth=2, hidden=128, heads=2, num_classes=10)
    sd = mo.seeded_state_dict(shape, 11)
    m = native_model(shape, sd, "fp32")
    d = create_diffusion("250", noise_schedule="squaredcos_cap_v2")
    sch = do.create_schedule("250", "squaredcos_cap_v2")
    (x, o, c), y = synthetic_windows(2, 64, 10, seed=3, train_offsets=False)
    x = torch.cat([x, x]); o = torch.cat([o, o]); c = torch.cat([c, c]); y = torch.cat([y, torch.full_like(y, 10)])
    t = torch.tensor([100, 100, 100, 100])
    kw = dict(o=o.to(DEV), c=c.to(DEV), y=y.to(DEV), cfg_scale=4.0, attn_mask=None)
    torch.manual_seed(1)
    with torch.no_grad():
        r = d.p_sample(m.forward_with_cfg, x.to(DEV), t.to(DEV), clip_denoised=True, model_kwargs=kw)
    torch.manual_seed(1)
    nz = torch.randn(x.shape, device=DEV).cpu()  # the native step draws randn_like(x) on the device
    mout = mo.forward_with_cfg(sd, shape, x, torch.from_numpy(sch.timestep_map)[t], o, c, y, 4.0)
    want = do.p_sample_step(sch, mout, x, t, nz)
    assert maxdiff(r["sample"].cpu(), want["sample"]) < 5e-4
    assert maxdiff(r["pred_xstart"].cpu(), want["pred_xstart"]) < 5e-4


@pytest.mark.parametrize("tag", ["p20", "ddim20_eta1", "ddim20_eta05", "p250"])
def test_chained_loop_final_coordinates_fp32(tag):
    """identical (seed -> noise, window, num-sampling-steps): final (x, y) within 1e-3 of the reference."""
    fx = load("g6_loop_" + tag)
    shape, sd = weights_for(fx)
    m = native_model(shape, sd, "fp32")
    d = create_diffusion(str(fx["respacing"]), noise_schedule="squaredcos_cap_v2")
    z = T(fx["z"]).to(DEV)
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    eta = float(fx["eta"])
    finals = {}
    for graph in ("graph", "eager"):
        _lib.set_option("sample_graph", 1 if graph == "graph" else 0)
        try:
            if eta < 0:
                fin = d.p_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=T(fx["noises"]))
            else:
                fin = d.ddim_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, eta=eta, step_noise=T(fx["noises"]))
        finally:
            _lib.set_option("sample_graph", -1)
        finals[graph] = fin.cpu()
        assert maxdiff(finals[graph], fx["final"]) < 1e-3, graph
    assert torch.equal(finals["graph"], finals["eager"])  # graph replay == eager launches, bit for bit


@pytest.mark.parametrize("precision", ["fp32", "fp16f8", "bf16x3"])
def test_p250_loop_on_undamped_weights(precision):
    """The same 250-step CFG-4 loop on reference-like UNDAMPED weights (pos_gain 1: position features at 512 rad per unit x).
    The fixture holds the reference's fp32 result and an fp64 evaluation of the same loop; their distance (3.8e-4) is what any
    fp32 implementation can claim here.  The parity tier AND the two tolerance tiers (fp16f8 = sample.py's default, bf16x3) must stay
    within 1e-3 of the reference's fp32 result; each is reported against the fp64 evaluation next to the reference's own distance."""
    fx = load("g6_loop_p250_undamped")
    shape = mo.DitShape(*(int(v) for v in fx["shape"][:3]), num_classes=int(fx["shape"][3]))
    sd = mo.seeded_state_dict(shape, int(fx["wseed"]), pos_gain=float(fx["pos_gain"]))
    m = native_model(shape, sd, precision)
    d = create_diffusion("250", noise_schedule="squaredcos_cap_v2")
    z = T(fx["z"]).to(DEV)
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    fin = d.p_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=T(fx["noises"])).cpu()
    d_ref, d_64, spread = maxdiff(fin, fx["final"]), maxdiff(fin, fx["final_fp64"]), maxdiff(fx["final"], fx["final_fp64"])
    print(f"MEASURED undamped p250 [{precision}]: native vs reference fp32 {d_ref:.3e}, vs fp64 evaluation {d_64:.3e} (reference fp32 vs fp64 {spread:.3e})")
    assert d_ref < 1e-3


@pytest.mark.parametrize("precision", ["fp32", "fp16f8", "bf16x3"])
@pytest.mark.parametrize("damping", ["", "_undamped"])
@pytest.mark.parametrize("part", ["head", "tail"])
def test_1000_step_schedule_head_and_tail(part, damping, precision):
    """the first five steps of the 1000-step schedule (t = 999..995: sqrt(1/ac - 1) ~ 2e4 multiplies eps, the
    clamp decides x0) and the last five (t = 4..0, the t = 0 step adds no noise), against reference p_sample calls."""
    fx = load("g6_steps_1000" + damping)
    shape = mo.DitShape(*(int(v) for v in fx["shape"][:3]), num_classes=int(fx["shape"][3]))
    sd = mo.seeded_state_dict(shape, int(fx["wseed"]), pos_gain=float(fx["pos_gain"]))
    m = native_model(shape, sd, precision)
    d = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    x = T(fx[part + "_start"]).to(DEV).clone()
    d.run_steps(m.forward_with_cfg, x, kw, first_step=int(fx[part + "_first"]), last_step=int(fx[part + "_last"]),
                step_noise=T(fx[part + "_noises"]))
    err = maxdiff(x.cpu(), fx[part + "_final"])
    print(f"MEASURED 1000-step {part}{damping} [{precision}]: native vs reference {err:.3e}")
    assert err < 1e-3


def test_chained_loop_bf16_drift_is_bounded():
    fx = load("g6_loop_p20")
    shape, sd = weights_for(fx)
    m = native_model(shape, sd, "bf16")
    d = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    z = T(fx["z"]).to(DEV)
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    fin = d.p_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=T(fx["noises"])).cpu()
    drift = maxdiff(fin, fx["final"])
    print(f"MEASURED loop_p20[bf16]: end-to-end drift after 20 CFG-4 steps {drift:.3e}")
    assert drift < BF16_P20_DRIFT


def test_loop_with_split_off_constant_first_linear_part(osud_option):
    """Inside a sampler loop the offsets / context share of the first linear is multiplied once and only the 256 coordinate features
    every step (bf16 tier).  Same sums in a different order: one step must agree with the one-product form to fp32 rounding carried
    through the bf16 trunk, the 20-step CFG-4 loop stays inside the tier's drift bound against the reference, and a second loop with
    OTHER offsets / context through the same buffers must not see stale values."""
    fx = load("g6_loop_p20")
    shape, sd = weights_for(fx)
    d = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    z = T(fx["z"]).to(DEV)
    o, c, y = T(fx["o"]).to(DEV), T(fx["c"]).to(DEV), T(fx["y"]).to(DEV)
    one, fins = {}, {}
    for flag in ("0", "1"):
        osud_option("embed_const", int(flag))
        m = native_model(shape, sd, "bf16")
        kw = dict(o=o.clone(), c=c.clone(), y=y, cfg_scale=4.0, attn_mask=None)
        one[flag] = d.run_steps(m.forward_with_cfg, z.clone(), kw, 19, 19, seed=3).cpu()
        a = d.p_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=T(fx["noises"])).cpu()
        kw["o"].add_(37.0)          # same buffers, new contents
        kw["c"].mul_(0.5)
        b = d.p_sample_loop(m.forward_with_cfg, z.shape, z, model_kwargs=kw, step_noise=T(fx["noises"])).cpu()
        fins[flag] = (a, b)
    assert maxdiff(one["0"], one["1"]) < 2e-3
    assert maxdiff(fins["1"][0], fx["final"]) < 5e-2 and maxdiff(fins["0"][0], fx["final"]) < 5e-2
    assert maxdiff(fins["0"][1], fins["1"][1]) < 5e-2
    assert maxdiff(fins["1"][0], fins["1"][1]) > 1e-1  # the second loop did see the new offsets / context


def test_generic_path_with_denoised_fn_uses_native_forward():
    """in-paint style hook (testing/test_toy.py:56-74): generic Python step around the native forward."""
    shape = mo.DitShape(depth=2, hidden=128, heads=2, num_classes=10)
    sd = mo.seeded_state_dict(shape, 11)
    m = native_model(shape, sd, "fp32")
    d = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    sch = do.create_schedule("20", "squaredcos_cap_v2")
    (x, o, c), y = synthetic_windows(2, 64, 10, seed=4, train_offsets=False)
    keep = torch.zeros(2, 2, 64, dtype=torch.bool)
    keep[:, :, :32] = True
    fn_dev = lambda v: torch.where(keep.to(v.device), x.to(v.device), v)  # noqa: E731
    fn_cpu = lambda v: torch.where(keep, x, v)  # noqa: E731
    torch.manual_seed(2)
    z = torch.randn(2, 2, 64)
    kw = dict(o=o.to(DEV), c=c.to(DEV), y=y.to(DEV))
    torch.manual_seed(5)
    got = d.p_sample_loop(m.forward, z.shape, z.to(DEV), denoised_fn=fn_dev, model_kwargs=kw, device=DEV)
    torch.manual_seed(5)
    noises = torch.stack([torch.randn(z.shape, device=DEV).cpu() for _ in range(20)])
    fn = lambda xx, tt: mo.forward(sd, shape, xx, tt, o, c, y)  # noqa: E731
    want = do.sample_loop(sch, fn, z, noises, denoised_fn=fn_cpu)
    assert maxdiff(got.cpu(), want) < 1e-3


def test_in_kernel_philox_noise_is_standard_normal_and_reproducible():
    shape = mo.DitShape(depth=1, hidden=128, heads=2, num_classes=4)
    m = native_model(shape, mo.seeded_state_dict(shape, 3), "bf16")
    d = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    (x, o, c), y = synthetic_windows(8, 128, 4, seed=5, train_offsets=False)
    kw = dict(o=o.to(DEV), c=c.to(DEV), y=y.to(DEV))
    z = torch.randn(8, 2, 128, device=DEV)
    a = d.p_sample_loop(m.forward, z.shape, z, model_kwargs=kw, seed=123)
    b = d.p_sample_loop(m.forward, z.shape, z, model_kwargs=kw, seed=123)
    c2 = d.p_sample_loop(m.forward, z.shape, z, model_kwargs=kw, seed=124)
    assert torch.equal(a, b) and not torch.equal(a, c2) and torch.isfinite(a).all()


def test_in_place_steps_on_one_buffer_follow_the_seed_not_the_cached_graph():
    """The cached hipGraph is keyed on pointers and shapes; the Philox seed travels in device memory, so run_steps on the SAME
    buffer (same pointers -> the same graph is replayed) must still draw different noise for a different seed."""
    shape = mo.DitShape(depth=1, hidden=128, heads=2, num_classes=4)
    m = native_model(shape, mo.seeded_state_dict(shape, 3), "bf16")
    d = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    (x, o, c), y = synthetic_windows(4, 64, 4, seed=6, train_offsets=False)
    kw = dict(o=o.to(DEV), c=c.to(DEV), y=y.to(DEV))
    z = torch.randn(4, 2, 64, device=DEV)
    buf = torch.empty_like(z)
    outs = []
    for seed in (7, 8, 7):
        buf.copy_(z)
        d.run_steps(m.forward, buf, kw, first_step=19, last_step=10, seed=seed)
        outs.append(buf.clone())
    assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1])


@pytest.mark.selfcheck
def test_gemm_tile_queue_equals_fixed_stride():
    """Multi-round launches (more tiles than compute units) with the per-XCD ticket queues switched on give the same bits as the
    fixed-stride schedule, launch after launch (the last workgroup re-arms the counters), and match an fp32 reference."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(21)
    try:
        for (My, Nx, K) in [(32768, 768, 256), (16384, 2304, 768), (32768, 3072, 128)]:
            y = torch.randn(My, K, generator=g).to(DEV)
            x = (torch.randn(Nx, K, generator=g) / K ** 0.5).to(DEV)
            bias = torch.randn(Nx, generator=g).to(DEV)
            yb, xb = to_elem(_lib.PREC_BF16, y), to_elem(_lib.PREC_BF16, x)
            outs = {}
            for mode in (0, 1, 1, 1):
                _lib.check(L.osud_set_gemm_dynamic_tiles(mode))
                out = torch.empty(My * Nx * 2, dtype=torch.uint8, device=DEV)
                _lib.check(L.osud_op_gemm(_lib.PREC_BF16, _lib.EPI_BIAS_TE, _lib.ptr(yb), K, _lib.ptr(xb), K, My, Nx, K, _lib.ptr(out), Nx,
                                          _lib.ptr(bias), None, 0, 0, 0, None))
                res = from_elem(_lib.PREC_BF16, out, (My, Nx))
                if mode in outs:
                    assert torch.equal(res, outs[mode])
                outs[mode] = res
            assert torch.equal(outs[0], outs[1])
            ref = from_elem(_lib.PREC_BF16, yb, (My, K)) @ from_elem(_lib.PREC_BF16, xb, (Nx, K)).t() + bias
            assert maxdiff(outs[1], ref) <= 1e-2 * float(ref.abs().max())
    finally:
        _lib.check(L.osud_set_gemm_dynamic_tiles(-1))
