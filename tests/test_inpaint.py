"""In-painting through `denoised_fn` (testing/test_toy.py:56-74) against the reference's outputs (fixture g11_inpaint):
CPU — the oracle and the generic Python path of the diffusion object; GPU — the fused native step and loop, which apply
the mask inside the sampler-update kernel (`osud_sampler_step_inpaint`, `osud_sample_loop_inpaint`)."""
import os

import pytest
import torch

from oracle import diffusion_oracle as do
from oracle import dit_oracle as mo
from osu_diffusion_amd import _lib
from osu_diffusion_amd.diffusion import InPaintMask, create_diffusion
from tests.helpers import T, load, maxdiff, weights_for

FX = load("g11_inpaint")
DEV = "cuda:0"


def step_inputs():
    return (T(FX[k]) for k in ("x", "t", "model_out", "known", "mask"))


def test_oracle_steps_match_reference():
    x, t, mout, known, mask = step_inputs()
    sch = do.create_schedule("250", "squaredcos_cap_v2")
    fn = InPaintMask(mask, known)
    o = do.p_sample_step(sch, mout, x, t, T(FX["p_noise"]), denoised_fn=fn)
    assert maxdiff(o["sample"], FX["p_sample"]) == 0.0 and maxdiff(o["pred_xstart"], FX["p_x0"]) == 0.0
    o = do.ddim_step(sch, mout, x, t, T(FX["ddim_noise"]), eta=0.5, denoised_fn=fn)
    assert maxdiff(o["sample"], FX["ddim_sample"]) == 0.0 and maxdiff(o["pred_xstart"], FX["ddim_x0"]) == 0.0
    # the mask acts before the clamp (gaussian_diffusion.py:341-346)
    forced = ~mask
    assert torch.equal(o["pred_xstart"][forced], known[forced].clamp(-1, 2))


def test_generic_python_path_matches_reference_on_cpu():
    x, t, mout, known, mask = step_inputs()
    d = create_diffusion("250", noise_schedule="squaredcos_cap_v2")
    model = lambda *_a, **_k: mout  # noqa: E731
    for closure in (InPaintMask(mask, known), lambda v: torch.where(mask, v, known)):
        torch.manual_seed(78)
        r = d.p_sample(model, x, t, clip_denoised=True, denoised_fn=closure)
        assert maxdiff(r["sample"], FX["p_sample"]) == 0.0 and maxdiff(r["pred_xstart"], FX["p_x0"]) == 0.0
        torch.manual_seed(78)
        r = d.ddim_sample(model, x, t, clip_denoised=True, denoised_fn=closure, eta=0.5)
        assert maxdiff(r["sample"], FX["ddim_sample"]) == 0.0


def test_oracle_loop_matches_reference():
    shape, sd = weights_for(FX)
    sch = do.create_schedule("20", "squaredcos_cap_v2")
    o, c, y = T(FX["loop_o"]), T(FX["loop_c"]), T(FX["loop_y"])
    fn = InPaintMask(T(FX["loop_mask"]), T(FX["loop_x0"]))
    got = do.sample_loop(sch, lambda xx, tt: mo.forward(sd, shape, xx, tt, o, c, y), T(FX["loop_z"]), T(FX["loop_noises"]),
                         denoised_fn=fn)
    assert maxdiff(got, FX["loop_final"]) < 5e-6


def test_library_rejects_half_an_inpaint_request():
    import ctypes as C

    Half = _lib.InPaintStruct
    d = create_diffusion("250", noise_schedule="squaredcos_cap_v2")
    buf = torch.zeros(2 * 2 * 8)
    half = Half(_lib.ptr(buf), None)
    ti = torch.zeros(2, dtype=torch.long)
    rc = _lib.lib().osud_sampler_step_inpaint(d._sched.handle, 0, 0.0, _lib.ptr(buf), _lib.ptr(buf), _lib.ptr(ti), _lib.ptr(buf),
                                              2, 8, -1.0, 1, C.cast(C.pointer(half), C.c_void_p), _lib.ptr(buf), None, None)
    assert rc != 0 and "in-painting" in _lib.last_error()


@pytest.mark.gpu
def test_native_step_with_inpaint_mask():
    x, t, mout, known, mask = (v.to(DEV) for v in step_inputs())
    d = create_diffusion("250", noise_schedule="squaredcos_cap_v2")
    L = _lib.lib()
    N, _, TT = x.shape
    ip, keep = InPaintMask(mask, known).native(x)
    for mode, eta, key in ((0, 0.0, "p"), (1, 0.5, "ddim")):
        out, x0 = torch.empty_like(x), torch.empty_like(x)
        nz = T(FX[key + "_noise"]).to(DEV)
        _lib.check(L.osud_sampler_step_inpaint(d._sched.handle, mode, eta, _lib.ptr(mout), _lib.ptr(x), _lib.ptr(t), _lib.ptr(nz),
                                               N, TT, -1.0, 1, ip, _lib.ptr(out), _lib.ptr(x0), None))
        assert maxdiff(x0.cpu(), FX[key + "_x0"]) == 0.0, key            # bit-exact: selection + clamp only
        assert maxdiff(out.cpu(), FX[key + "_sample"]) < 1e-6, key       # expf / sqrtf last ulp
    del keep


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16x3", 1e-3), ("fp16f8", 1e-3), ("bf16", 2.1e-3)])  # measured 1.9e-5 / (bf16x3: printed) / 7.0e-4 (bf16 bound = 3x)
def test_native_loop_with_inpaint_mask(precision, tol):
    """test_toy.py's use: every object of the window given except the last; 20 fused steps with `model.forward`."""
    from osu_diffusion_amd.models import DiT

    shape, sd = weights_for(FX)
    m = DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads, context_size=shape.context,
            num_classes=shape.num_classes, precision=precision)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    d = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    kw = dict(o=T(FX["loop_o"]).to(DEV), c=T(FX["loop_c"]).to(DEV), y=T(FX["loop_y"]).to(DEV), attn_mask=None)
    mask, x0 = T(FX["loop_mask"]).to(DEV), T(FX["loop_x0"]).to(DEV)
    fn = InPaintMask(mask, x0)
    z = T(FX["loop_z"]).to(DEV)
    finals = {}
    for graph in ("graph", "eager"):
        _lib.set_option("sample_graph", 1 if graph == "graph" else 0)
        try:
            with torch.no_grad():
                finals[graph] = d.p_sample_loop(m.forward, z.shape, z, denoised_fn=fn, clip_denoised=True, model_kwargs=kw,
                                                step_noise=T(FX["loop_noises"])).cpu()
        finally:
            _lib.set_option("sample_graph", -1)
    assert torch.equal(finals["graph"], finals["eager"])
    fin = finals["graph"]
    print(f"MEASURED inpaint_loop[{precision}]: final max|d| vs reference {maxdiff(fin, FX['loop_final']):.3e}")
    assert maxdiff(fin, FX["loop_final"]) < tol
    given = ~T(FX["loop_mask"])
    assert torch.equal(fin[given], T(FX["loop_x0"])[given].clamp(-1, 2))   # the given coordinates come out untouched
    # a plain closure takes the generic Python route around the native forward and lands on the same result
    if precision == "fp32":
        torch.manual_seed(0)
        with torch.no_grad():
            ref_route = d.p_sample_loop(m.forward, z.shape, z, denoised_fn=lambda v: torch.where(mask, v, x0), model_kwargs=kw,
                                        device=DEV)
        assert ref_route.shape == fin.shape and torch.equal(ref_route.cpu()[given], fin[given])
