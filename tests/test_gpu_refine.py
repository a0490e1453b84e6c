"""GPU: the refine pass of sample.py:186-205 -- after the sampling loop the weights of `--refine-ckpt` are loaded into the SAME model
and the result goes through `refine_iters` calls of `diffusion.p_sample(model.forward_with_cfg, img, t = 0, clip_denoised=True)`.
Here that is one native call (`SpacedDiffusion.p_sample_repeat` -> `osud_sample_repeat`: the captured sampler step of the loops
replayed with its device-side step counter held still).  Fixtures `g15_refine_{tiny,dit_b}` hold the reference's own run."""
import pytest
import torch

from oracle import dit_oracle as mo
from osu_diffusion_amd import _lib
from osu_diffusion_amd.diffusion import create_diffusion
from osu_diffusion_amd.models import DiT
from tests.helpers import T, load, maxdiff, shape_from, weights_for

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# tier -> bound on max |d(x, y)| after 10 iterations (normalised coordinates; north_star: 1e-3 for the tiers that carry the claim)
BOUND = {"fp32": 2e-5, "bf16x3": 1e-3, "fp16f8": 1e-3, "bf16": 3e-2}


def native_model(shape, sd, precision):
    m = DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads, context_size=shape.context,
            num_classes=shape.num_classes, class_dropout_prob=0.2, precision=precision)
    m.load_state_dict(sd, strict=True)
    return m.to(DEV).eval()


def refine_weights(fx):
    sd = mo.seeded_state_dict(shape_from(fx), int(fx["refine_wseed"]))
    wsum = float(sum(v.double().abs().sum() for v in sd.values()))
    assert abs(wsum - float(fx["refine_wsum"])) <= 1e-9 * abs(wsum), "seeded refine weights differ from the fixture's"
    return sd


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16f8", "bf16"])
@pytest.mark.parametrize("tag", ["tiny", "dit_b"])
def test_refine_pass_matches_the_reference(tag, precision):
    """Weights A loaded, then weights B loaded over them (`load_state_dict` on the live model, as sample.py:189-190 does), then the
    fused pass from the fixture's start image: after 1, 5 and 10 iterations within the tier's bound of the reference's run."""
    fx = load(f"g15_refine_{tag}")
    shape, sd_a = weights_for(fx)
    m = native_model(shape, sd_a, precision)
    d = create_diffusion(str(fx["respacing"]), noise_schedule="squaredcos_cap_v2")
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=float(fx["cfg_scale"]), attn_mask=None)
    start = T(fx["start"]).to(DEV)
    with torch.no_grad():
        m.forward_with_cfg(start, torch.zeros(start.shape[0], dtype=torch.long, device=DEV), **kw)  # (weights A have been in use)
        m.load_state_dict(refine_weights(fx))
        worst = 0.0
        for iters in (1, 5, 10):
            got = d.p_sample_repeat(m.forward_with_cfg, start, iters, t=0, clip_denoised=True, model_kwargs=kw)
            worst = max(worst, maxdiff(got.cpu(), fx[f"after_{iters}"]))
        assert torch.equal(start.cpu(), T(fx["start"]))  # the input is left alone
    print(f"MEASURED refine[{tag},{precision}]: max|d| to the reference over 1 / 5 / 10 iterations = {worst:.3e}")
    assert worst < BOUND[precision], worst
    assert maxdiff(fx["after_10"], fx["start"]) > 10 * BOUND["fp32"]  # (the pass does move the image: the test is not vacuous)


def test_refine_pass_replays_the_loop_graph_and_equals_single_steps(osud_option):
    """The fused pass (a) through the captured graph and through eager launches gives the same bits; (b) equals `refine_iters`
    separate `p_sample(t = 0)` calls -- the reference's own loop shape -- up to the summation order of the first linear (the fused
    loop multiplies the offset / context columns once per call, evaluates the timestep MLP once, and folds the guidance combine
    into the sampler kernel)."""
    fx = load("g15_refine_tiny")
    shape, _ = weights_for(fx)
    m = native_model(shape, refine_weights(fx), "fp32")
    d = create_diffusion(str(fx["respacing"]), noise_schedule="squaredcos_cap_v2")
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=float(fx["cfg_scale"]), attn_mask=None)
    start = T(fx["start"]).to(DEV)
    with torch.no_grad():
        fused = d.p_sample_repeat(m.forward_with_cfg, start, 10, model_kwargs=kw)
        osud_option("sample_graph", 0)
        eager = d.p_sample_repeat(m.forward_with_cfg, start, 10, model_kwargs=kw)
        assert torch.equal(fused, eager)
        img = start
        for _ in range(10):
            img = d.p_sample(m.forward_with_cfg, img, torch.zeros(img.shape[0], dtype=torch.long, device=DEV), clip_denoised=True,
                             model_kwargs=kw)["sample"]
        assert maxdiff(img.cpu(), fused.cpu()) < 1e-5
        osud_option("embed_const", 0)
        osud_option("tvec_table", 0)
        plain = d.p_sample_repeat(m.forward_with_cfg, start, 10, model_kwargs=kw)
        assert maxdiff(plain.cpu(), img.cpu()) < 2e-6
    assert maxdiff(fused.cpu(), fx["after_10"]) < BOUND["fp32"]


def test_refine_pass_argument_checks():
    fx = load("g15_refine_tiny")
    shape, sd = weights_for(fx)
    m = native_model(shape, sd, "fp32")
    d = create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    kw = dict(o=T(fx["o"]).to(DEV), c=T(fx["c"]).to(DEV), y=T(fx["y"]).to(DEV), cfg_scale=4.0, attn_mask=None)
    start = T(fx["start"]).to(DEV)
    with torch.no_grad():
        assert torch.equal(d.p_sample_repeat(m.forward_with_cfg, start, 0, model_kwargs=kw), start)  # zero iterations: a copy
        with pytest.raises(AssertionError, match="sample_repeat"):
            d.p_sample_repeat(m.forward_with_cfg, start, 3, t=20, model_kwargs=kw)                   # step outside the schedule
