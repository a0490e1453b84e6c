"""CPU: the oracle (oracle/*.py) against the golden vectors frozen from the reference
(tests/golden/make_golden.py).  This is what pins the oracle on machines without
/root/reference."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import diffusion_oracle as do
from oracle import dit_oracle as mo
from tests.helpers import GOLDEN, T, load, maxdiff, weights_for

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))

SCHED_TABLES = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
                "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
                "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                "posterior_mean_coef1", "posterior_mean_coef2"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g1_schedule_*.npz"))))
def test_schedule_tables_bit_exact(path):
    fx = np.load(path)
    sch = do.create_schedule(str(fx["respacing"]), str(fx["noise_schedule"]))
    for n in SCHED_TABLES:
        assert np.array_equal(getattr(sch, n), fx[n]), n
    assert np.array_equal(sch.timestep_map, fx["timestep_map"])


def test_known_answer_constants():
    """SURVEY.md Appendix B spot values."""
    s = do.create_schedule("1000", "squaredcos_cap_v2")
    assert s.betas[0] == 4.128422482196914e-05 and s.betas[999] == 0.999
    assert s.sqrt_recip_alphas_cumprod[999] == 20291.169634661146
    assert s.posterior_log_variance_clipped[0] == s.posterior_log_variance_clipped[1] == -10.734082532465003
    s250 = do.create_schedule("250", "squaredcos_cap_v2")
    assert list(s250.timestep_map[:6]) == [0, 4, 8, 12, 16, 20] and s250.timestep_map[-1] == 999
    assert s250.betas[-1] == 0.9999374992411221
    lin = do.create_schedule("", "linear")
    assert abs(lin.betas[0] - 1e-4) < 1e-15 and abs(lin.betas[-1] - 0.02) < 1e-15  # re-derived: 1 - ac/ac_prev
    assert do.space_timesteps(1000, "ddim50")[:3] == [0, 20, 40]
    with pytest.raises(ValueError):
        do.space_timesteps(10, "11")


def test_embeddings_bit_exact():
    fx = load("g2_embeddings")
    v = T(fx["v"])
    assert maxdiff(mo.sincos_embedding(v, 128), fx["emb128"]) == 0.0
    assert maxdiff(mo.sincos_embedding(v, 256), fx["emb256"]) == 0.0


@pytest.mark.parametrize("tag", ["tiny_T64", "tiny_T128", "tiny_T200_band", "tiny_T128_allfalse", "small_T128",
                                 "tiny_T128_rough", "dit_b_T128", "dit_b_T128_rough"])
def test_forward_and_cfg(tag):
    fx = load("g3_forward_" + tag)
    shape, sd = weights_for(fx)
    mask = T(fx["attn_mask"]) if "attn_mask" in fx else None
    args = (T(fx["x"]), T(fx["t"]), T(fx["o"]), T(fx["c"]), T(fx["y"]))
    with torch.no_grad():
        out = mo.forward(sd, shape, *args, attn_mask=mask)
        cfg4 = mo.forward_with_cfg(sd, shape, *args, 4.0, attn_mask=mask)
        cfg1 = mo.forward_with_cfg(sd, shape, *args, 1.0, attn_mask=mask)
    tol = (1e-4 if tag.startswith("dit_b") else 5e-5) if tag.endswith("rough") else 2e-5
    assert maxdiff(out, fx["out"]) < tol
    assert maxdiff(cfg4, fx["out_cfg4"]) < 5 * tol
    assert maxdiff(cfg1, fx["out_cfg1"]) < tol


@pytest.mark.parametrize("tag", ["1000", "250"])
def test_sampler_steps_bit_exact(tag):
    fx = load("g5_step_" + tag)
    sch = do.create_schedule(tag, "squaredcos_cap_v2")
    x, t, mout = T(fx["x"]), T(fx["t"]), T(fx["model_out"])
    r = do.p_sample_step(sch, mout, x, t, T(fx["p_noise"]))
    assert maxdiff(r["sample"], fx["p_sample"]) == 0.0 and maxdiff(r["pred_xstart"], fx["p_x0"]) == 0.0
    for eta, k in [(0.0, "ddim0"), (1.0, "ddim1")]:
        r = do.ddim_step(sch, mout, x, t, T(fx[k + "_noise"]), eta=eta)
        assert maxdiff(r["sample"], fx[k + "_sample"]) == 0.0
    r = do.p_sample_step(sch, mout, x, t, T(fx["p_noise"]), clip_denoised=False)
    assert maxdiff(r["sample"], fx["p_noclip_sample"]) == 0.0


@pytest.mark.parametrize("tag", ["p20", "ddim20_eta1", "ddim20_eta05", "p250"])
def test_chained_loop(tag):
    fx = load("g6_loop_" + tag)
    shape, sd = weights_for(fx)
    sch = do.create_schedule(str(fx["respacing"]), "squaredcos_cap_v2")
    o, c, y = T(fx["o"]), T(fx["c"]), T(fx["y"])
    fn = lambda xx, tt: mo.forward_with_cfg(sd, shape, xx, tt, o, c, y, 4.0)  # noqa: E731
    eta = float(fx["eta"])
    final = do.sample_loop(sch, fn, T(fx["z"]), T(fx["noises"]), ddim_eta=None if eta < 0 else eta)
    assert maxdiff(final, fx["final"]) < 1e-3  # north_star tolerance on final coordinates


@pytest.mark.parametrize("damping", ["", "_undamped"])
def test_1000_step_head_and_tail(damping):
    fx = load("g6_steps_1000" + damping)
    shape = mo.DitShape(*(int(v) for v in fx["shape"][:3]), num_classes=int(fx["shape"][3]))
    sd = mo.seeded_state_dict(shape, int(fx["wseed"]), pos_gain=float(fx["pos_gain"]))
    sch = do.create_schedule("1000", "squaredcos_cap_v2")
    o, c, y = T(fx["o"]), T(fx["c"]), T(fx["y"])
    tm = torch.from_numpy(sch.timestep_map)
    for part in ("head", "tail"):
        x = T(fx[part + "_start"]).clone()
        with torch.no_grad():
            for k, i in enumerate(range(int(fx[part + "_first"]), int(fx[part + "_last"]) - 1, -1)):
                tt = torch.tensor([i] * len(x))
                x = do.p_sample_step(sch, mo.forward_with_cfg(sd, shape, x, tm[tt], o, c, y, 4.0), x, tt, T(fx[part + "_noises"])[k])["sample"]
        assert maxdiff(x, fx[part + "_final"]) < 1e-3, part


def test_training_at_dit_b_width():
    """fixture g7_train_dit_b: per-tensor norm, random projection and strided sample of every gradient (make_golden.grad_probe)."""
    fx = load("g7_train_dit_b")
    shape, sd = weights_for(fx)
    sd = {k: v.clone().requires_grad_(k != "xoc_embedder.playfield_size") for k, v in sd.items()}
    sch = do.create_schedule("", "squaredcos_cap_v2")
    o, c, y, drop = T(fx["o"]), T(fx["c"]), T(fx["y"]), T(fx["drop"])
    terms = do.training_losses(sch, lambda xx, tt: mo.forward(sd, shape, xx, tt, o, c, y, drop_mask=drop), T(fx["x"]), T(fx["t"]),
                               T(fx["noise"]), loss="l1")
    terms["loss"].mean().backward()
    assert maxdiff(terms["loss"].detach(), fx["loss"]) < 2e-5 and maxdiff(terms["vb"].detach(), fx["vb"]) < 2e-5
    for k, n in zip((str(s) for s in fx["grad_keys"]), fx["grad_norms"]):
        g = sd[k].grad.double().flatten()
        assert abs(float(g.norm()) - n) <= 1e-4 * max(n, 1e-3), k
        stride = max(1, g.numel() // 4096)
        assert maxdiff(g[::stride][:4096].float(), fx["sample:" + k]) < 1e-5, k


@pytest.mark.parametrize("tag", ["trim250"])  # "full100" (757 objects x 100 steps) is checked when the fixture is made: too slow here
def test_cli_sampling_flow_on_the_toy_beatmap(tag):
    """fixture g12_cli_toy (the reference's sample.py flow, --seed 0, cfg 4, banded mask): the oracle with noise drawn from the
    CPU generator in the reference's order -- randn(n, 2, T) once, then randn_like(x) per step."""
    from osu_diffusion_amd.synthetic import banded_attn_mask

    fx = load("g12_cli_toy")
    shape, sd = weights_for(fx)
    Tn, steps = int(fx[tag + ":T"]), int(fx[tag + ":steps"])
    o, c = T(fx[tag + ":o"]).repeat(2, 1), T(fx[tag + ":c"]).repeat(2, 1, 1)
    y = torch.tensor([int(fx["label"]), shape.num_classes])
    mask = banded_attn_mask(Tn, 128)
    sch = do.create_schedule(str(steps), "squaredcos_cap_v2")
    torch.manual_seed(int(fx["seed"]))
    from osu_diffusion_amd.models import DiT_models  # sample.py:69-76 constructs the model after seeding: its init advances the generator
    DiT_models["DiT-S"](num_classes=shape.num_classes, context_size=144)
    z = torch.randn(1, 2, Tn)
    z = torch.cat([z, z])
    noises = torch.stack([torch.randn_like(z) for _ in range(steps)])
    fn = lambda xx, tt: mo.forward_with_cfg(sd, shape, xx, tt, o, c, y, float(fx["cfg_scale"]), attn_mask=mask)  # noqa: E731
    final = do.sample_loop(sch, fn, z, noises)
    assert maxdiff(final[:1], fx[tag + ":final"]) < 1e-3


@pytest.mark.parametrize("loss", ["l1", "mse"])
def test_training_losses_and_grads(loss):
    fx = load("g7_train_" + loss)
    shape, sd = weights_for(fx)
    sd = {k: v.clone().requires_grad_(k != "xoc_embedder.playfield_size") for k, v in sd.items()}
    sch = do.create_schedule("", "squaredcos_cap_v2")
    o, c, y, drop = T(fx["o"]), T(fx["c"]), T(fx["y"]), T(fx["drop"])
    fn = lambda xx, tt: mo.forward(sd, shape, xx, tt, o, c, y, drop_mask=drop)  # noqa: E731
    terms = do.training_losses(sch, fn, T(fx["x"]), T(fx["t"]), T(fx["noise"]), loss=loss)
    terms["loss"].mean().backward()
    assert maxdiff(terms["loss"].detach(), fx["loss"]) < 2e-5
    assert maxdiff(terms["vb"].detach(), fx["vb"]) < 2e-5
    assert maxdiff(terms[loss].detach(), fx["main"]) < 2e-5
    for k in fx:
        if k.startswith("grad:"):
            assert maxdiff(sd[k[5:]].grad, fx[k]) < 1e-5, k
    norms = dict(zip((str(s) for s in fx["grad_keys"]), fx["grad_norms"]))
    for k, n in norms.items():
        assert abs(float(sd[k].grad.double().norm()) - n) <= 1e-4 * max(n, 1e-3), k


def test_registry_keys_match_reference():
    fx = load("g9_registry_dit_s")
    shp = mo.param_shapes(mo.shape_of("DiT-S", num_classes=10))
    assert [str(k) for k in fx["keys"]] == list(shp.keys())
    assert [str(s) for s in fx["shapes"]] == [str(tuple(v)) for v in shp.values()]


def test_p1000_dit_b_fixture_is_self_consistent():
    """g6_loop_p1000_dit_b (the reference's 1000-step CFG-4 loop at DiT-B's geometry): seeded weights match the pinned checksum, the
    per-step noise redrawn from the stored seed matches the stored checksum and samples (what the GPU test feeds the native loop),
    the oracle's own run of the loop was within 5e-4 of the reference when the fixture was made, and the recorded states are sane.
    (The loop itself -- 7 minutes of CPU in the oracle -- is replayed on the GPU: tests/test_gpu_x3.py.)"""
    fx = load("g6_loop_p1000_dit_b")
    shape, _ = weights_for(fx)
    assert (shape.depth, shape.hidden, shape.heads) == (12, 768, 12) and str(fx["respacing"]) == "1000" and float(fx["cfg_scale"]) == 4.0
    z = T(fx["z"])
    assert z.shape == (4, 2, 128) and torch.equal(z[:2], z[2:])  # [cond; uncond] start from the same noise (sample.py:97-100)
    torch.manual_seed(int(fx["noise_seed"]))
    noises = torch.stack([torch.randn_like(z) for _ in range(1000)])
    assert abs(float(noises.double().sum()) - float(fx["noise_sum"])) < 1e-6
    assert abs(float(noises.double().abs().sum()) - float(fx["noise_abs_sum"])) < 1e-5
    assert torch.equal(noises[0, 0, 0, :8], T(fx["noise_head"])) and torch.equal(noises[-1, -1, -1, -8:], T(fx["noise_tail"]))
    assert float(fx["oracle_fp32_vs_reference"]) < 5e-4
    final = T(fx["final"])
    assert torch.isfinite(final).all() and float(final.min()) >= -1.0 and float(final.max()) <= 2.0  # clamp of the last step (x0 range)
    assert maxdiff(final, fx["final_fp64"]) < 1e-3  # how far the reference's fp32 arithmetic itself is from exact: 3.5e-4
    for k in (250, 500, 750):
        assert T(fx[f"after_{k}"]).shape == final.shape
