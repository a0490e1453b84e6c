#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE implementation.

Run in the build container only (needs /root/reference; the GPU box has no copy):

    python tests/golden/make_golden.py

It imports the reference's ``models`` / ``diffusion`` / ``positional_embedding`` modules,
(1) asserts the CPU oracle (oracle/*.py) matches them on the same inputs and
(2) freezes inputs + expected outputs as small ``.npz`` files.  Only data is written —
no reference source or bytecode.  Weights are not stored: they are rebuilt from
``oracle.dit_oracle.seeded_state_dict(shape, seed)`` and pinned by a checksum.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = os.environ.get("OSUD_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import models as ref_models  # noqa: E402  (reference)
import positional_embedding as ref_pe  # noqa: E402
from diffusion import create_diffusion as ref_create_diffusion  # noqa: E402

from oracle import diffusion_oracle as do  # noqa: E402
from oracle import dit_oracle as mo  # noqa: E402
from osu_diffusion_amd.synthetic import banded_attn_mask, synthetic_windows  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)

TINY = mo.DitShape(depth=2, hidden=128, heads=2, num_classes=10)
SMALL = mo.shape_of("DiT-S", num_classes=10)


def ref_model_for(shape: mo.DitShape, sd, dropout=0.2):
    m = ref_models.DiT(depth=shape.depth, hidden_size=shape.hidden, num_heads=shape.heads,
                       context_size=shape.context, num_classes=shape.num_classes,
                       class_dropout_prob=dropout)
    m.load_state_dict(sd, strict=True)
    return m.eval()


def checksum(sd) -> float:
    return float(sum(v.double().abs().sum() for v in sd.values()))


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        out[k] = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KiB)")


def close(a, b, tol, what):
    err = (a.double() - b.double()).abs().max().item()
    print(f"  oracle vs reference [{what}]: max|d| = {err:.3e}")
    assert err <= tol, (what, err)


# ------------------------------------------------------------------ G1 schedules
def g1_schedules():
    print("G1 schedules")
    names = ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
             "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
             "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
             "posterior_mean_coef1", "posterior_mean_coef2"]
    for tag, resp, sched in [("1000_cos", "1000", "squaredcos_cap_v2"), ("250_cos", "250", "squaredcos_cap_v2"),
                             ("ddim50_cos", "ddim50", "squaredcos_cap_v2"), ("train_linear", "", "linear"),
                             ("20_cos", "20", "squaredcos_cap_v2"), ("train_cos", "", "squaredcos_cap_v2")]:
        ref = ref_create_diffusion(resp, noise_schedule=sched)
        ora = do.create_schedule(resp, sched)
        arrs = {}
        for n in names:
            r = getattr(ref, n)
            assert np.array_equal(r, getattr(ora, n)), (tag, n)
            arrs[n] = r
        assert list(ora.timestep_map) == list(ref.timestep_map)
        arrs["timestep_map"] = np.array(ref.timestep_map, dtype=np.int64)
        save(f"g1_schedule_{tag}", respacing=resp, noise_schedule=sched, **arrs)


# ------------------------------------------------------------------ G2 embeddings
def g2_embeddings():
    print("G2 embeddings")
    v = torch.tensor([0.0, 1.0, 0.5, 511.99, 383.5, 999.0, 500.0, 14000.25, 1.4e5 / 10, 3.14159], dtype=torch.float32)
    e128 = ref_pe.timestep_embedding(v, 128)
    e256 = ref_pe.timestep_embedding(v, 256)
    close(mo.sincos_embedding(v, 128), e128, 0.0, "emb128")
    close(mo.sincos_embedding(v, 256), e256, 0.0, "emb256")
    save("g2_embeddings", v=v, emb128=e128, emb256=e256)


# ------------------------------------------------------------------ G3/G4 forward
def inputs(n, T, shape, seed, ts, null_last=True):
    (x, o, c), y = synthetic_windows(n, T, shape.num_classes, seed=seed, train_offsets=True)
    if null_last:
        y[-1] = shape.num_classes
    t = torch.tensor([ts[i % len(ts)] for i in range(n)], dtype=torch.long)
    return x, t, o, c, y


def g3_forward():
    print("G3/G4 forward")
    for tag, shape, wseed, n, T, mask in [
        ("tiny_T64", TINY, 11, 4, 64, None),
        ("tiny_T128", TINY, 11, 4, 128, None),
        ("tiny_T200_band", TINY, 11, 2, 200, banded_attn_mask(200, 128)),
        ("tiny_T128_allfalse", TINY, 11, 2, 128, torch.zeros(128, 128, dtype=torch.bool)),
        ("small_T128", SMALL, 12, 2, 128, None),
        ("tiny_T128_rough", TINY, 13, 2, 128, None),
    ]:
        rough = tag.endswith("rough")
        sd = mo.seeded_state_dict(shape, wseed, pos_gain=1.0, mod_std=0.2) if rough else mo.seeded_state_dict(shape, wseed)
        ref = ref_model_for(shape, sd)
        x, t, o, c, y = inputs(n, T, shape, seed=100 + T, ts=[0, 1, 500, 999])
        out_ref = ref(x, t, o, c, y, attn_mask=mask)
        out_ora = mo.forward(sd, shape, x, t, o, c, y, attn_mask=mask)
        close(out_ora, out_ref, 5e-5 if rough else 2e-5, tag)
        cfg4 = ref.forward_with_cfg(x, t, o, c, y, 4.0, attn_mask=mask)
        cfg1 = ref.forward_with_cfg(x, t, o, c, y, 1.0, attn_mask=mask)
        close(mo.forward_with_cfg(sd, shape, x, t, o, c, y, 4.0, attn_mask=mask), cfg4, 1e-4, tag + " cfg4")
        extra = {} if mask is None else {"attn_mask": mask}
        save(f"g3_forward_{tag}", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]),
             wseed=wseed, wsum=checksum(sd), rough=rough, x=x, t=t, o=o, c=c, y=y, out=out_ref, out_cfg4=cfg4,
             out_cfg1=cfg1, **extra)


# ------------------------------------------------------------------ G5 sampler step
def g5_step():
    print("G5 sampler steps")
    g = torch.Generator().manual_seed(5)
    N, T = 6, 64
    for tag, resp in [("1000", "1000"), ("250", "250")]:
        ref = ref_create_diffusion(resp, noise_schedule="squaredcos_cap_v2")
        ora = do.create_schedule(resp, "squaredcos_cap_v2")
        nt = ref.num_timesteps
        t = torch.tensor([0, 1, nt // 2, nt - 1, 0, nt - 2], dtype=torch.long)
        x = torch.randn(N, 2, T, generator=g) * torch.tensor([0.3, 1.0, 1.0, 1.0, 3.0, 1.0]).view(-1, 1, 1) + 0.5
        mout = torch.randn(N, 4, T, generator=g)
        mout[:, 2:] = mout[:, 2:].clamp(-1.5, 1.5)
        noise = torch.randn(N, 2, T, generator=g)
        model = lambda *_a, **_k: mout  # noqa: E731

        outs = {}
        for name, fn, kw in [("p", ref.p_sample, {}), ("ddim0", ref.ddim_sample, {"eta": 0.0}),
                             ("ddim1", ref.ddim_sample, {"eta": 1.0})]:
            torch.manual_seed(77)
            r = fn(model, x, t, clip_denoised=True, **kw)
            torch.manual_seed(77)
            nz = torch.randn_like(x)
            if name == "p":
                o = do.p_sample_step(ora, mout, x, t, nz)
            else:
                o = do.ddim_step(ora, mout, x, t, nz, eta=kw["eta"])
            close(o["sample"], r["sample"], 0.0, f"{tag} {name} sample")
            close(o["pred_xstart"], r["pred_xstart"], 0.0, f"{tag} {name} x0")
            outs[name + "_sample"] = r["sample"]
            outs[name + "_x0"] = r["pred_xstart"]
            outs[name + "_noise"] = nz
        torch.manual_seed(77)
        r = ref.p_sample(model, x, t, clip_denoised=False)
        outs["p_noclip_sample"] = r["sample"]
        save(f"g5_step_{tag}", x=x, t=t, model_out=mout, **outs)


# ------------------------------------------------------------------ G6 chained loop
def g6_loop():
    print("G6 chained sampling loops")
    shape, wseed = TINY, 11
    sd = mo.seeded_state_dict(shape, wseed)
    ref = ref_model_for(shape, sd)
    n, T = 2, 64
    (x0, o, c), y = synthetic_windows(n, T, shape.num_classes, seed=7, train_offsets=False)
    o = torch.cat([o, o])
    c = torch.cat([c, c])
    y = torch.cat([y, torch.full_like(y, shape.num_classes)])
    for tag, resp, ddim in [("p20", "20", None), ("ddim20_eta1", "20", 1.0), ("ddim20_eta05", "20", 0.5)]:
        dref = ref_create_diffusion(resp, noise_schedule="squaredcos_cap_v2")
        ora = do.create_schedule(resp, "squaredcos_cap_v2")
        torch.manual_seed(3)
        z = torch.randn(n, 2, T)
        z = torch.cat([z, z])
        kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=None)
        torch.manual_seed(4)
        if ddim is None:
            final = dref.p_sample_loop(ref.forward_with_cfg, z.shape, z, clip_denoised=True, model_kwargs=kw, device="cpu")
        else:
            final = dref.ddim_sample_loop(ref.forward_with_cfg, z.shape, z, clip_denoised=True, model_kwargs=kw,
                                          device="cpu", eta=ddim)
        torch.manual_seed(4)
        noises = torch.stack([torch.randn_like(z) for _ in range(ora.num_timesteps)])
        fn = lambda xx, tt: mo.forward_with_cfg(sd, shape, xx, tt, o, c, y, 4.0)  # noqa: E731
        mine = do.sample_loop(ora, fn, z, noises, ddim_eta=ddim)
        close(mine, final, 5e-4, tag)
        save(f"g6_loop_{tag}", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]),
             wseed=wseed, wsum=checksum(sd), z=z, o=o, c=c, y=y, noises=noises, final=final, respacing=resp,
             eta=-1.0 if ddim is None else ddim)


# ------------------------------------------------------------------ G7 training
def g7_training():
    print("G7 training losses + grads + AdamW/EMA step")
    torch.set_grad_enabled(True)
    shape, wseed = TINY, 11
    B, T = 4, 64
    (x, o, c), y = synthetic_windows(B, T, shape.num_classes, seed=21, train_offsets=True)
    t = torch.tensor([0, 3, 500, 999], dtype=torch.long)
    g = torch.Generator().manual_seed(22)
    noise = torch.randn(B, 2, T, generator=g)
    drop = torch.tensor([False, True, False, False])
    y_eff = torch.where(drop, torch.full_like(y, shape.num_classes), y)
    for loss_name, use_l1 in [("l1", True), ("mse", False)]:
        sd = mo.seeded_state_dict(shape, wseed)
        ref = ref_model_for(shape, sd)  # eval(): labels pre-dropped == train-mode dropout with that mask
        for p in ref.parameters():
            p.grad = None
        dref = ref_create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=use_l1)
        terms = dref.training_losses(ref, x, t, dict(o=o, c=c, y=y_eff), noise=noise)
        loss = terms["loss"].mean()
        loss.backward()
        grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
        # oracle
        osd = {k: v.clone().requires_grad_(k != "xoc_embedder.playfield_size") for k, v in sd.items()}
        ora = do.create_schedule("", "squaredcos_cap_v2")
        fn = lambda xx, tt: mo.forward(osd, shape, xx, tt, o, c, y, drop_mask=drop)  # noqa: E731
        ot = do.training_losses(ora, fn, x, t, noise, loss=loss_name)
        ot["loss"].mean().backward()
        for k in ("loss", "vb", loss_name):
            close(ot[k].detach(), terms[k].detach(), 2e-5, f"{loss_name} term {k}")
        worst = max((osd[k].grad - grads[k]).abs().max().item() for k in grads)
        print(f"  oracle vs reference [{loss_name} grads]: max|d| = {worst:.3e}")
        assert worst < 1e-4
        keep = ["xoc_embedder.mlp.0.weight", "t_embedder.mlp.2.bias", "y_embedder.embedding_table.weight",
                "blocks.0.attn.in_proj_weight", "blocks.0.attn.in_proj_bias", "blocks.1.attn.out_proj.weight",
                "blocks.0.mlp.fc1.weight", "blocks.1.mlp.fc2.bias", "blocks.1.adaLN_modulation.1.weight",
                "final_layer.linear.weight", "final_layer.adaLN_modulation.1.bias"]
        gsave = {"grad:" + k: grads[k] for k in keep}
        gnorm = {k: float(v.double().norm()) for k, v in grads.items()}
        # one AdamW + EMA step (train.py:161,258-261; update_ema :36-45)
        import copy
        ema = copy.deepcopy(ref)
        opt = torch.optim.AdamW(ref.parameters(), lr=1e-4, weight_decay=0)
        opt.step()
        with torch.no_grad():
            for (k, pe), (_, pm) in zip(ema.named_parameters(), ref.named_parameters()):
                pe.mul_(0.9999).add_(pm.data, alpha=1 - 0.9999)
        after = {"after:" + k: dict(ref.named_parameters())[k].detach() for k in keep}
        ema_after = {"ema:" + k: dict(ema.named_parameters())[k].detach() for k in keep[:3]}
        save(f"g7_train_{loss_name}", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]),
             wseed=wseed, wsum=checksum(sd), x=x, o=o, c=c, y=y, t=t, noise=noise, drop=drop,
             loss=terms["loss"].detach(), vb=terms["vb"].detach(), main=terms[loss_name].detach(),
             grad_keys=np.array(sorted(gnorm)), grad_norms=np.array([gnorm[k] for k in sorted(gnorm)]),
             **gsave, **after, **ema_after)
    torch.set_grad_enabled(False)


# ------------------------------------------------------------------ G9 windows / init
def g9_init_and_keys():
    print("G9 registry: state-dict keys, shapes, init statistics")
    torch.manual_seed(0)
    m = ref_models.DiT_models["DiT-S"](num_classes=10, context_size=144, class_dropout_prob=0.2)
    keys = [k for k, _ in m.named_parameters()]
    shapes = [tuple(p.shape) for _, p in m.named_parameters()]
    assert keys == list(mo.param_shapes(SMALL).keys()), "oracle key order differs from reference"
    assert shapes == list(mo.param_shapes(SMALL).values())
    sd = m.state_dict()
    probe = {k: sd[k].flatten()[:8].clone() for k in ["xoc_embedder.mlp.0.weight", "blocks.0.attn.in_proj_weight",
                                                      "blocks.3.mlp.fc1.weight", "y_embedder.embedding_table.weight",
                                                      "t_embedder.mlp.2.weight", "blocks.11.attn.out_proj.weight"]}
    save("g9_registry_dit_s", keys=np.array(keys), shapes=np.array([str(s) for s in shapes]),
         state_keys=np.array(list(sd.keys())), **{"probe:" + k: v for k, v in probe.items()})


# ------------------------------------------------------------------ G9 windows (loader contract)
def g9_windows():
    """data_loading.py imports the third-party `slider` package at module scope (absent here): stub modules let the
    reference's pure-tensor window code and its iterables run; `.osu` parsing itself is NOT exercised (out of scope)."""
    import random
    import types
    print("G9 windows: window contract of the loader (slider stubbed)")
    for name in ("slider", "slider.beatmap", "slider.curve"):
        sys.modules.setdefault(name, types.ModuleType(name))
    setattr(sys.modules["slider"], "Position", type("Position", (), {}))
    for cls in ("Beatmap", "Circle", "Slider", "Spinner", "HitObject", "HoldNote"):
        setattr(sys.modules["slider.beatmap"], cls, type(cls, (), {}))
    for cls in ("Linear", "Catmull", "Perfect", "MultiBezier"):
        setattr(sys.modules["slider.curve"], cls, type(cls, (), {}))
    import data_loading as ref_dl  # noqa: E402  (reference)
    from osu_diffusion_amd import windows as W

    ref_dl.Beatmap.from_path = staticmethod(lambda path: path)  # the iterable only hands the result to seq_func
    sources = W.synthetic_sequences(8, min_len=100, max_len=700, seed=5)
    by_name = dict(sources)
    # per-sequence maths
    seq = sources[3][1]
    close(W.calc_distances(seq.clone()), ref_dl.calc_distances(seq.clone()), 0.0, "calc_distances")
    (rx, ro, rc), rl = ref_dl.split_and_process_sequence_no_augment(seq.clone())
    (mx, mo_, mc), ml = W.split_and_process_sequence_no_augment(seq.clone())
    assert rl == ml
    close(mx, rx, 0.0, "no_augment x"); close(mo_, ro, 0.0, "no_augment o"); close(mc, rc, 0.0, "no_augment c")

    def run(make_iter, n_max):
        random.seed(2024)
        out = []
        for (x, o, c), y in make_iter():
            out.append((x.clone(), o.clone(), c.clone(), int(y)))
            if len(out) == n_max:
                break
        return out

    def ref_iter(files, seq_len, stride):
        return ref_dl.BeatmapDatasetIterable(files, seq_len, stride, lambda p: ref_dl.split_and_process_sequence(by_name[p].clone()),
                                             ref_dl.window_and_relative_time)

    cases = {}
    names = [n for n, _ in sources]
    for tag, seq_len, stride, cycle in (("plain", 128, 16, 1), ("inter", 64, 32, 3)):
        if cycle == 1:
            ref_out = run(lambda: ref_iter(names, seq_len, stride), 200)
            my_out = run(lambda: W.WindowIterable(sources, seq_len, stride, lambda t: W.split_and_process_sequence(t.clone())), 200)
        else:
            ref_out = run(lambda: ref_dl.InterleavingBeatmapDatasetIterable(names, lambda f: ref_iter(f, seq_len, stride), cycle), 200)
            my_out = run(lambda: W.InterleavingIterable(
                sources, lambda srcs: W.WindowIterable(srcs, seq_len, stride, lambda t: W.split_and_process_sequence(t.clone())), cycle), 200)
        assert len(ref_out) == len(my_out) and len(ref_out) > 10, (tag, len(ref_out), len(my_out))
        for (a, b) in zip(ref_out, my_out):
            assert a[3] == b[3]
            for u, v in zip(a[:3], b[:3]):
                assert torch.equal(u, v), tag
        print(f"  windows [{tag}]: {len(ref_out)} windows identical to the reference")
        cases[tag + ":count"] = len(ref_out)
        cases[tag + ":labels"] = np.array([w[3] for w in ref_out])
        cases[tag + ":offsets"] = np.array([float(w[1][0]) for w in ref_out])            # the random time offsets
        cases[tag + ":xsum"] = np.array([float(w[0].double().sum()) for w in ref_out])
        cases[tag + ":csum"] = np.array([float(w[2].double().sum()) for w in ref_out])
        for k in (0, 1, len(ref_out) - 1):                                           # three windows in full
            cases[f"{tag}:x{k}"], cases[f"{tag}:o{k}"], cases[f"{tag}:c{k}"] = ref_out[k][0], ref_out[k][1], ref_out[k][2]
    save("g9_windows", seq3=seq, dist3=ref_dl.calc_distances(seq.clone()), x3=rx, o3=ro, c3=rc, **cases)


def g10_curves():
    """Slider geometry of the reference's export step (export/path_approximator.py, export/slider_path.py: numpy only,
    importable without `slider`) vs osu_diffusion_amd.curves on the same control points."""
    import export.slider_path as ref_sp  # noqa: E402  (reference)
    from export.path_approximator import approximate_bezier, approximate_catmull, approximate_circular_arc
    from osu_diffusion_amd import curves as C
    print("G10 curves: flattened slider paths, arc-length queries")
    rng = np.random.default_rng(77)
    cases = []
    # hand-made: the reference's own demo path, red anchors, degenerate arcs, single / double points
    cases.append(("Bezier", 100 * np.array([[0, 0], [1, 1], [1, -1], [2, 0], [2, 0], [3, -1], [2, -2]], dtype=float), None))
    cases.append(("Bezier", np.array([[100, 100]], dtype=float), None))
    cases.append(("Bezier", np.array([[100, 100], [200, 150]], dtype=float), 60.0))
    cases.append(("Bezier", np.array([[100, 100], [200, 150]], dtype=float), 200.0))
    cases.append(("Linear", np.array([[10, 10], [110, 10], [110, 210], [50, 300]], dtype=float), 250.0))
    cases.append(("PerfectCurve", np.array([[100, 100], [150, 50], [200, 100]], dtype=float), None))
    cases.append(("PerfectCurve", np.array([[100, 100], [150, 50], [200, 100]], dtype=float), 120.0))
    cases.append(("PerfectCurve", np.array([[100, 100], [150, 160], [200, 100]], dtype=float), 400.0))
    cases.append(("PerfectCurve", np.array([[100, 100], [150, 100], [200, 100]], dtype=float), None))      # collinear -> bezier
    cases.append(("PerfectCurve", np.array([[100, 100], [150, 50], [200, 100], [250, 80]], dtype=float), None))  # 4 points -> bezier
    cases.append(("Catmull", np.array([[50, 50], [120, 200], [260, 90], [400, 300]], dtype=float), None))
    cases.append(("Catmull", np.array([[50, 50], [120, 200]], dtype=float), 100.0))
    for k in range(24):  # random: integer (as in .osu files) and fractional control points
        n = int(rng.integers(2, 9))
        pts = rng.uniform(0, 512, (n, 2)) * np.array([1.0, 0.75])
        if k % 2 == 0:
            pts = np.round(pts)
        kind = ("Bezier", "Catmull", "PerfectCurve", "Linear")[k % 4]
        if kind == "PerfectCurve":
            pts = pts[:3] if len(pts) >= 3 else np.vstack([pts, pts[-1:] + [[30.0, 40.0]]])
        if kind == "Bezier" and n >= 5 and k % 3 == 0:
            pts[2] = pts[3]  # a red anchor
        want = None if k % 3 == 0 else float(rng.uniform(20, 600))
        cases.append((kind, pts, want))
    out = {"count": len(cases)}
    progresses = np.array([0.0, 0.1, 0.3333, 0.5, 0.77, 1.0])
    for i, (kind, pts, want) in enumerate(cases):
        ref = ref_sp.SliderPath(kind, pts.copy(), want)
        mine = C.SliderPath(kind, pts.copy(), want)
        rcum = np.asarray(ref.cumulative_length, dtype=float)
        rpath = np.vstack(ref.calculated_path)[:len(rcum)] if len(ref.calculated_path) else np.zeros((0, 2))
        tol = 1e-9 if kind == "PerfectCurve" else 0.0
        assert mine.calculated_path.shape == rpath.shape, (i, kind, mine.calculated_path.shape, rpath.shape)
        assert np.abs(mine.calculated_path - rpath).max(initial=0) <= tol, (i, kind)
        assert np.abs(mine.cumulative_length - rcum).max() <= tol * 100, (i, kind)
        rpos = np.stack([ref.position_at(p) for p in progresses])
        mpos = np.stack([mine.position_at(p) for p in progresses])
        assert np.abs(rpos - mpos).max() <= tol * 100, (i, kind)
        target = rpos[3] + np.array([3.0, -2.0])
        # position_to_progress lives in export/create_beatmap.py, which imports `slider`: restated inline for the fixture
        # from its text (create_beatmap.py:156-170) on the REFERENCE path object
        t = 1
        for _ in range(100):
            g = np.linalg.norm(ref.position_at(t) - target) - np.linalg.norm(ref.position_at(t - 1e-4) - target)
            t -= g
            if g == 0 or t < 0 or t > 1:
                break
        rprog = float(np.clip(t, 0, 1))
        assert abs(C.position_to_progress(mine, target) - rprog) <= 1e-9, (i, kind)
        sub_r = []
        ref.get_path_to_progress(sub_r, 0.2, 0.9)
        sub_m = mine.path_to_progress(0.2, 0.9)
        assert np.abs(np.vstack(sub_r) - sub_m).max() <= tol * 100, (i, kind)
        out.update({f"{i}:kind": kind, f"{i}:points": pts, f"{i}:want": np.nan if want is None else want,
                    f"{i}:path": rpath, f"{i}:cum": rcum, f"{i}:pos": rpos, f"{i}:target": target, f"{i}:progress": rprog,
                    f"{i}:sub": np.vstack(sub_r)})
    # the flatteners on their own
    bez = rng.uniform(0, 400, (6, 2))
    assert np.array_equal(C.flatten_bezier(bez), approximate_bezier(bez))
    cat = np.round(rng.uniform(0, 400, (5, 2)))
    ref_cat = np.vstack(approximate_catmull(cat))
    keep = np.concatenate([[True], (ref_cat[1:] != ref_cat[:-1]).any(1)])
    assert np.array_equal(C.flatten_catmull(cat)[np.concatenate([[True], (C.flatten_catmull(cat)[1:] != C.flatten_catmull(cat)[:-1]).any(1)])],
                          ref_cat[keep])
    arc = np.array([[0.0, 0.0], [30.0, 40.0], [90.0, 10.0]])
    assert np.abs(C.flatten_arc(arc) - np.vstack(approximate_circular_arc(arc))).max() <= 1e-9
    print(f"  {len(cases)} slider paths identical to the reference (arcs to 1e-9 px)")
    out.update(bez_in=bez, bez_out=approximate_bezier(bez), progresses=progresses)
    save("g10_curves", **out)


def g11_inpaint():
    """In-painting through `denoised_fn` (testing/test_toy.py:56-74: keep the prediction where mask is True, force the known
    coordinates elsewhere): single p_sample / ddim_sample steps on fixed model outputs and a chained 20-step loop with
    `model.forward` (no guidance), from the reference."""
    print("G11 in-painting (denoised_fn masks)")
    g = torch.Generator().manual_seed(11)
    N, T = 6, 64
    ref = ref_create_diffusion("250", noise_schedule="squaredcos_cap_v2")
    ora = do.create_schedule("250", "squaredcos_cap_v2")
    nt = ref.num_timesteps
    t = torch.tensor([0, 1, nt // 2, nt - 1, 0, nt - 2], dtype=torch.long)
    x = torch.randn(N, 2, T, generator=g) + 0.5
    mout = torch.randn(N, 4, T, generator=g)
    mout[:, 2:] = mout[:, 2:].clamp(-1.5, 1.5)
    known = torch.rand(N, 2, T, generator=g) * 3.4 - 1.2          # some values outside [-1, 2]: the clamp comes after the mask
    mask = torch.rand(N, 2, T, generator=g) < 0.3
    mask[:, :, -1] = True
    fn = lambda v: torch.where(mask, v, known)  # noqa: E731
    model = lambda *_a, **_k: mout  # noqa: E731
    outs = {}
    for name, step, kw in [("p", ref.p_sample, {}), ("ddim", ref.ddim_sample, {"eta": 0.5})]:
        torch.manual_seed(78)
        r = step(model, x, t, clip_denoised=True, denoised_fn=fn, **kw)
        torch.manual_seed(78)
        nz = torch.randn_like(x)
        o = (do.p_sample_step(ora, mout, x, t, nz, denoised_fn=fn) if name == "p"
             else do.ddim_step(ora, mout, x, t, nz, eta=0.5, denoised_fn=fn))
        close(o["sample"], r["sample"], 0.0, f"inpaint {name} sample")
        close(o["pred_xstart"], r["pred_xstart"], 0.0, f"inpaint {name} x0")
        outs[name + "_sample"], outs[name + "_x0"], outs[name + "_noise"] = r["sample"], r["pred_xstart"], nz
    # chained loop, as test_toy.py runs it: only the last object of each window is sampled, the rest is given
    shape, wseed = TINY, 11
    sd = mo.seeded_state_dict(shape, wseed)
    net = ref_model_for(shape, sd)
    n, Tw = 4, 64
    (x0, o, c), y = synthetic_windows(n, Tw, shape.num_classes, seed=9, train_offsets=False)
    y = torch.full_like(y, shape.num_classes)                     # null class (test_toy.py:45)
    lmask = torch.zeros(n, 2, Tw, dtype=torch.bool)
    lmask[:, :, -1] = True
    lfn = lambda v: torch.where(lmask, v, x0)  # noqa: E731
    d20 = ref_create_diffusion("20", noise_schedule="squaredcos_cap_v2")
    o20 = do.create_schedule("20", "squaredcos_cap_v2")
    torch.manual_seed(5)
    z = lfn(torch.randn(n, 2, Tw))
    kw = dict(o=o, c=c, y=y, attn_mask=None)
    torch.manual_seed(6)
    final = d20.p_sample_loop(net.forward, z.shape, z, denoised_fn=lfn, clip_denoised=True, model_kwargs=kw, device="cpu")
    torch.manual_seed(6)
    noises = torch.stack([torch.randn_like(z) for _ in range(20)])
    mine = do.sample_loop(o20, lambda xx, tt: mo.forward(sd, shape, xx, tt, o, c, y), z, noises, denoised_fn=lfn)
    close(mine, final, 5e-4, "inpaint loop")
    assert torch.equal(final[~lmask], x0[~lmask].clamp(-1, 2))    # the given coordinates come out untouched
    save("g11_inpaint", x=x, t=t, model_out=mout, known=known, mask=mask, **outs,
         shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]), wseed=wseed, wsum=checksum(sd),
         loop_x0=x0, loop_o=o, loop_c=c, loop_y=y, loop_mask=lmask, loop_z=z, loop_noises=noises, loop_final=final)


# ------------------------------------------------------------------ round 2: DiT-B geometry, long loops
BASE = mo.shape_of("DiT-B", num_classes=10)                                  # D=768, 12 heads, 12 blocks: the benchmarked geometry
BASE2 = mo.DitShape(depth=2, hidden=768, heads=12, num_classes=10)           # the same width, two blocks (gradient fixture)


def g3_forward_dit_b():
    """One forward at the geometry bench.py times (D=768, 12 heads of 64, depth 12, T=128): the GEMM instantiations with
    K = 768 / 3072 and the 12-head attention are checked against REFERENCE outputs, not only against themselves."""
    print("G3 forward, DiT-B geometry")
    for tag, wseed, kw in [("dit_b_T128", 14, {}), ("dit_b_T128_rough", 15, dict(pos_gain=1.0, mod_std=0.2))]:
        sd = mo.seeded_state_dict(BASE, wseed, **kw)
        ref = ref_model_for(BASE, sd)
        x, t, o, c, y = inputs(4, 128, BASE, seed=300, ts=[0, 1, 500, 999])
        out_ref = ref(x, t, o, c, y)
        close(mo.forward(sd, BASE, x, t, o, c, y), out_ref, 1e-4 if kw else 4e-5, tag)
        cfg4 = ref.forward_with_cfg(x, t, o, c, y, 4.0)
        cfg1 = ref.forward_with_cfg(x, t, o, c, y, 1.0)
        close(mo.forward_with_cfg(sd, BASE, x, t, o, c, y, 4.0), cfg4, 4e-4, tag + " cfg4")
        save(f"g3_forward_{tag}", shape=np.array([BASE.depth, BASE.hidden, BASE.heads, BASE.num_classes]), wseed=wseed,
             wsum=checksum(sd), rough=bool(kw), x=x, t=t, o=o, c=c, y=y, out=out_ref, out_cfg4=cfg4, out_cfg1=cfg1)


def grad_probe(name, g):
    """What a fixture keeps of one gradient tensor: norm, a seeded random projection (catches transposes / permutations) and a
    strided sample of the values (every tensor in full would be 90 MB at this width)."""
    flat = g.detach().double().flatten()
    gen = torch.Generator().manual_seed(sum(ord(ch) * (i + 1) for i, ch in enumerate(name)) % (2 ** 31))
    proj = torch.randn(flat.numel(), generator=gen, dtype=torch.float64)
    stride = max(1, flat.numel() // 4096)
    return float(flat.norm()), float((flat * proj).sum() / proj.norm()), flat[::stride][:4096].float()


def g7_training_dit_b():
    """training_losses + backward at D=768 / 12 heads (two blocks, B=4, T=128): the 256x256 and 256x192 GEMM tiles, the
    split-K weight-gradient kernel and the attention backward of the benchmarked geometry against REFERENCE gradients."""
    print("G7 training, DiT-B width")
    torch.set_grad_enabled(True)
    shape, wseed = BASE2, 16
    B, T = 4, 128
    (x, o, c), y = synthetic_windows(B, T, shape.num_classes, seed=31, train_offsets=True)
    t = torch.tensor([0, 7, 500, 999], dtype=torch.long)
    noise = torch.randn(B, 2, T, generator=torch.Generator().manual_seed(32))
    drop = torch.tensor([False, False, True, False])
    y_eff = torch.where(drop, torch.full_like(y, shape.num_classes), y)
    sd = mo.seeded_state_dict(shape, wseed)
    ref = ref_model_for(shape, sd)
    dref = ref_create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True)
    terms = dref.training_losses(ref, x, t, dict(o=o, c=c, y=y_eff), noise=noise)
    terms["loss"].mean().backward()
    grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    osd = {k: v.clone().requires_grad_(k != "xoc_embedder.playfield_size") for k, v in sd.items()}
    ora = do.create_schedule("", "squaredcos_cap_v2")
    ot = do.training_losses(ora, lambda xx, tt: mo.forward(osd, shape, xx, tt, o, c, y, drop_mask=drop), x, t, noise, loss="l1")
    ot["loss"].mean().backward()
    for k in ("loss", "vb", "l1"):
        close(ot[k].detach(), terms[k].detach(), 2e-5, f"dit-b width term {k}")
    worst = max((osd[k].grad - grads[k]).abs().max().item() for k in grads)
    print(f"  oracle vs reference [dit-b width grads]: max|d| = {worst:.3e}")
    assert worst < 1e-4
    keys = sorted(grads)
    probes = {k: grad_probe(k, grads[k]) for k in keys}
    arrs = {"sample:" + k: probes[k][2] for k in keys}
    save("g7_train_dit_b", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]), wseed=wseed,
         wsum=checksum(sd), x=x, o=o, c=c, y=y, t=t, noise=noise, drop=drop, loss=terms["loss"].detach(),
         vb=terms["vb"].detach(), main=terms["l1"].detach(), grad_keys=np.array(keys),
         grad_norms=np.array([probes[k][0] for k in keys]), grad_projs=np.array([probes[k][1] for k in keys]), **arrs)
    torch.set_grad_enabled(False)


def g6_long_loops():
    """a full 250-step respaced p_sample_loop, and the head (t = 999..995, where sqrt(1/ac - 1) ~ 2e4 and the
    clamp decides x0) and tail (t = 4..0) of the 1000-step schedule, each on the damped seeded weights (pos_gain 0.1: fp32 and
    fp64 evaluations of the reference agree to 1e-4, the 1e-3 bound is well-posed) and on UNDAMPED reference-like weights
    (pos_gain 1), for which the reference's own fp32 result is stored next to an fp64 evaluation of the same loop: their
    distance is the accuracy any fp32 implementation can claim on those weights."""
    print("G6 long loops: p250, 1000-step head and tail")
    shape, wseed = TINY, 11
    n, T = 2, 64
    (x0, o, c), y = synthetic_windows(n, T, shape.num_classes, seed=7, train_offsets=False)
    o, c = torch.cat([o, o]), torch.cat([c, c])
    y = torch.cat([y, torch.full_like(y, shape.num_classes)])
    kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=None)
    for damp_tag, pos_gain in (("", 0.1), ("_undamped", 1.0)):
        sd = mo.seeded_state_dict(shape, wseed, pos_gain=pos_gain)
        sd64 = mo.to_dtype(sd, torch.float64)
        ref = ref_model_for(shape, sd)
        fn = lambda xx, tt: mo.forward_with_cfg(sd, shape, xx, tt, o, c, y, 4.0)  # noqa: E731
        fn64 = lambda xx, tt: mo.forward_with_cfg(sd64, shape, xx.double(), tt, o.double(), c.double(), y, 4.0).float()  # noqa: E731
        # (a) 250-step loop
        dref = ref_create_diffusion("250", noise_schedule="squaredcos_cap_v2")
        ora = do.create_schedule("250", "squaredcos_cap_v2")
        torch.manual_seed(13)
        z = torch.randn(n, 2, T)
        z = torch.cat([z, z])
        torch.manual_seed(14)
        final = dref.p_sample_loop(ref.forward_with_cfg, z.shape, z, clip_denoised=True, model_kwargs=kw, device="cpu")
        torch.manual_seed(14)
        noises = torch.stack([torch.randn_like(z) for _ in range(250)])
        mine = do.sample_loop(ora, fn, z, noises)
        d32 = (mine - final).abs().max().item()
        f64 = do.sample_loop(ora, fn64, z, noises)
        d64 = (f64 - final).abs().max().item()
        print(f"  p250{damp_tag}: oracle fp32 vs reference {d32:.3e}; fp64 evaluation vs reference fp32 {d64:.3e}")
        if pos_gain < 1:
            assert d32 <= 5e-4 and d64 <= 5e-4
        save(f"g6_loop_p250{damp_tag}", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]), wseed=wseed,
             wsum=checksum(sd), pos_gain=pos_gain, z=z, o=o, c=c, y=y, noises=noises, final=final,
             final_fp64=f64, respacing="250", eta=-1.0)
        # (b) head and tail of the 1000-step schedule: reference p_sample, step by step
        d1000 = ref_create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
        o1000 = do.create_schedule("1000", "squaredcos_cap_v2")
        out = {}
        for part, first, last, start in (("head", 999, 995, z),
                                         ("tail", 4, 0, torch.cat([x0, x0]) + 0.02 * torch.randn(2 * n, 2, T, generator=torch.Generator().manual_seed(15)))):
            xs = start.clone()
            torch.manual_seed(16)
            for i in range(first, last - 1, -1):
                tt = torch.tensor([i] * len(xs))
                xs = d1000.p_sample(ref.forward_with_cfg, xs, tt, clip_denoised=True, model_kwargs=kw)["sample"]
            torch.manual_seed(16)
            nz = torch.stack([torch.randn_like(start) for _ in range(first - last + 1)])
            xo = start.clone()
            tm = torch.from_numpy(o1000.timestep_map)
            for k, i in enumerate(range(first, last - 1, -1)):
                tt = torch.tensor([i] * len(xo))
                xo = do.p_sample_step(o1000, fn(xo, tm[tt]), xo, tt, nz[k])["sample"]
            dd = (xo - xs).abs().max().item()
            print(f"  1000-step {part}{damp_tag}: oracle vs reference {dd:.3e}")
            if pos_gain < 1:
                assert dd <= 5e-4
            out.update({part + "_start": start, part + "_noises": nz, part + "_final": xs, part + "_first": first, part + "_last": last})
        save(f"g6_steps_1000{damp_tag}", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]), wseed=wseed,
             wsum=checksum(sd), pos_gain=pos_gain, o=o, c=c, y=y, **out)


def g6_loop_p1000_dit_b():
    """BASELINE configs[3] end to end against the reference (reference: sample.py:174-182, gaussian_diffusion.py:469-561): DiT-B
    (12 blocks, D = 768, 12 heads; seeded non-zero weights), n = 2 windows -> N = 4 rows [cond; uncond], T = 128, "1000" steps,
    cfg 4.0, the reference's own p_sample_loop_progressive with torch's CPU generator seeded right before the loop.  Stored: the
    final coordinates and the state after 250 / 500 / 750 executed steps.  The per-step noise is NOT stored (4 MB): tests redraw it
    from the same seed in the same order (one randn_like per step) and check it against the stored checksum and first values.
    An fp64 evaluation of the same loop is stored next to it: its distance from the reference's fp32 result is the accuracy any
    fp32-class implementation can claim on these weights."""
    print("G6 1000-step CFG-4 loop, DiT-B geometry (minutes of CPU)")
    shape, wseed, n, T, steps, nseed = BASE, 16, 2, 128, 1000, 24
    (x0, o, c), y = synthetic_windows(n, T, shape.num_classes, seed=7, train_offsets=False)
    o, c = torch.cat([o, o]), torch.cat([c, c])
    y = torch.cat([y, torch.full_like(y, shape.num_classes)])
    kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=None)
    sd = mo.seeded_state_dict(shape, wseed)
    ref = ref_model_for(shape, sd)
    dref = ref_create_diffusion(str(steps), noise_schedule="squaredcos_cap_v2")
    ora = do.create_schedule(str(steps), "squaredcos_cap_v2")
    torch.manual_seed(23)
    z = torch.randn(n, 2, T)
    z = torch.cat([z, z])
    snaps = {}
    torch.manual_seed(nseed)
    k = 0
    final = None
    for out in dref.p_sample_loop_progressive(ref.forward_with_cfg, z.shape, z, clip_denoised=True, model_kwargs=kw, device="cpu"):
        k += 1
        final = out["sample"]
        if k in (250, 500, 750):
            snaps[f"after_{k}"] = final.clone()
    assert k == steps
    torch.manual_seed(nseed)
    noises = torch.stack([torch.randn_like(z) for _ in range(steps)])
    fn = lambda xx, tt: mo.forward_with_cfg(sd, shape, xx, tt, o, c, y, 4.0)  # noqa: E731
    mine = do.sample_loop(ora, fn, z, noises)
    d32 = (mine - final).abs().max().item()
    print(f"  p1000 dit_b: oracle fp32 vs reference {d32:.3e}")
    assert d32 <= 5e-4, d32
    extra = {}
    if os.environ.get("OSUD_GOLDEN_FP64", "1") != "0":
        sd64 = mo.to_dtype(sd, torch.float64)
        fn64 = lambda xx, tt: mo.forward_with_cfg(sd64, shape, xx.double(), tt, o.double(), c.double(), y, 4.0).float()  # noqa: E731
        f64 = do.sample_loop(ora, fn64, z, noises)
        print(f"  p1000 dit_b: fp64 evaluation vs reference fp32 {(f64 - final).abs().max().item():.3e}")
        extra["final_fp64"] = f64
    save("g6_loop_p1000_dit_b", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]), wseed=wseed,
         wsum=checksum(sd), z=z, o=o, c=c, y=y, final=final, noise_seed=nseed, noise_sum=float(noises.double().sum()),
         noise_abs_sum=float(noises.double().abs().sum()), noise_head=noises[0, 0, 0, :8], noise_tail=noises[-1, -1, -1, -8:],
         respacing=str(steps), cfg_scale=4.0, oracle_fp32_vs_reference=d32, **snaps, **extra)


def g12_cli_toy():
    """What `sample.py --noise cpu` must reproduce: the reference's own sampling call sequence (sample.py:39-108,174-182 --
    torch.manual_seed(seed); z = randn(n, 2, T); doubled [cond; null] batch; banded mask; p_sample_loop with randn_like per
    step) on the toy beatmap's window tensors.  The reference's sample.py cannot be imported (`slider`), so its few lines of
    setup are re-enacted here around the REFERENCE model and diffusion objects; the window tensors come from this repository's
    `.osu` reader (parsing itself is unpinned) and are stored in the fixture."""
    print("G12 CLI: reference sampling flow on the toy beatmap")
    from osu_diffusion_amd.beatmap import Beatmap, beatmap_to_sequence
    from osu_diffusion_amd.windows import split_and_process_sequence_no_augment
    shape, wseed, num_classes, label = SMALL, 21, 10, 3
    sd = mo.seeded_state_dict(shape, wseed)
    seq_full = beatmap_to_sequence(Beatmap.from_path(os.path.join(HERE, "toy_beatmap.osu")))
    out = {}
    for tag, steps, plot_time, seq_len_flag in (("full100", 100, None, 128), ("trim250", 250, 30000.0, 128)):
        seq = seq_full
        if plot_time is not None:  # sample.py:59-63
            start = int(torch.nonzero(seq[2] >= plot_time)[0])
            seq = seq[:, start:start + seq_len_flag]
        (sx, so, sc), T = split_and_process_sequence_no_augment(seq)
        so = so - so[0]                                                    # sample.py:65
        mask = banded_attn_mask(T, seq_len_flag)                           # sample.py:81-84
        r, i = torch.meshgrid(torch.arange(T), torch.arange(T), indexing="ij")
        assert torch.equal(mask, ~((r >= i - seq_len_flag) & (r < i + seq_len_flag)))
        dref = ref_create_diffusion(str(steps), noise_schedule="squaredcos_cap_v2")
        torch.manual_seed(0)                                               # sample.py:41 (--seed 0)
        # sample.py:69-76: the model is CONSTRUCTED after seeding -- its random initialisation advances the generator before the
        # checkpoint overwrites the weights -- so the noise below starts from that generator state
        ref = ref_models.DiT_models["DiT-S"](num_classes=num_classes, context_size=19 - 3 + 128)
        ref.load_state_dict(sd)
        ref.eval()
        rng_after_init = torch.get_rng_state()
        n = 1
        z = torch.randn(n, 2, T)                                           # sample.py:97
        o, c = so.repeat(n, 1), sc.repeat(n, 1, 1)
        y = torch.tensor([label])
        z, o, c = torch.cat([z, z], 0), torch.cat([o, o], 0), torch.cat([c, c], 0)
        y = torch.cat([y, torch.tensor([num_classes] * n)], 0)
        kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=mask)
        final = dref.p_sample_loop(ref.forward_with_cfg, z.shape, z, clip_denoised=True, model_kwargs=kw, progress=False, device="cpu")
        samples, _ = final.chunk(2, dim=0)
        # the oracle through the same flow (a 20-step run over the whole 757-object map is NOT well-posed to 1e-3: two fp32
        # evaluations -- this oracle and the reference -- end 4.7e-3 apart; 50 steps: 7.6e-4; 100 steps: 2e-4)
        torch.set_rng_state(rng_after_init)
        zz = torch.randn(n, 2, T)
        zz = torch.cat([zz, zz])
        nz = torch.stack([torch.randn_like(zz) for _ in range(steps)])
        mine = do.sample_loop(do.create_schedule(str(steps), "squaredcos_cap_v2"),
                              lambda xx, tt: mo.forward_with_cfg(sd, shape, xx, tt, o, c, y, 4.0, attn_mask=mask), zz, nz)
        close(mine[:n], samples, 5e-4, f"cli {tag}")
        print(f"  {tag}: T={T}, {steps} steps, final range [{samples.min():.3f}, {samples.max():.3f}]")
        out.update({tag + ":T": T, tag + ":steps": steps, tag + ":final": samples, tag + ":x": sx, tag + ":o": so, tag + ":c": sc})
        if plot_time is not None:
            out[tag + ":plot_time"] = plot_time
        if tag == "trim250":
            # sample.py:186-205 on the same run: --refine-ckpt's weights into the SAME model object, then refine_iters = 10 calls of
            # p_sample at t = 0 on the loop's result (both halves of the doubled batch, mask and all) -> fixture g15_refine_cli
            rseed = 22
            sd_r = mo.seeded_state_dict(shape, rseed)
            ref.load_state_dict(sd_r)
            img = final
            for _ in range(10):
                img = dref.p_sample(ref.forward_with_cfg, img, torch.tensor([0] * img.shape[0]), clip_denoised=True, model_kwargs=kw)["sample"]
            print(f"  {tag}: refine pass moved the result by {(img - final).abs().max().item():.3e}")
            save("g15_refine_cli", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]), wseed=wseed, wsum=checksum(sd),
                 refine_wseed=rseed, refine_wsum=checksum(sd_r), label=label, style_id=5, cfg_scale=4.0, seed=0, steps=steps, plot_time=plot_time,
                 T=T, final=samples, refined=img.chunk(2, dim=0)[0], refine_iters=10)
    save("g12_cli_toy", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]), wseed=wseed, wsum=checksum(sd),
         label=label, style_id=5, cfg_scale=4.0, seed=0, **out)


def g13_export():
    """export/create_beatmap.py:22-147 (sampled sequence -> hit objects + slider-velocity timing points) run for real on a
    jittered copy of the toy beatmap's sequence.  Its `slider` imports are satisfied by plain records (Position, Circle,
    Slider, Spinner, TimingPoint, Curve, a Beatmap holding the toy map's timing points): the reference's own logic -- pixel
    rounding, anchor bookkeeping, SliderPath geometry, nearest-progress search, span count, velocity formula -- is what runs.
    What `slider` itself would do with the objects (packing the .osu text) stays unpinned."""
    import collections
    import types
    from datetime import timedelta
    print("G13 export: reference create_beatmap on the toy sequence")
    from osu_diffusion_amd import beatmap as B

    class Position(collections.namedtuple("Position", "x y")):
        pass

    class Rec:
        def __init__(self, *a, **k):
            self.args, self.__dict__ = a, dict(self.__dict__, **k)

    class Circle(Rec):
        def __init__(self, position, time, hitsound, new_combo=False):
            super().__init__(position=position, time=time, hitsound=hitsound, new_combo=new_combo)

    class Spinner(Rec):
        def __init__(self, position, time, hitsound, end_time, new_combo=False):
            super().__init__(position=position, time=time, hitsound=hitsound, end_time=end_time, new_combo=new_combo)

    class Slider(Rec):
        pass

    class TimingPoint(Rec):
        def __init__(self, offset, ms_per_beat, meter, sample_type, sample_set, volume, parent, kiai_mode):
            super().__init__(offset=offset, ms_per_beat=ms_per_beat, meter=meter, sample_type=sample_type, sample_set=sample_set,
                             volume=volume, parent=parent, kiai_mode=kiai_mode)

    class Curve(Rec):
        @staticmethod
        def from_kind_and_points(kind, points, req_length):
            return Curve(kind=kind, points=points, req_length=req_length)

    class MultiBezier(Rec):
        def __init__(self, points, req_length):
            super().__init__(points=points, req_length=req_length)

    class RefBeatmap(Rec):
        def timing_point_at(self, time):  # slider.Beatmap.timing_point_at: the last point at or before `time`, else the first
            cur = self.timing_points[0]
            for tp in self.timing_points:
                if tp.offset <= time:
                    cur = tp
                else:
                    break
            return cur

    mods = {"slider": dict(Position=Position),
            "slider.beatmap": dict(Beatmap=RefBeatmap, Circle=Circle, HitObject=Rec, Slider=Slider, Spinner=Spinner, TimingPoint=TimingPoint,
                                   HoldNote=Rec),
            "slider.curve": dict(Catmull=Rec, Curve=Curve, Linear=Rec, MultiBezier=MultiBezier, Perfect=Rec)}
    saved = {k: sys.modules.get(k) for k in list(mods) + ["export.create_beatmap"]}
    for name, attrs in mods.items():
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
    sys.modules.pop("export.create_beatmap", None)
    try:
        import export.create_beatmap as ref_cb  # noqa: E402  (reference)
        src = B.Beatmap.from_path(os.path.join(HERE, "toy_beatmap.osu"))
        seq = B.beatmap_to_sequence(src)                               # (19, L): px, px, ms, one-hot types
        g = torch.Generator().manual_seed(13)
        jit = seq.clone()
        jit[:2] += torch.randn(2, seq.shape[1], generator=g) * 6.0     # "sampled" positions: a few pixels off the source's
        norm = jit.clone()
        norm[0] /= 512.0
        norm[1] /= 384.0
        # reference-side beatmap record: the toy map's timing points (red points have parent None, green ones point at theirs)
        tps, by_id = [], {}
        for tp in src.timing_points:
            r = TimingPoint(timedelta(milliseconds=tp.offset), tp.ms_per_beat, tp.meter, tp.sample_type, tp.sample_set, tp.volume,
                            by_id.get(id(tp.parent)), tp.kiai_mode)
            by_id[id(tp)] = r
            tps.append(r)
        fields = ("format_version audio_filename audio_lead_in preview_time countdown sample_set stack_leniency mode letterbox_in_breaks "
                  "widescreen_storyboard bookmarks distance_spacing beat_divisor grid_size timeline_zoom title title_unicode artist "
                  "artist_unicode creator source tags beatmap_set_id hp_drain_rate circle_size overall_difficulty approach_rate "
                  "slider_tick_rate").split()
        ref_bm = RefBeatmap(timing_points=tps, slider_multiplier=src.slider_multiplier, **{f: None for f in fields})
        out = ref_cb.create_beatmap(norm, ref_bm, "golden")
        objs, n_tp = out.hit_objects, len(out.timing_points)
        ms = lambda td: td.total_seconds() * 1000.0  # noqa: E731
        kind, xy, t0, t1, rep, length, letter, npts, pts = [], [], [], [], [], [], [], [], []
        for ho in objs:
            k = 0 if isinstance(ho, Circle) else (1 if isinstance(ho, Spinner) else 2)
            kind.append(k)
            xy.append([ho.position.x, ho.position.y])
            t0.append(ms(ho.time))
            t1.append(ms(ho.end_time) if k else ms(ho.time))
            rep.append(ho.repeat if k == 2 else 0)
            length.append(float(ho.length) if k == 2 else 0.0)
            letter.append(ho.curve.kind if k == 2 else "-")
            cp = [[q.x, q.y] for q in ho.curve.points] if k == 2 else []
            npts.append(len(cp))
            pts += cp
        new_tp = out.timing_points[len([tp for tp in tps if tp.parent is None]):]
        print(f"  {len(objs)} hit objects ({kind.count(2)} sliders, {kind.count(1)} spinners), {len(new_tp)} slider-velocity points")
        save("g13_export", seq=norm, kind=np.array(kind), xy=np.array(xy, dtype=np.float64), time=np.array(t0), end_time=np.array(t1),
             repeat=np.array(rep), length=np.array(length), letter=np.array(letter), n_points=np.array(npts),
             points=np.array(pts, dtype=np.float64).reshape(-1, 2), new_combo=np.array([bool(ho.new_combo) for ho in objs]),
             tp_offset=np.array([ms(tp.offset) for tp in new_tp]), tp_ms_per_beat=np.array([tp.ms_per_beat for tp in new_tp]),
             tp_parent_ms_per_beat=np.array([tp.parent.ms_per_beat for tp in new_tp]), n_timing_points=n_tp)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def g14_timestep_sampler():
    """diffusion/timestep_sampler.py of the reference (unused by its scripts; its loss-aware class cannot even be constructed
    under numpy >= 1.24 -- `np.int` -- so the alias is restored for this run): sampled timesteps and weights of both samplers
    under a seeded numpy generator, before and after a loss history."""
    print("G14 timestep samplers")
    import diffusion.timestep_sampler as ref_ts  # noqa: E402  (reference)
    if not hasattr(np, "int"):
        np.int = int
    dref = ref_create_diffusion("", noise_schedule="squaredcos_cap_v2")
    out = {}
    uni = ref_ts.create_named_schedule_sampler("uniform", dref)
    np.random.seed(11)
    t, w = uni.sample(32, "cpu")
    out["uniform_t"], out["uniform_w"] = t, w
    lsm = ref_ts.create_named_schedule_sampler("loss-second-moment", dref)
    np.random.seed(12)
    t, w = lsm.sample(16, "cpu")          # not warmed up: uniform
    out["cold_t"], out["cold_w"] = t, w
    rng = np.random.default_rng(13)
    ts_hist = np.concatenate([np.tile(np.arange(1000), 10), rng.integers(0, 1000, 3000)])   # full history, then overwrites
    loss_hist = np.abs(rng.normal(size=ts_hist.shape)) * (1 + ts_hist / 250.0)
    lsm.update_with_all_losses(list(ts_hist), list(loss_hist))
    out["weights"] = lsm.weights()
    np.random.seed(14)
    t, w = lsm.sample(64, "cpu")
    out["warm_t"], out["warm_w"] = t, w
    save("g14_timestep_sampler", ts_hist=ts_hist, loss_hist=loss_hist, **out)


def g15_refine():
    """The refine pass of sample.py:186-205: after the sampling loop the model's weights are replaced by a second checkpoint's
    (`--refine-ckpt`) and the result goes through `refine_iters` = 10 calls of `diffusion.p_sample(model.forward_with_cfg, img,
    t = 0, clip_denoised=True)` -- the spaced diffusion's step index 0 (model timestep 0), no noise (nonzero_mask = 0).  Two cases:
    the whole re-enactment on the tiny model ("20"-step loop with weights A, refine with weights B), and the refine pass alone at
    DiT-B geometry (12 blocks, T = 128) from windows with a little noise.  Stored: the image before the pass and after 1 / 5 / 10
    iterations."""
    print("G15 refine pass (p_sample at t = 0, repeated, second weight set)")
    for tag, shape, wseed, rseed, n, T, resp, with_loop in [("tiny", TINY, 11, 12, 2, 64, "20", True), ("dit_b", BASE, 16, 17, 2, 128, "250", False)]:
        sd_a, sd_b = mo.seeded_state_dict(shape, wseed), mo.seeded_state_dict(shape, rseed)
        (x0, o, c), y = synthetic_windows(n, T, shape.num_classes, seed=7, train_offsets=False)
        o, c = torch.cat([o, o]), torch.cat([c, c])
        y = torch.cat([y, torch.full_like(y, shape.num_classes)])
        kw = dict(o=o, c=c, y=y, cfg_scale=4.0, attn_mask=None)
        dref = ref_create_diffusion(resp, noise_schedule="squaredcos_cap_v2")
        ora = do.create_schedule(resp, "squaredcos_cap_v2")
        ref = ref_model_for(shape, sd_a)
        torch.manual_seed(31)
        if with_loop:
            z = torch.randn(n, 2, T)
            z = torch.cat([z, z])
            torch.manual_seed(32)
            img = dref.p_sample_loop(ref.forward_with_cfg, z.shape, z, clip_denoised=True, model_kwargs=kw, device="cpu")
        else:
            z = (x0 + 0.05 * torch.randn(n, 2, T)).clamp(-1, 1)
            z = torch.cat([z, z])
            img = z
        start = img.clone()
        ref.load_state_dict(sd_b)  # sample.py:189-190
        snaps = {}
        mine = start.clone()
        fn = lambda xx, tt: mo.forward_with_cfg(sd_b, shape, xx, tt, o, c, y, 4.0)  # noqa: E731
        tmap = torch.from_numpy(ora.timestep_map)
        for it in range(1, 11):
            t = torch.tensor([0] * img.shape[0])
            img = dref.p_sample(ref.forward_with_cfg, img, t, clip_denoised=True, model_kwargs=kw)["sample"]
            mine = do.p_sample_step(ora, fn(mine, tmap[t]), mine, t, torch.zeros_like(mine))["sample"]
            if it in (1, 5, 10):
                snaps[f"after_{it}"] = img.clone()
        close(mine, img, 2e-5, f"refine {tag}")
        print(f"  refine {tag}: moved by {(img - start).abs().max().item():.3e} over 10 iterations")
        save(f"g15_refine_{tag}", shape=np.array([shape.depth, shape.hidden, shape.heads, shape.num_classes]), wseed=wseed, wsum=checksum(sd_a),
             refine_wseed=rseed, refine_wsum=checksum(sd_b), z=z, o=o, c=c, y=y, start=start, respacing=resp, cfg_scale=4.0,
             loop_noise_seed=32 if with_loop else -1, **snaps)


if __name__ == "__main__":
    steps = [g1_schedules, g2_embeddings, g3_forward, g5_step, g6_loop, g7_training, g9_init_and_keys, g9_windows, g10_curves, g11_inpaint,
             g3_forward_dit_b, g7_training_dit_b, g6_long_loops, g6_loop_p1000_dit_b, g12_cli_toy, g13_export, g14_timestep_sampler, g15_refine]
    only = set(sys.argv[1:])  # e.g. `make_golden.py g10_curves` regenerates one family
    for fn in steps:
        if not only or fn.__name__ in only:
            fn()
    print("all golden fixtures written; oracle pinned against the reference")
