"""GPU: the two entry scripts end to end on `.osu` data — sample.py (`.osu` -> sequence -> banded-mask CFG sampling on the
native path -> `.osu` per variant, sample.py:39-205) and train.py (`<root>/TrackNNNNN/beatmaps/*.osu` -> windows -> native
training steps -> checkpoint, train.py:163-293)."""
import glob
import os
import subprocess
import sys

import pytest
import torch

from osu_diffusion_amd import beatmap as B
from tests.helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOY = os.path.join(GOLDEN, "toy_beatmap.osu")
pytestmark = pytest.mark.gpu


def run(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable] + args, cwd=cwd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout + r.stderr


def test_sample_script_osu_in_osu_out(tmp_path):
    out = run([os.path.join(ROOT, "sample.py"), "--beatmap", TOY, "--model", "DiT-S", "--num-classes", "10", "--num-sampling-steps", "20",
               "--cfg-scale", "2.0", "--seq-len", "64", "--seed", "3"], str(tmp_path))
    assert "seq len 757" in out
    files = glob.glob(os.path.join(str(tmp_path), "results", "*", "*.osu"))
    assert len(files) == 1 and os.path.basename(files[0]) == "1828346 result None 0.osu"
    src, new = B.Beatmap.from_path(TOY), B.Beatmap.from_path(files[0])
    a, b = B.beatmap_to_sequence(src), B.beatmap_to_sequence(new)
    assert b.shape == a.shape and torch.equal(a[3:].argmax(0), b[3:].argmax(0))   # same objects, same types ...
    heads = a[3:].argmax(0) <= 5
    assert (a[2, heads] - b[2, heads]).abs().max() <= 1.0                          # ... at the same times ...
    assert (a[:2, heads] - b[:2, heads]).abs().max() > 10                          # ... at newly sampled positions
    placed = b[:2, a[3:].argmax(0) <= 10]   # every column but the slider ends (those are re-derived from the path on reading)
    assert float(placed.min()) >= -512 and float(placed.max()) <= 1024 and torch.isfinite(b).all()   # clamp(-1, 2) of the sampler
    assert new.version.startswith("Diffusion None 0") and new.beatmap_id == 0
    saved = torch.load(glob.glob(os.path.join(str(tmp_path), "results", "*", "result.pt"))[0])
    assert saved.shape == (1, 19, 757)


def _cli_fixture(tmp_path):
    """The synthetic checkpoint and style index the CLI golden (g12_cli_toy) was made with."""
    import pickle

    from oracle import dit_oracle as mo
    from tests.helpers import load, weights_for

    fx = load("g12_cli_toy")
    shape, sd = weights_for(fx)
    ckpt = tmp_path / "ckpt.pt"
    torch.save({"ema": sd, "model": sd}, ckpt)             # find_model takes the "ema" weights (sample.py:31-36)
    idx = tmp_path / "beatmap_idx.pickle"
    with open(idx, "wb") as f:
        pickle.dump({int(fx["style_id"]): int(fx["label"])}, f)
    return fx, str(ckpt), str(idx)


@pytest.mark.parametrize("precision", ["fp32", "fp16f8"])
def test_sample_cli_refine_ckpt_reproduces_the_reference_run(tmp_path, precision):
    """sample.py:186-205 through the CLI: `--refine-ckpt` loads a second checkpoint into the model after the loop and runs
    `--refine-iters` = 10 p_sample steps at t = 0 (here: ONE native call).  Fixture g15_refine_cli is the reference's own flow on the
    toy beatmap's 128-object window (250 steps, cfg 4, then 10 refine iterations on the other weights): both files the CLI writes
    -- result.pt before, result_10.pt after the pass -- land within 1e-3 of it."""
    from oracle import dit_oracle as mo
    from tests.helpers import load, shape_from

    fx, ckpt, idx = _cli_fixture(tmp_path)
    rf = load("g15_refine_cli")
    sd_r = mo.seeded_state_dict(shape_from(rf), int(rf["refine_wseed"]))
    assert abs(float(sum(v.double().abs().sum() for v in sd_r.values())) - float(rf["refine_wsum"])) <= 1e-9 * float(rf["refine_wsum"])
    refine_ckpt = tmp_path / "refine.pt"
    torch.save({"ema": sd_r, "model": sd_r}, refine_ckpt)
    args = [os.path.join(ROOT, "sample.py"), "--beatmap", TOY, "--ckpt", ckpt, "--model", "DiT-S", "--num-classes", "10",
            "--num-sampling-steps", str(int(rf["steps"])), "--cfg-scale", "4.0", "--seed", "0", "--seq-len", "128",
            "--style-id", str(int(rf["style_id"])), "--beatmap-idx", idx, "--noise", "cpu", "--precision", precision,
            "--plot-time", str(float(rf["plot_time"])), "--refine-ckpt", str(refine_ckpt), "--refine-iters", str(int(rf["refine_iters"]))]
    run(args, str(tmp_path))
    pf = torch.tensor([512.0, 384.0]).view(1, 2, 1)
    before = torch.load(glob.glob(os.path.join(str(tmp_path), "results", "*", "result.pt"))[0])[:, :2] / pf
    after = torch.load(glob.glob(os.path.join(str(tmp_path), "results", "*", "result_10.pt"))[0])[:, :2] / pf
    e0 = float((before - torch.from_numpy(rf["final"])).abs().max())
    e1 = float((after - torch.from_numpy(rf["refined"])).abs().max())
    moved = float((torch.from_numpy(rf["refined"]) - torch.from_numpy(rf["final"])).abs().max())
    print(f"MEASURED cli_refine[{precision}]: loop result {e0:.3e}, refined result {e1:.3e} from the reference's (the pass moves the result by {moved:.3e})")
    assert e0 < 1e-3 and e1 < 1e-3 and moved > 1e-2
    assert len(glob.glob(os.path.join(str(tmp_path), "results", "*", "*result 5 0 10.osu"))) == 1  # the refined difficulty is written too


@pytest.mark.parametrize("precision", ["fp16m8", "fp16w8"])
def test_sample_cli_fp16_activation_forms_are_close_but_not_inside_the_tolerance(tmp_path, precision):
    """The tiers whose big GEMMs round the ACTIVATION operand to fp16 (fp16w8: all four, fp16m8: all but fc1) on the same CLI fixture
    (128-object window, 250 steps): the bulk of the coordinates lands where the tolerance tiers do (p99 < 5e-4, mean ~1e-4), but the
    sampler has coordinates that amplify a 1e-4-sized perturbation a hundredfold -- ONE of the 256 here ends 2-3e-2 away (fp16f8, whose
    perturbation is ten times smaller: 7e-5 everywhere).  That is why these tiers are NOT what this build calls "within 1e-3 of the
    reference" (bench.py's tolerance tier stays fp16f8), however fast they are; the test pins the behaviour so that the claim cannot
    drift: every coordinate but at most two inside 1e-3, the outliers below 5e-2."""
    fx, ckpt, idx = _cli_fixture(tmp_path)
    tag = "trim250"
    args = [os.path.join(ROOT, "sample.py"), "--beatmap", TOY, "--ckpt", ckpt, "--model", "DiT-S", "--num-classes", "10",
            "--num-sampling-steps", str(int(fx[tag + ":steps"])), "--cfg-scale", "4.0", "--seed", "0", "--seq-len", "128",
            "--style-id", str(int(fx["style_id"])), "--beatmap-idx", idx, "--noise", "cpu", "--precision", precision,
            "--plot-time", str(float(fx[tag + ":plot_time"]))]
    run(args, str(tmp_path))
    saved = torch.load(glob.glob(os.path.join(str(tmp_path), "results", "*", "result.pt"))[0])
    d = (saved[:, :2] / torch.tensor([512.0, 384.0]).view(1, 2, 1) - torch.from_numpy(fx[tag + ":final"])).abs().flatten().double()
    outside = int((d > 1e-3).sum())
    print(f"MEASURED cli_fast_forms[{precision}]: max {float(d.max()):.3e}, p99 {float(torch.quantile(d, 0.99)):.3e}, mean {float(d.mean()):.3e}, "
          f"{outside} of {d.numel()} coordinates outside 1e-3")
    assert outside <= 2 and float(d.max()) < 5e-2 and float(torch.quantile(d, 0.99)) < 5e-4


@pytest.mark.parametrize("precision", ["fp32", "bf16x3", "fp16f8"])
@pytest.mark.parametrize("tag", ["full100", "trim250"])
def test_sample_cli_with_cpu_noise_reproduces_the_reference_run(tmp_path, tag, precision):
    """north_star: identical (seed, beatmap, num-sampling-steps) -> final (x, y) within 1e-3 of the reference's CPU path.
    `sample.py --noise cpu --precision fp32 | bf16x3` against fixture g12_cli_toy (the reference's sampling flow on the toy beatmap:
    the whole 757-object map with 100 steps -- banded attention mask, T not a multiple of 64 -- and a 128-object window from
    t = 30 s with 250 steps; DiT-S, cfg-scale 4).  Both the exact-f32 tier and the split-bf16 tier (the MFMA-speed one) meet it."""
    fx, ckpt, idx = _cli_fixture(tmp_path)
    args = [os.path.join(ROOT, "sample.py"), "--beatmap", TOY, "--ckpt", ckpt, "--model", "DiT-S", "--num-classes", "10",
            "--num-sampling-steps", str(int(fx[tag + ":steps"])), "--cfg-scale", "4.0", "--seed", "0", "--seq-len", "128",
            "--style-id", str(int(fx["style_id"])), "--beatmap-idx", idx, "--noise", "cpu", "--precision", precision]
    if tag + ":plot_time" in fx:
        args += ["--plot-time", str(float(fx[tag + ":plot_time"]))]
    out = run(args, str(tmp_path))
    assert f"seq len {int(fx[tag + ':T'])}" in out
    saved = torch.load(glob.glob(os.path.join(str(tmp_path), "results", "*", "result.pt"))[0])  # (1, 19, T), positions in osu! pixels
    got = saved[:, :2] / torch.tensor([512.0, 384.0]).view(1, 2, 1)
    err = float((got - torch.from_numpy(fx[tag + ":final"])).abs().max())
    print(f"MEASURED cli[{tag},{precision}]: sample.py --noise cpu vs the reference's CPU run: max|d| = {err:.3e} (normalised coordinates)")
    assert err < 1e-3


def test_sample_cli_rows_sharded_over_two_ranks_equal_one_process(tmp_path):
    """torchrun --nproc-per-node 2 sample.py (both ranks on GPU 0, gloo): 3 variants -> shards of 2 and 1 rows, gathered on
    rank 0; equal to the single-process run because every rank draws the noise of all variants and keeps its rows."""
    import socket

    fx, ckpt, idx = _cli_fixture(tmp_path)
    common = ["--beatmap", TOY, "--ckpt", ckpt, "--model", "DiT-S", "--num-classes", "10", "--num-sampling-steps", "8", "--cfg-scale",
              "4.0", "--seed", "1", "--seq-len", "128", "--style-id", str(int(fx["style_id"])), "--beatmap-idx", idx, "--num-variants", "3",
              "--plot-time", "30000", "--precision", "fp32"]
    results = {}
    for name, noise in (("one", "cpu"), ("two", "cpu"), ("one_gpu", "gpu"), ("two_gpu", "gpu")):
        cwd = tmp_path / name
        cwd.mkdir()
        if name.startswith("one"):
            run([os.path.join(ROOT, "sample.py")] + common + ["--noise", noise], str(cwd))
        else:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            env = dict(os.environ, PYTHONPATH=ROOT, OSUD_DIST_BACKEND="gloo", OSUD_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
            r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                                "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "sample.py")] + common + ["--noise", noise],
                               cwd=str(cwd), env=env, capture_output=True, text=True, timeout=900)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        results[name] = torch.load(glob.glob(os.path.join(str(cwd), "results", "*", "result.pt"))[0])
        assert results[name].shape[0] == 3
        assert len(glob.glob(os.path.join(str(cwd), "results", "*", "*.osu"))) == 3
    assert float((results["one"] - results["two"]).abs().max()) <= 1e-3      # pixels
    assert float((results["one_gpu"] - results["two_gpu"]).abs().max()) <= 1e-3
    assert float((results["one"][0, :2] - results["one"][1, :2]).abs().max()) > 1.0   # the variants differ


def test_train_script_on_osu_dataset(tmp_path):
    for track, ident in ((0, "000003"), (1, "000007")):
        folder = tmp_path / "data" / f"Track{track:05d}" / "beatmaps"
        folder.mkdir(parents=True)
        (folder / f"{ident} toy.osu").write_text(open(TOY, encoding="utf-8").read(), encoding="utf-8")
    env_args = [os.path.join(ROOT, "train.py"), "--data-path", str(tmp_path / "data"), "--data-start", "0", "--data-end", "2",
                "--model", "DiT-S", "--num-classes", "10", "--global-batch-size", "8", "--epochs", "1", "--num-workers", "0",
                "--log-every", "2", "--ckpt-every", "4", "--seq-len", "64", "--stride", "16",
                "--results-dir", str(tmp_path / "results")]
    out = run(env_args, str(tmp_path))
    assert "Dataset contains 2 beatmap sets" in out and "Train Loss" in out
    ckpts = glob.glob(os.path.join(str(tmp_path), "results", "*", "checkpoints", "*.pt"))
    assert ckpts, out[-2000:]
    ck = torch.load(sorted(ckpts)[0], map_location="cpu", weights_only=False)
    assert set(ck) >= {"model", "ema", "opt", "args"}
    losses = [float(line.split("Train Loss: ")[1].split(",")[0]) for line in out.splitlines() if "Train Loss: " in line]
    assert all(l == l and l < 10 for l in losses)


def test_train_script_fp8_tier_from_the_cli(tmp_path):
    """BASELINE configs[4]'s arithmetic from the command line (reference surface: train.py:249-259,327 `--use-amp`): `train.py --precision
    fp8` runs every Linear product of the blocks on e4m3 operands with delayed scaling (first step bf16: no amax history yet) and its
    loss follows the bf16 tier's on the same synthetic stream."""
    curves = {}
    for prec in ("bf16", "fp8"):
        out = run([os.path.join(ROOT, "train.py"), "--synthetic", "--model", "DiT-S", "--num-classes", "10", "--global-batch-size", "8",
                   "--epochs", "1", "--steps-per-epoch", "12", "--log-every", "3", "--ckpt-every", "1000", "--seq-len", "128",
                   "--precision", prec, "--results-dir", str(tmp_path / prec)], str(tmp_path))
        assert f"arithmetic tier: {prec}" in out
        curves[prec] = [float(line.split("Train Loss: ")[1].split(",")[0]) for line in out.splitlines() if "Train Loss: " in line]
        assert len(curves[prec]) == 4 and all(l == l and l < 10 for l in curves[prec]), out[-1500:]
    rel = max(abs(a - b) / max(abs(a), 1e-6) for a, b in zip(curves["bf16"], curves["fp8"]))
    print(f"MEASURED train_cli_fp8: logged losses bf16 {curves['bf16']} fp8 {curves['fp8']} (worst relative difference {rel:.3e})")
    assert rel < 0.05
