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
