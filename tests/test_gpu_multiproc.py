"""Data-parallel training across processes, on the real GPU path: 2 ranks (both on GPU 0, gloo transport) must end
bit-identical to each other and equal to one process that trains on the concatenated batch (train.py:152,255-261:
DDP averages gradients; the loss is a batch mean)."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_equal_one_process_on_the_full_batch():
    from tests import mp_worker

    with tempfile.TemporaryDirectory() as d:
        env = dict(os.environ, OSUD_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "mp_worker.py"), d]
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        r0, r1 = torch.load(os.path.join(d, "rank0.pt")), torch.load(os.path.join(d, "rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["ema"], r1["ema"])  # replicas stay in lock step
    # one process, whole batch, starting from rank 0's weights (what the broadcast hands to everybody)
    flat, ema = mp_worker.run(0, 1)
    scale = float(flat.abs().max())
    assert float((flat - r0["flat"]).abs().max()) <= 2e-6 * scale, float((flat - r0["flat"]).abs().max())
    assert float((ema - r0["ema"]).abs().max()) <= 2e-6 * scale
    start = mp_worker.build(100)
    from osu_diffusion_amd.training import ParamArena
    assert float((ParamArena(start).flat.cpu() - flat).abs().max()) > 1e-4  # the two steps really moved the weights
