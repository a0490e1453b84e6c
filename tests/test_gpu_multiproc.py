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


def _torchrun(nproc, script_args, env_extra, timeout=900, expect_failure=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    if expect_failure:  # (torchrun reports a failed worker with its own exit code 1; the workers' codes are in its log)
        assert r.returncode != 0, r.stdout[-2000:] + r.stderr[-4000:]
        return r.stdout, r.stderr
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


def test_sharded_optimizer_two_ranks_equal_the_allreduce_path():
    """ZeRO-1 between reduce-scatter and all-gather (training.NativeTrainer._backward_sharded), 2 ranks on GPU 0 over gloo:
    replicas bit-identical, weights / EMA / both moments equal to the all-reduce path to 2e-6 (the shard sums are formed in a
    different order), also after gathering the sharded optimizer state for a checkpoint."""
    res = {}
    for mode in ("allreduce", "zero1"):
        with tempfile.TemporaryDirectory() as d:
            _torchrun(2, [os.path.join(ROOT, "tests", "mp_worker.py"), d], dict(OSUD_DIST_BACKEND="gloo", OSUD_TEST_MODE=mode))
            res[mode] = [torch.load(os.path.join(d, f"rank{r}.pt")) for r in (0, 1)]
    z0, z1 = res["zero1"]
    for k in ("flat", "ema", "exp_avg", "exp_avg_sq"):
        assert torch.equal(z0[k], z1[k]), k                                  # replicas in lock step, full state on both
    a0 = res["allreduce"][0]
    scale = float(a0["flat"].abs().max())
    assert float((z0["flat"] - a0["flat"]).abs().max()) <= 2e-6 * scale
    assert float((z0["ema"] - a0["ema"]).abs().max()) <= 2e-6 * scale


def test_rank0_only_checkpoint_with_sharded_optimizer():
    """train.py:285-297 with the sharded optimizer (--zero1): the moments / EMA gather is a collective every rank enters, then rank 0 alone saves and
    everybody meets at the barrier.  checkpoint() on one rank with stale shards refuses instead of hanging in an all-gather."""
    with tempfile.TemporaryDirectory() as d:
        _torchrun(2, [os.path.join(ROOT, "tests", "mp_worker.py"), d], dict(OSUD_DIST_BACKEND="gloo", OSUD_TEST_MODE="zero1_ckpt"), timeout=600)
        ck = torch.load(os.path.join(d, "ckpt.pt"), weights_only=False)
        r0, r1 = torch.load(os.path.join(d, "rank0.pt")), torch.load(os.path.join(d, "rank1.pt"))
    assert ck["refused"] is True
    assert set(ck) >= {"model", "ema", "opt", "scaler", "args"}
    assert torch.equal(r0["exp_avg"], r1["exp_avg"]) and torch.equal(r0["ema"], r1["ema"])  # full state on both ranks
    # the saved optimizer state is the gathered one: the moment of a block tensor owned by rank 1's shard is non-zero
    st = ck["opt"]["state"]
    assert all(float(v["exp_avg"].abs().sum()) > 0 for k, v in st.items() if v["exp_avg"].numel() > 4096)


def test_rccl_is_executed_world_size_one():
    """The process group the 8-GPU job uses -- backend "nccl" = RCCL -- initialised with one rank on this box: the phased backward
    with its per-slice all-reduces, the row exchange of the class table, and the reduce-scatter / all-gather of the sharded
    optimizer all run through RCCL (a 1-rank collective is a copy, but every call, stream hand-over and buffer aliasing rule is
    the real one), in fp32 and with the bf16 wire.  Results equal the no-process-group run."""
    from tests import mp_worker

    want_flat, want_ema = mp_worker.run(0, 1)
    for mode, tol in (("allreduce", 0.0), ("zero1", 0.0), ("zero1_bf16", 2e-3)):
        with tempfile.TemporaryDirectory() as d:
            _torchrun(1, [os.path.join(ROOT, "tests", "mp_worker.py"), d],
                      dict(OSUD_DIST_BACKEND="nccl", OSUD_TEST_MODE=mode, OSUD_TEST_FORCE_PHASED="1"))
            got = torch.load(os.path.join(d, "rank0.pt"))
        err = float((got["flat"] - want_flat).abs().max())
        scale = float(want_flat.abs().max())
        assert err <= max(tol, 2e-6) * scale, (mode, err)


def test_native_rccl_communicator_through_the_c_abi():
    """osud_comm_unique_id / osud_comm_init / osud_allreduce_grads / osud_broadcast_params / osud_reduce_scatter_grads /
    osud_allgather_params (include/osud.h: the collectives of train.py:106,152,257 on the library's own RCCL communicator), with
    the one rank this box has: a 1-rank collective is a copy, but the library load (dlopen of the process's librccl), the
    communicator, every call and the side-stream hand-over are the real ones."""
    from osu_diffusion_amd.comm import NativeComm

    assert NativeComm.rccl_version() > 20000
    c = NativeComm(0, 1, NativeComm.unique_id(), device="cuda:0")
    x = torch.randn(1 << 20, device="cuda:0")
    want = x.clone()
    c.all_reduce_(x)
    h = c.all_reduce_(x, async_op=True)
    h.wait()
    xb = x.to(torch.bfloat16)
    c.all_reduce_(xb)  # bf16 wire
    c.broadcast_(x, 0)
    shard = torch.empty(1 << 18, device="cuda:0")
    c.reduce_scatter(shard, x[: 1 << 18].contiguous())
    full = torch.zeros(1 << 18, device="cuda:0")
    c.all_gather(full, shard, async_op=True).wait()
    torch.cuda.synchronize()
    assert torch.equal(x, want) and torch.equal(xb, want.to(torch.bfloat16)) and torch.equal(full, want[: 1 << 18])
    c.close()


def test_trainer_exchanges_gradients_through_the_native_communicator():
    """native_comm=True under torchrun (one rank, phased backward forced): the per-slice all-reduces, the init broadcast and the
    sharded optimizer's reduce-scatter / all-gather run through libosud's RCCL calls; results equal the run without a process group."""
    from tests import mp_worker

    want_flat, _ = mp_worker.run(0, 1)
    for mode in ("allreduce", "zero1"):
        with tempfile.TemporaryDirectory() as d:
            _torchrun(1, [os.path.join(ROOT, "tests", "mp_worker.py"), d],
                      dict(OSUD_DIST_BACKEND="gloo", OSUD_TEST_MODE=mode, OSUD_TEST_FORCE_PHASED="1", OSUD_TEST_NATIVE_COMM="1"))
            got = torch.load(os.path.join(d, "rank0.pt"))
        err = float((got["flat"] - want_flat).abs().max())
        assert err <= 2e-6 * float(want_flat.abs().max()), (mode, err)


def test_stubbed_exchange_is_the_same_schedule_without_collectives():
    """bench.py's exposed-communication figure is (step with the exchange) - (step with `stub_exchange=True`).  With one rank a
    collective is a copy, so the stubbed schedule must end in the same bits as the real one (RCCL, world 1, phased backward forced) --
    i.e. the stub removes the collectives and nothing else."""
    from tests import mp_worker

    outs = {}
    for stub in ("0", "1"):
        with tempfile.TemporaryDirectory() as d:
            _torchrun(1, [os.path.join(ROOT, "tests", "mp_worker.py"), d],
                      dict(OSUD_DIST_BACKEND="nccl", OSUD_TEST_MODE="allreduce", OSUD_TEST_FORCE_PHASED="1", OSUD_TEST_STUB=stub))
            outs[stub] = torch.load(os.path.join(d, "rank0.pt"))
    assert torch.equal(outs["0"]["flat"], outs["1"]["flat"]) and torch.equal(outs["0"]["ema"], outs["1"]["ema"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL accepts one rank per GPU: needs a node with at least two")
@pytest.mark.parametrize("schedule", ["allreduce", "zero1", "allreduce_native", "zero1_native"])
def test_two_ranks_over_rccl_equal_one_process(schedule):
    """The 8-GPU job's own transport with more than one rank (skipped on the one-GPU box): 2 ranks, one GPU each, backend "nccl" =
    RCCL -- overlapped per-slice all-reduce and the sharded optimizer, through torch's process group and through the library's own
    communicator (`native_comm=True`: its tail all-reduces are joined before the class-table rows travel on torch's group, the case
    ADVICE r3 names).  Replicas bit-identical, equal to one process on the whole batch to summation order."""
    from tests import mp_worker

    mode, _, native = schedule.partition("_")
    with tempfile.TemporaryDirectory() as d:
        _torchrun(2, [os.path.join(ROOT, "tests", "mp_worker.py"), d],
                  dict(OSUD_DIST_BACKEND="nccl", OSUD_TEST_MODE=mode, OSUD_TEST_PER_RANK_DEVICE="1",
                       OSUD_TEST_NATIVE_COMM="1" if native else "0"))
        r0, r1 = torch.load(os.path.join(d, "rank0.pt")), torch.load(os.path.join(d, "rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["ema"], r1["ema"])
    flat, ema = mp_worker.run(0, 1)
    scale = float(flat.abs().max())
    assert float((flat - r0["flat"]).abs().max()) <= 2e-6 * scale
    assert float((ema - r0["ema"]).abs().max()) <= 2e-6 * scale


def test_bench_two_ranks_over_gloo():
    """bench.py's N > 1 code path (rank set-up, barriers, max-over-ranks timing, one JSON line from rank 0) with both ranks on
    GPU 0; once with the all-reduce exchange, once with the sharded optimizer."""
    import json

    for extra in ([], ["--zero1"]):
        out = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "2", "--warmup", "1", "--batch", "16",
                            "--model", "DiT-S", "--no-cpu-baseline", "--no-roofline"] + extra,
                        dict(OSUD_DIST_BACKEND="gloo", OSUD_SINGLE_DEVICE="1"))
        line = [ln for ln in out.splitlines() if ln.startswith("{")][-1]
        res = json.loads(line)
        assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["scaling"] == "weak"
        assert res["config"]["global_batch"] == 32
        # the self-diagnosing part of a multi-GPU line: ranks seen by the backend, the in-run A/B of the exchange schedules, the
        # exposed-communication figure and the bytes on the wire
        mg = res["multi_gpu"]
        assert mg["rccl"]["ranks_seen"] == 2 and mg["rccl"]["world_size"] == 2 and mg["rccl"]["backend"] == "gloo"
        for name in ("allreduce", "zero1", "zero1_no_overlap", "no_exchange"):
            assert mg["schedules"][name]["ms_per_step"] > 0 and mg["schedules"][name]["tokens_per_s"] > 0, (name, mg["schedules"][name])
        assert "skipped" in mg["schedules"]["native_comm"]
        assert mg["fastest_schedule"] in ("allreduce", "zero1", "zero1_no_overlap")
        assert isinstance(mg["exposed_comm_ms_per_step"], float)
        # without a flag the timed schedule is the one MEASURED faster at start-up (training.select_exchange_schedule), the same on both ranks
        if extra:
            assert mg["timed_schedule"]["name"] == "zero1" and "schedule_selected" not in mg
        else:
            sel = mg["schedule_selected"]
            assert sel["name"] in ("allreduce", "zero1") and sel["allreduce_ms"] > 0 and sel["zero1_ms"] > 0 and sel["world_size"] == 2
            assert mg["timed_schedule"]["name"] == sel["name"]
        assert mg["timed_schedule"]["ms_per_step"] > 0
        wb = mg["wire_bytes_per_step"]
        assert wb["dense_slices_payload"] > 0 and wb["class_table_rows_allgather"] < wb["class_table_dense_would_be"] and wb["ring_bytes_sent_per_gpu"] > 0


def test_bench_line_survives_a_diagnostic_that_never_returns():
    """Everything after the timed region of an N > 1 run is collective and has never run on more than one GPU: if one of those
    diagnostics hangs, the watchdog prints the line with the timed result (`diagnostics_incomplete` names the stage) and every rank
    exits with bench.WATCHDOG_EXIT (3): the run FAILS for torchrun / the driver / CI, the partial line is still on stdout.  Two ranks
    over gloo, the first diagnostic replaced by a sleep that never ends."""
    import json

    out, err = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", "train", "--steps", "2", "--warmup", "1", "--batch", "16",
                             "--model", "DiT-S", "--no-cpu-baseline", "--no-roofline", "--simulate-hang", "--watchdog-s", "3"],
                         dict(OSUD_DIST_BACKEND="gloo", OSUD_SINGLE_DEVICE="1"), timeout=300, expect_failure=True)
    assert "exitcode: 3" in err or "exitcode  : 3" in err or "exitcode 3" in err, err[-3000:]  # torchrun's failure report names the workers' code
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["value"] > 0 and res["ms_per_step"] > 0
    assert "multi_gpu" not in res and res["diagnostics_incomplete"]["stage"] == "after the timed region"
