"""bench.py's stdout line is what the driver parses: it must stay a LINE (round 5's 22.5 KB report was not parsed and the round lost
its driver-recorded headline).  `compact_line` is run here on the largest full reports the repo holds (a committed single-GPU run with
every optional object, and the same report with a world-8 `multi_gpu` object) and must stay below the limit with every field the
measurement contract names."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _full_report():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default.json")))


def _multi_gpu_stub(world=8):
    legs = ("allreduce", "zero1", "zero1_no_overlap", "native_comm", "no_exchange")
    return {"rccl": {"backend": "nccl (RCCL)", "world_size": world, "ranks_seen": world, "version": 22203, "torch_nccl_version": "2.22.3"},
            "schedules": {n: {"ms_per_step": 24.123, "tokens_per_s": 10_867_000.1} for n in legs[:-2]} | {
                "native_comm": {"error": "RuntimeError: " + "x" * 280}, "no_exchange": {"ms_per_step": 23.9, "tokens_per_s": 10_900_000.0}},
            "schedules_measured_over": {"steps": 20, "warmup": 3, "note": "n" * 120},
            "fastest_schedule": "zero1", "schedule_selected": {"name": "zero1", "allreduce_ms": 25.1, "zero1_ms": 24.2},
            "exposed_comm_ms_per_step": 0.223, "timed_schedule": {"name": "allreduce", "ms_per_step": 24.5},
            "wire_bytes_per_step": {"dense_slices_payload": 519_663_504, "class_table_rows_allgather": 6_307_840,
                                    "class_table_dense_would_be": 161_805_312, "ring_bytes_sent_per_gpu": 914_930_492, "slices": 15},
            "predicted_comm_ms_per_step": {"link_GBps": 153.0, "ring_allreduce_unoverlapped": 5.98, "mesh_reduce_scatter_allgather_unoverlapped": 0.854,
                                           "note": "n" * 150}}


def test_line_is_a_line_with_every_contract_field():
    import bench

    res = _full_report()
    assert len(json.dumps(res)) > 20_000  # (the report this was cut from)
    line = bench.compact_line(res)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 4096, len(line)
    out = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in out, k
    assert out["value"] == res["value"] and out["ms_per_step"] == res["ms_per_step"] and out["config"]["workload"].startswith("train.py step: DiT-B")
    rf = out["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes", "avg_launch_us", "measured", "lowest_fraction_top_family"):
        assert k in rf, k
    assert rf["frac"] == res["roofline"]["frac"] and abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3
    assert rf["non_mfma_ms_per_step"] == res["roofline"]["per_family"]["non_mfma_ms_per_step"]
    cb = out["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample"} and cb["kind"] == "port"
    sp = out["sampling"]
    for k in ("value", "tier", "ms_per_step", "meets_1e-3", "held_by_every_reference_fixture", "cpu_baseline", "mfma_frac"):
        assert k in sp, k
    assert sp["value"] == res["sampling"]["value"] and sp["tier"] == "fp16f8" and sp["meets_1e-3"] is True
    assert sp["also"]["bf16"]["meets_1e-3"] is False  # the faster tier outside the tolerance is never the credited value
    assert set(out["xl"]) == {"bf16", "fp8"}


def test_multi_gpu_line_stays_below_the_limit():
    import bench

    res = _full_report()
    res["n_gpus"] = 8
    res["config"]["parallelism"] = "dp8: flat fp32 gradient arena, per-slice RCCL all-reduces overlapped with the phased backward" + " (more words)" * 20
    res["multi_gpu"] = _multi_gpu_stub()
    res["diagnostics_incomplete"] = {"stage": "multi_gpu.schedules.native_comm", "after_s": 900.0, "note": "n" * 200}
    line = bench.compact_line(res)
    assert len(line) < bench.LINE_LIMIT, len(line)
    out = json.loads(line)
    for k in ("metric", "value", "ms_per_step", "roofline", "cpu_baseline", "sampling"):
        assert k in out
    mg = out.get("multi_gpu")
    assert mg is not None and mg["rccl"]["ranks_seen"] == 8 and mg["schedules"]["zero1"]["ms_per_step"] > 0
    assert mg["schedule_selected"]["name"] == "zero1" and len(mg["schedules"]["native_comm"]["error"]) <= 80


def test_emit_prints_one_stdout_line_and_keeps_the_detail(tmp_path, capsys, monkeypatch):
    import bench

    monkeypatch.setattr(bench, "DETAIL_FILE", str(tmp_path / "bench_detail.json"))
    res = _full_report()
    bench.emit(res)
    cap = capsys.readouterr()
    lines = [ln for ln in cap.out.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < bench.LINE_LIMIT and json.loads(lines[0])["value"] == res["value"]
    assert json.load(open(tmp_path / "bench_detail.json")) == res
    assert cap.err.startswith("bench_detail {")
