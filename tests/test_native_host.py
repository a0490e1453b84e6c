"""CPU (-m "not gpu"): the C-ABI library loads and exports every symbol include/osud.h declares,
the library's host-side schedule code against the reference's golden tables, the model registry /
state-dict / seeded-init contract, and the loud-failure behaviour without a GPU."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest
import torch

from osu_diffusion_amd import _lib
from osu_diffusion_amd.diffusion import create_diffusion, space_timesteps
from osu_diffusion_amd.diffusion import gaussian_diffusion as gd
from osu_diffusion_amd.models import DiT, DiT_models
from tests.helpers import GOLDEN, load

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "osud.h")).read()
    declared = sorted(set(re.findall(r"\b(osud_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 15
    handle = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared if not hasattr(handle, n)]
    assert not missing, f"libosud.so lacks symbols declared in include/osud.h: {missing}"
    assert set(_lib._SIGNATURES) <= set(declared), "python binds a symbol the header does not declare"
    L = _lib.lib()
    assert L.osud_build_arch() == b"gfx950" and L.osud_version() >= 1


def test_option_table_is_the_only_switchboard():
    """include/osud.h: osud_set_option / osud_get_option -- every option the header documents exists with its documented default,
    takes -1 as "back to the default", rejects unknown names and values out of range; and the library's sources read no environment
    variable but OSUD_OPTIONS (initial values of this table) and OSUD_RCCL_LIB."""
    header = open(os.path.join(ROOT, "include", "osud.h")).read()
    rows = re.findall(r"^ \*   ([a-z0-9_]+) +(-?\d+) +\S", header, flags=re.M)
    assert len(rows) >= 10, rows
    for name, default in rows:
        assert _lib.get_option(name) == int(default), name
    _lib.set_option("gemm_tile", 256)
    assert _lib.get_option("gemm_tile") == 256
    _lib.set_option("gemm_tile", -1)
    assert _lib.get_option("gemm_tile") == 0
    with _lib.option("sample_graph", 0):
        assert _lib.get_option("sample_graph") == 0
    assert _lib.get_option("sample_graph") == 1
    with pytest.raises(AssertionError, match="unknown option"):
        _lib.set_option("no_such_option", 1)
    with pytest.raises(AssertionError, match="takes 0..1"):
        _lib.set_option("sample_graph", 7)
    env_reads = set()
    for path in glob.glob(os.path.join(ROOT, "osu_diffusion_amd", "csrc", "*.h*")):
        env_reads |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(path).read()))
    assert env_reads == {"OSUD_OPTIONS", "OSUD_RCCL_LIB"}, env_reads
    for path in glob.glob(os.path.join(ROOT, "osu_diffusion_amd", "**", "*.py"), recursive=True):
        names = set(re.findall(r'environ[^\n]*?"(OSUD_[A-Z0-9_]+)"', open(path).read()))
        assert names <= {"OSUD_LIB", "OSUD_PRECISION"}, (path, names)  # where the library file is; the default tier of DiT(...)


def test_osud_options_environment_values_must_be_numbers():
    """OSUD_OPTIONS="name=value,...": a value that is not a number ("off", the empty string) is reported on stderr and leaves the default in
    place -- it must not be read as 0 (atoi would); a numeric one in range is taken; the phased GEMM loop's option exists with default 1."""
    import subprocess
    import sys

    code = ("from osu_diffusion_amd import _lib; "
            "print(_lib.get_option('sample_graph'), _lib.get_option('gemm_tile'), _lib.get_option('embed_const'), _lib.get_option('gemm_loop'))")
    env = dict(os.environ, OSUD_OPTIONS="sample_graph=off,gemm_tile=,embed_const=0,gemm_loop=0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["1", "0", "0", "0"], r.stdout
    assert "ignoring 'sample_graph=off'" in r.stderr and "ignoring 'gemm_tile='" in r.stderr, r.stderr[-1000:]
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PYTHONPATH=ROOT, OSUD_OPTIONS=""), cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.stdout.split() == ["1", "0", "1", "1"], r.stdout + r.stderr[-500:]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "g1_schedule_*.npz"))))
def test_native_schedule_tables_match_reference(path):
    """osud_sched_create (C++ host code) vs the reference's numpy tables: bit-equal up to libm's
    last-ulp log(), and exactly equal once cast to the fp32 the kernels consume."""
    fx = np.load(path)
    d = create_diffusion(str(fx["respacing"]), noise_schedule=str(fx["noise_schedule"]))
    assert list(d.timestep_map) == list(fx["timestep_map"])
    assert d.num_timesteps == len(fx["betas"])
    for n in ["betas", "alphas_cumprod", "alphas_cumprod_prev", "alphas_cumprod_next", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod",
              "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
              "posterior_mean_coef1", "posterior_mean_coef2"]:
        a, b = getattr(d, n), fx[n]
        assert np.array_equal(a.astype(np.float32), b.astype(np.float32)), n
        assert np.allclose(a, b, rtol=3e-16, atol=0), n


def test_create_diffusion_flags():
    d = create_diffusion("", noise_schedule="squaredcos_cap_v2", use_l1=True)
    assert d.loss_type == gd.LossType.L1 and d.model_var_type == gd.ModelVarType.LEARNED_RANGE
    assert d.model_mean_type == gd.ModelMeanType.EPSILON and d.num_timesteps == 1000
    assert create_diffusion("250").loss_type == gd.LossType.MSE
    assert create_diffusion("", learn_sigma=False).model_var_type == gd.ModelVarType.FIXED_LARGE
    assert create_diffusion("", learn_sigma=False, sigma_small=True).model_var_type == gd.ModelVarType.FIXED_SMALL
    assert create_diffusion("", use_kl=True).loss_type == gd.LossType.RESCALED_KL
    assert sorted(space_timesteps(1000, "ddim50"))[:3] == [0, 20, 40]
    assert sorted(space_timesteps(300, [10, 15, 20]))[:3] == [0, 11, 22]
    with pytest.raises(ValueError):
        space_timesteps(10, "11")
    with pytest.raises(ValueError):
        space_timesteps(1000, "ddim999")
    with pytest.raises(NotImplementedError):
        create_diffusion("", noise_schedule="nope")


def test_sched_create_rejects_bad_betas():
    L = _lib.lib()
    bad = np.array([0.1, 0.0, 0.2])
    use = np.arange(3, dtype=np.int64)
    h = ctypes.c_void_p()
    rc = L.osud_sched_create(bad.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 3,
                             use.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), 3, ctypes.byref(h))
    assert rc == _lib.ERR_ARG and "betas" in _lib.last_error()
    with pytest.raises(AssertionError):
        _lib.check(rc)


def test_registry_and_seeded_init_equal_reference():
    """Same keys, shapes, parameters() order AND the same initial values for the same seed
    (probe values frozen from the reference's DiT-S, torch.manual_seed(0))."""
    fx = load("g9_registry_dit_s")
    assert sorted(DiT_models) == ["DiT-B", "DiT-L", "DiT-S", "DiT-XL"]
    torch.manual_seed(0)
    m = DiT_models["DiT-S"](num_classes=10, context_size=144, class_dropout_prob=0.2)
    assert [k for k, _ in m.named_parameters()] == [str(k) for k in fx["keys"]]
    assert [str(tuple(p.shape)) for p in m.parameters()] == [str(s) for s in fx["shapes"]]
    assert list(m.state_dict().keys()) == [str(k) for k in fx["state_keys"]]
    sd = m.state_dict()
    for k in fx:
        if k.startswith("probe:"):
            assert np.array_equal(sd[k[6:]].flatten()[:8].numpy(), fx[k]), k
    # adaLN-Zero: zero-initialised modulation and output layers (models.py:295-304)
    assert float(m.blocks[3].adaLN_modulation[1].weight.abs().sum()) == 0.0
    assert float(m.final_layer.linear.weight.abs().sum()) == 0.0
    assert not m.xoc_embedder.playfield_size.requires_grad
    assert list(m.parameters())[7] is m.y_embedder.embedding_table.weight  # optimizer-state index 7 quirk


def test_configs():
    for name, (depth, hidden, heads) in {"DiT-XL": (28, 1152, 16), "DiT-L": (24, 1024, 16), "DiT-B": (12, 768, 12),
                                         "DiT-S": (12, 384, 6)}.items():
        if name in ("DiT-XL", "DiT-L"):
            continue  # large: constructing them on CPU is slow; covered by shape arithmetic below
        m = DiT_models[name](num_classes=4, context_size=144)
        assert (m.depth, m.hidden_size, m.num_heads) == (depth, hidden, heads)
    with pytest.raises(ValueError):
        DiT(depth=1, hidden_size=128, num_heads=2, precision="int4")


def test_no_cpu_fallback():
    m = DiT(depth=1, hidden_size=128, num_heads=2, context_size=144, num_classes=4).eval()
    args = (torch.zeros(2, 2, 64), torch.zeros(2, dtype=torch.long), torch.zeros(2, 64), torch.zeros(2, 144, 64),
            torch.zeros(2, dtype=torch.long))
    with torch.no_grad(), pytest.raises(_lib.NativeError, match="no CPU fallback"):
        m(*args)
    with torch.no_grad(), pytest.raises(_lib.NativeError):
        m.forward_with_cfg(*args, 4.0)


def test_deepcopy_does_not_share_native_handle():
    import copy

    m = DiT(depth=1, hidden_size=128, num_heads=2, context_size=144, num_classes=4)
    m._handle = ctypes.c_void_p(1234)  # pretend a handle exists
    e = copy.deepcopy(m)
    assert e._handle is None and e._uploaded == {}
    m._handle = None


def test_generic_diffusion_path_on_cpu_with_plain_callable():
    """The generic (any-callable) path of the diffusion object is host logic and runs anywhere; here
    with a stub model, against the frozen reference outputs."""
    fx = load("g5_step_1000")
    d = create_diffusion("1000", noise_schedule="squaredcos_cap_v2")
    x, t, mout = (torch.from_numpy(fx[k]) for k in ("x", "t", "model_out"))
    model = lambda *_a, **_k: mout  # noqa: E731
    torch.manual_seed(77)
    r = d.p_sample(model, x, t, clip_denoised=True)
    assert float((r["sample"] - torch.from_numpy(fx["p_sample"])).abs().max()) == 0.0
    torch.manual_seed(77)
    r = d.ddim_sample(model, x, t, clip_denoised=True, eta=1.0)
    assert float((r["sample"] - torch.from_numpy(fx["ddim1_sample"])).abs().max()) == 0.0
    assert float((r["pred_xstart"] - torch.from_numpy(fx["ddim1_x0"])).abs().max()) == 0.0


def test_timestep_samplers_match_the_reference():
    """fixture g14_timestep_sampler (the reference's diffusion/timestep_sampler.py under seeded numpy generators)."""
    import numpy as np
    from osu_diffusion_amd.diffusion.timestep_sampler import create_named_schedule_sampler
    from tests.helpers import load

    fx = load("g14_timestep_sampler")
    d = create_diffusion("", noise_schedule="squaredcos_cap_v2")
    uni = create_named_schedule_sampler("uniform", d)
    np.random.seed(11)
    t, w = uni.sample(32, "cpu")
    assert np.array_equal(t.numpy(), fx["uniform_t"]) and np.array_equal(w.numpy(), fx["uniform_w"])
    lsm = create_named_schedule_sampler("loss-second-moment", d)
    np.random.seed(12)
    t, w = lsm.sample(16, "cpu")
    assert np.array_equal(t.numpy(), fx["cold_t"]) and np.array_equal(w.numpy(), fx["cold_w"])
    lsm.update_with_all_losses(list(fx["ts_hist"]), list(fx["loss_hist"]))
    assert np.allclose(lsm.weights(), fx["weights"], rtol=1e-12, atol=0)
    np.random.seed(14)
    t, w = lsm.sample(64, "cpu")
    assert np.array_equal(t.numpy(), fx["warm_t"]) and np.allclose(w.numpy(), fx["warm_w"], rtol=1e-6)
    with pytest.raises(NotImplementedError):
        create_named_schedule_sampler("nope", d)
