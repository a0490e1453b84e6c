import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "selfcheck: compares two schedules / kernels of this library with each other (no oracle "
                                       "involved): collected after every test that checks against the reference's fixtures")


# Collection order: parity first.  A run under `-x` stops at the first failure, so the tests that carry the parity claim -- native
# path against the golden fixtures generated from the reference, and against the oracle -- come before everything else, the
# end-to-end loops of every tier right behind the forward pass; tests that compare one schedule or kernel of this library with
# another (`selfcheck`) come last: one of those failing must never hide a parity test again.
_FILE_ORDER = ["test_oracle_golden", "test_bench_line", "test_native_host", "test_windows", "test_curves", "test_beatmap", "test_toy_dataset",
               "test_distributed_cpu", "test_gpu_forward", "test_gpu_x3", "test_gpu_h8", "test_gpu_f16", "test_inpaint", "test_gpu_refine",
               "test_gpu_train", "test_gpu_gelu_code", "test_gpu_fullsize", "test_gpu_w8", "test_gpu_fp8", "test_gpu_configs", "test_gpu_exchange", "test_gpu_scripts", "test_gpu_multiproc",
               "test_gpu_phased"]


def pytest_collection_modifyitems(config, items):
    def key(pair):
        idx, item = pair
        name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        rank = _FILE_ORDER.index(name) if name in _FILE_ORDER else len(_FILE_ORDER)
        return (1 if item.get_closest_marker("selfcheck") else 0, rank, idx)

    items[:] = [item for _, item in sorted(enumerate(items), key=key)]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def osud_option():
    """`osud_option(name, value)`: set one of the library's process-wide options (include/osud.h: osud_set_option) for the rest of
    the test; every option touched goes back to its default afterwards."""
    from osu_diffusion_amd import _lib

    touched = []

    def set_(name, value):
        touched.append(name)
        _lib.set_option(name, value)

    yield set_
    for name in touched:
        _lib.set_option(name, -1)
