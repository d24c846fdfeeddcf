import os
import subprocess
import sys

import pytest

# Count-only scans of a filterable query run as ONE kernel (k_sweep_fused) from 4 GiB on and as two
# kernels below; the test texts are small, so the tests ask for the fused kernel at every
# size -- _check() in test_gpu_parity.py runs the two-kernel form explicitly beside it.
os.environ.setdefault("AGH_FUSED_MIN_MB", "0")
# The table engine's fast form (k_tablescan_fast2 / k_tablescan_fast + k_table_replay) at every size, whatever
# AGH_TF_FAST_MIN_MB the build ships (0 since the chunk is chosen by the size of the text); the tests run the
# exact kernel beside it (AGH_FS_FAST=0) and every chunk size (test_table_engine_two_streams_per_lane).
os.environ.setdefault("AGH_TF_FAST_MIN_MB", "0")
# The library reads its environment switches once per query; the tests flip them between scans of one
# query (AGH_FUSED, AGH_FS_FAST, ...): AGH_ENV_LIVE=1 makes every scan call read them again.
os.environ.setdefault("AGH_ENV_LIVE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "ref: needs the compiled reference under oracle/_ref")


def _have_gpu():
    """True iff libagrep_hip.so is built and sees a HIP device (no torch import needed)."""
    try:
        import agrep_amd
        return agrep_amd.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests are skipped on a box without a HIP device (plain `pytest tests` stays
    green on CPU); AGH_REQUIRE_GPU=1 (the GPU box) turns a missing device into failures."""
    # strict on anything that looks like a GPU box: a broken build must fail there, not skip
    if os.environ.get("AGH_REQUIRE_GPU") == "1" or os.path.exists("/dev/kfd"):
        return
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items or _have_gpu():
        return
    skip = pytest.mark.skip(reason="no HIP device / libagrep_hip.so (set AGH_REQUIRE_GPU=1 to fail instead)")
    for it in gpu_items:
        it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is test infrastructure; build it on demand (gcc only)."""
    lib = os.path.join(ROOT, "oracle", "liboracle.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in
            ("agrep_oracle.c", "agrep_oracle.h", "corpus_gen.c", "corpus_gen.h")]
    if (not os.path.exists(lib)) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    yield
