import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The libraries are normally prebuilt by __graft_entry__.build(); build what is missing."""
    need = [ROOT / "oracle" / "liboracle.so", ROOT / "oracle" / "selection_oracle_cli",
            ROOT / "cuda_selection_criteria_amd" / "lib" / "libselhost.so",
            ROOT / "cuda_selection_criteria_amd" / "lib" / "libselhip.so"]
    if all(p.exists() for p in need):
        return
    subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    subprocess.run(["make", "-C", str(ROOT / "cuda_selection_criteria_amd" / "csrc")], check=True, capture_output=True)


_ensure_built()

# For sets of up to 2 048 genomes (BASELINE configs[1]) the library skips the stage-2 grouping and, for criterion smh_a, runs the whole pass
# in one launch (small_pass_kernel).  Nearly every parity case is that small, so the suite switches both off -- its cases then run the
# kernels the BASELINE-sized sets run -- and tests the small-set defaults on their own (test_small_sets_skip_grouping,
# test_small_pass_one_launch)
import cuda_selection_criteria_amd as _pkg  # noqa: E402
_pkg.Selector.DEFAULT_PARAMS = {"group_min_n": 0, "small_pass": 0}


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def host():
    import cuda_selection_criteria_amd as pkg
    return pkg.host_lib()


def has_gpu():
    try:
        import cuda_selection_criteria_amd as pkg
        return pkg.hip_lib().selhip_device_count() > 0
    except Exception:
        return False
