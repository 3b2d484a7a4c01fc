"""bench.py's process handling, checked without a GPU: `python bench.py --gpus N` (N > 1, no torch.distributed environment) must start
its own ranks as a child `torch.distributed.run` and hand the child's exit status back (it used to stop with an error
before launching anything); without an MI355X the ranks refuse to run -- there is no CPU fallback -- so the status is non-zero."""
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.timeout(240)
def test_bench_starts_its_own_ranks_and_propagates_their_status():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: this is the no-GPU behaviour check")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=200)
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "must be launched with torch.distributed.run" not in r.stderr
    assert r.stderr.count("needs an MI355X") >= 2                       # both ranks were started
    r1 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=200)
    assert r1.returncode != 0 and "needs an MI355X" in r1.stderr       # N = 1: same refusal, no spawn
