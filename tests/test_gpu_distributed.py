"""GPU side of the multi-GPU path on the one GPU a test box has: the sum buffer
aliased as a torch tensor (no copy) and one RCCL collective on it in a world of
one rank, in a fresh process that loads torch before the library, as bench.py
does.  The partition arithmetic and the two-rank reduce are covered on CPU
(test_distributed_cpu.py, gloo)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_sum_buffer_alias_and_rccl_collective():
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "rccl_single_rank.py")
    out = subprocess.run([sys.executable, script], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout[-2000:] + out.stderr[-4000:]
