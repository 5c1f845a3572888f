"""Two ranks on hardware (SURVEY.md 8 rows A15 / e; reference `main_dist.py:68-79`): the distributed training step with
world size 2 -- HIP-produced gradients of two different clip shards summed across two processes, segment graphs
replayed with a bucket all-reduce behind each, 1 / world folded into Adam -- against the single-process sum
(tests/two_rank_child.py).  The pool's boxes have one GPU, so both ranks share cuda:0 and the transport is gloo (RCCL
refuses two ranks on one device; the RCCL entry itself is tests/test_gpu_dist_nccl.py, the 8-GPU run is the driver's)."""
import io
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("model", ["slow_fast_mini:64", "slow_fast_nl_r50_8x8:64"])
def test_two_rank_step_sums_hip_gradients_across_processes(model, dev, monkeypatch):
    from vidsitu_amd import dist_launch

    monkeypatch.setenv("VS_TWO_RANK_MODEL", model)
    out, err = io.StringIO(), io.StringIO()
    rc = dist_launch.launch_ranks(2, [sys.executable, os.path.join(HERE, "two_rank_child.py")], out=out, err=err,
                                  check_devices=False, grace_s=20.0)
    err_lines = [ln for ln in err.getvalue().splitlines() if "frame #" not in ln]
    assert rc == 0 and "TWO_RANK_CHILD_OK" in out.getvalue(), out.getvalue()[-2000:] + "\n".join(err_lines)[-6000:]
