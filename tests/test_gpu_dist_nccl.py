"""RCCL entered on hardware (SURVEY.md 8 rows A15 / e; reference `main_dist.py:68-79`,
`utils/trn_dist_utils.py:5-42`): a fresh child process initialises a 1-rank "nccl" process group and runs
bench.py's segmented, bucket-overlapped training step (`vidsitu_amd/train_step.py`); its gradients,
parameters and loss must equal the single-graph, single-process step bit for bit (tests/nccl_child.py).
A multi-rank run needs a multi-GPU node, which the test pool does not have: the driver's SCALE run is the
only place N > 1 executes on hardware; the 2-rank logic is covered on CPU by tests/test_dist_gloo.py."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("model", ["slow_fast_mini:64", "slow_fast_nl_r50_8x8:64"])
def test_one_rank_rccl_overlapped_step_is_bitwise_the_single_graph_step(model, dev):
    env = dict(os.environ)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()), "RANK": "0", "WORLD_SIZE": "1",
                "LOCAL_RANK": "0", "VS_NCCL_TEST_MODEL": model})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "nccl_child.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    err_lines = [ln for ln in r.stderr.splitlines() if "frame #" not in ln]
    tail = (r.stdout[-3000:] + "\n--- stderr (stack frames dropped) ---\n" + "\n".join(err_lines)[-6000:])
    assert r.returncode == 0 and "NCCL_CHILD_OK" in r.stdout, tail
