"""The rank launcher behind `bench.py --gpus N` and `main_dist.py` (`vidsitu_amd/dist_launch.py`; the reference:
`launch_job`, utils/trn_dist_utils.py:32-39) -- host logic only, driven with a stub child: rank environment, free
rendezvous port, rank 0's stdout relayed, any failing rank fails the job and stops the others, and the
`--gpus` / WORLD_SIZE reconciliation that keeps bench.py from printing an N'-rank line under an N-GPU label."""
import io
import json
import os
import subprocess
import sys
import time

import pytest

from vidsitu_amd import dist_launch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r"""
import json, os, sys, time
rank = int(os.environ["RANK"])
mode = sys.argv[1]
if mode == "env":
    if rank == 0:
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}))
    else:
        print("rank %d must not reach the parent's stdout" % rank)
elif mode == "fail1":
    if rank == 1:
        sys.stderr.write("rank 1 going down\n")
        sys.exit(7)
    time.sleep(60)  # would hang in a collective: the launcher must stop it
elif mode == "chatty":
    if rank == 0:
        sys.stdout.write("x" * 300000 + "\n")
"""


def _run(tmp_path, world, mode, **kw):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.time()
    rc = dist_launch.launch_ranks(world, [sys.executable, str(stub), mode], out=out, err=err, check_devices=False, **kw)
    return rc, out.getvalue(), err.getvalue(), time.time() - t0


def test_rank_environment_and_stdout_relay(tmp_path):
    rc, out, err, _ = _run(tmp_path, 3, "env")
    assert rc == 0, err
    env = json.loads(out)
    assert env["RANK"] == "0" and env["LOCAL_RANK"] == "0" and env["WORLD_SIZE"] == "3"
    assert env["MASTER_ADDR"] == "127.0.0.1" and 1024 < int(env["MASTER_PORT"]) < 65536
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "must not reach" not in out


def test_two_jobs_get_two_ports(tmp_path):
    ports = {json.loads(_run(tmp_path, 1, "env")[1])["MASTER_PORT"] for _ in range(3)}
    assert len(ports) >= 2  # free ports picked per job (the reference's fixed 9997 collides)
    rc, out, _, _ = _run(tmp_path, 1, "env", port=23456)
    assert rc == 0 and json.loads(out)["MASTER_PORT"] == "23456"


def test_a_failing_rank_fails_the_job_and_stops_the_others(tmp_path):
    rc, out, err, dt = _run(tmp_path, 3, "fail1", grace_s=5.0)
    assert rc == 7
    assert dt < 30, "the surviving ranks were not terminated"
    assert "rank 1 exited with 7" in err and "rank 1 going down" in err


def test_large_rank0_output_does_not_block(tmp_path):
    rc, out, _, _ = _run(tmp_path, 2, "chatty")
    assert rc == 0 and len(out) > 300000


def test_refuses_more_ranks_than_devices():
    err = io.StringIO()
    have = dist_launch.visible_gpus()
    rc = dist_launch.launch_ranks(have + 1, [sys.executable, "-c", "raise SystemExit(0)"], err=err)
    assert rc == 2 and "not launching" in err.getvalue()


def test_world_size_reconciliation():
    assert dist_launch.world_from_env(1, {}) == (1, False)
    assert dist_launch.world_from_env(8, {}) == (8, True)  # no launcher: bench.py spawns the ranks itself
    assert dist_launch.world_from_env(8, {"WORLD_SIZE": "8"}) == (8, False)  # torch.distributed.run did
    with pytest.raises(ValueError, match="WORLD_SIZE=1"):
        dist_launch.world_from_env(8, {"WORLD_SIZE": "1"})
    with pytest.raises(ValueError):
        dist_launch.world_from_env(1, {"WORLD_SIZE": "2"})


def test_bench_gpus_flag_fails_loudly_without_the_devices():
    """`python bench.py --gpus N` with fewer than N GPUs visible (here: none) exits non-zero before any GPU work --
    round 2's bench.py ignored the flag and printed a 1-rank line."""
    env = {k: v for k, v in os.environ.items() if k not in dist_launch.RANK_ENV}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "not launching" in r.stderr and r.stdout.strip() == ""
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "1", "0", "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_visible_device_tokens_are_counted_like_the_runtime_does():
    """`*_VISIBLE_DEVICES` is more than a comma count -- out-of-range, repeated or empty tokens end the list
    (the runtime stops there); UUID tokens are taken at face value."""
    f = dist_launch._count_visible_tokens
    assert f("0,1,2", 8) == 3
    assert f("0,1,9", 8) == 2          # 9 does not exist: the runtime stops at it
    assert f("3,3", 8) == 1            # a repeated index ends the list
    assert f("", 8) == 0 and f("0,,1", 8) == 1
    assert f("-1", 8) == 0
    assert f("GPU-abcdef,GPU-123456", 8) == 2
    assert f("1, 2", 4) == 2


def test_visible_gpus_is_bounded_by_what_the_environment_selects(monkeypatch):
    n = dist_launch.visible_gpus()
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert dist_launch.visible_gpus() == 0
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3,4,5,6,7,8,9")
    assert dist_launch.visible_gpus() <= n


def test_rocr_filters_the_physical_devices_and_hip_reindexes_what_is_left():
    """ROCR_VISIBLE_DEVICES addresses physical devices, HIP_ / CUDA_VISIBLE_DEVICES index the ROCR-filtered
    set -- ROCR=2,3 with HIP=0,1 on an 8-GPU node is two devices, not zero."""
    f = dist_launch._apply_visible_env
    assert f(8, {}) == 8
    assert f(8, {"ROCR_VISIBLE_DEVICES": "2,3", "HIP_VISIBLE_DEVICES": "0,1"}) == 2
    assert f(8, {"ROCR_VISIBLE_DEVICES": "2,3", "HIP_VISIBLE_DEVICES": "2,3"}) == 0   # HIP indices past the ROCR set
    assert f(8, {"ROCR_VISIBLE_DEVICES": "6,7", "CUDA_VISIBLE_DEVICES": "1"}) == 1
    assert f(8, {"ROCR_VISIBLE_DEVICES": "0,1,2,3", "HIP_VISIBLE_DEVICES": "0,1,2", "CUDA_VISIBLE_DEVICES": "0"}) == 1
    assert f(2, {"ROCR_VISIBLE_DEVICES": "2,3"}) == 0                                  # physical index out of range
    assert f(8, {"HIP_VISIBLE_DEVICES": "4,5,6"}) == 3
