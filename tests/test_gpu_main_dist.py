"""The boundary's command line on hardware (reference `main_dist.py:132-172`, `utils/trn_dist_utils.py:5-42`):
`python main_dist.py <uid> --dotted.key=value` trains through `TrainStep` (eager step, hipGraph capture, replays),
writes the reference-format checkpoint and resumes from it; and `python bench.py --gpus N` goes through the rank
launcher (`vidsitu_amd/dist_launch.py`): a forced 1-rank RCCL job prints a line that names its rank count, `--gpus 2`
on a 1-GPU box fails loudly instead of printing a 1-rank line.  Each run is a fresh child process, as a user's is."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, extra_env=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(extra_env or {})
    r = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    err = "\n".join(ln for ln in r.stderr.splitlines() if "frame #" not in ln)
    return r.returncode, r.stdout, err[-6000:]


MINI = ["--mdl.sf_mdl_name=slow_fast_mini", "--sf_mdl.DATA.TRAIN_CROP_SIZE=64", "--synth.num_verbs=31",
        "--train.bs=2", "--ds.vsitu.num_ev=2", "--train.lr=1e-3", "--overfit_batch=True"]


def _losses(out):
    m = re.search(r"loss ([-\d.e]+) -> ([-\d.e]+)", out)
    assert m, out
    return float(m.group(1)), float(m.group(2))


def test_main_dist_trains_through_the_graph_step_saves_and_resumes(dev, tmp_path):
    args = ["main_dist.py", "t_md"] + MINI + [f"--misc.tmp_path={tmp_path}", "--steps=6"]
    rc, out, err = _run(args)
    assert rc == 0, out[-3000:] + err
    assert "hipGraph replay" in out and "world 1" in out, out
    first, last = _losses(out)
    assert last < first, out  # one batch, lr 1e-3: the loss goes down
    ck = tmp_path / "models" / "t_md.pth"
    assert ck.exists() and "valid" in out
    # resume: picks the file up, continues from iteration 6
    rc, out2, err = _run(args + ["--train.resume=True"])
    assert rc == 0, out2[-3000:] + err
    assert "resumed" in out2 and "at iteration 6" in out2, out2
    assert _losses(out2)[0] < first
    # the eager loop is the same TrainStep
    rc, out3, err = _run(args + ["--graph=0"])
    assert rc == 0 and "(eager" in out3, out3[-2000:] + err
    e_first, e_last = _losses(out3)
    assert abs(e_first - first) < 1e-6 and abs(e_last - last) < 5e-3 * max(1.0, abs(last))


def test_main_dist_forced_one_rank_rccl_job(dev, tmp_path):
    """VS_FORCE_DIST=1: the parent starts ONE rank through the launcher; the rank initialises RCCL and runs the
    segmented step (one graph per backward segment, bucket all-reduces behind them)."""
    rc, out, err = _run(["main_dist.py", "t_md1"] + MINI + [f"--misc.tmp_path={tmp_path}", "--steps=4", "--num_gpus=1"],
                        {"VS_FORCE_DIST": "1"})
    assert rc == 0, out[-3000:] + err
    assert "hipGraph replay" in out and "segment(s)" in out, out
    first, last = _losses(out)
    assert last < first


def test_bench_through_the_launcher_names_its_ranks(dev):
    rc, out, err = _run(["bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                         "--no-roofline"], {"VS_BENCH_FORCE_DIST": "1"}, timeout=1500)
    assert rc == 0, out[-2000:] + err
    lines = out.strip().splitlines()
    assert len(lines) == 1, out  # ONE JSON line on stdout, nothing else (RCCL's own chatter goes to stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["value"] > 0
    assert "bucket" in line["config"]["grad_allreduce"]


def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly(dev):
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 GPUs")
    rc, out, err = _run(["bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1"])
    assert rc != 0 and out.strip() == "" and "not launching" in err


def test_bench_two_ranks_sharing_the_gpu_runs_the_whole_n_gt_1_path(dev):
    """`python bench.py --gpus 2` end to end on a one-GPU box: the launcher starts two ranks, both on cuda:0 with a gloo
    group (VS_BENCH_SHARE_GPU / VS_BENCH_DIST_BACKEND are test-only switches: RCCL refuses two ranks on one device).
    Exercises what the driver's scaling run will execute on 2 / 4 / 8 GPUs: parameter broadcast, the five segment graphs
    with a bf16 bucket all-reduce behind each, barriers on both sides of the timed region, max over ranks, rank 0's
    instrumented pass while rank 1 waits, ONE JSON line.  The throughput of two ranks sharing a chip means nothing."""
    rc, out, err = _run(["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "2", "--no-cpu-baseline"],
                        {"VS_BENCH_SHARE_GPU": "1", "VS_BENCH_DIST_BACKEND": "gloo"}, timeout=1500)
    assert rc == 0, out[-2000:] + err
    lines = out.strip().splitlines()
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and "gloo" in line["dist_backend"]
    assert line["value"] > 0 and abs(line["value"] - 16 / (line["ms_per_step"] * 1e-3)) < 0.02 * line["value"]
    ga = line["config"]["grad_allreduce"]
    assert "4 bucket" in ga and "bf16" in ga and "overlapped" in ga, ga
    assert line["cpu_baseline"] is None and line["roofline"] is not None


def test_feat_extractor_cli_from_a_trained_checkpoint_and_from_a_caffe2_pickle(dev, tmp_path):
    """`python -m vidsitu_amd.feat_extractor <weights> <name>` (reference `feat_extractor.py:119-176`): features of the
    synthetic videos from (a) the checkpoint `main_dist.py` just wrote and (b) a model-zoo style Caffe2 pickle of the
    trunk (`--is_cu=True`, names converted on load); [E, D] float32 files, readable by `read_frm_feats`."""
    import numpy as np

    common = ["--mdl.sf_mdl_name=slow_fast_mini", "--sf_mdl.DATA.TRAIN_CROP_SIZE=64", "--synth.num_verbs=31",
              "--ds.vsitu.num_ev=2", f"--ds.vsitu.vsitu_frm_feats={tmp_path}/feats", "--train.bsv=2"]
    rc, out, err = _run(["main_dist.py", "t_fx"] + MINI + [f"--misc.tmp_path={tmp_path}", "--steps=2"])
    assert rc == 0, out[-2000:] + err
    ck = tmp_path / "models" / "t_fx.pth"
    rc, out, err = _run(["-m", "vidsitu_amd.feat_extractor", str(ck), "trained_mini", "--n_videos=3"] + common)
    assert rc == 0 and "wrote 6 feature files" in out, out[-2000:] + err
    # (round 5: on by default, --calibrate=0 switches it off; round 6: the calibration clips are the first videos of the
    #  dataset being extracted, not a stand-in batch)
    assert "weight-rounding correction calibrated on the first 2 video(s) of split 'valid'" in out
    files = sorted((tmp_path / "feats" / "trained_mini").glob("*_feats.npy"))
    assert len(files) == 6
    a = np.load(files[0])
    assert a.dtype == np.float32 and a.ndim == 2 and a.shape[0] == 2 and np.isfinite(a).all() and np.abs(a).max() > 0
    # (b) a Caffe2-style pickle written with the model zoo's blob names (tests/test_c2_loading.py's hand-written inverse)
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_c2_loading import _fake_c2_file
    from oracle import slowfast_ref
    from vidsitu_amd.extended_config import get_cfg

    cfg = get_cfg({"mdl.sf_mdl_name": "slow_fast_mini"})
    pk = tmp_path / "zoo.pkl"
    _fake_c2_file(slowfast_ref.VideoTrunk(cfg.sf_mdl), pk, "mini")
    rc, out, err = _run(["-m", "vidsitu_amd.feat_extractor", str(pk), "zoo_mini", "--is_cu=True", "--n_videos=2",
                         "--splits=valid", "--calibrate=0"] + common)
    assert rc == 0 and "Using Caffe2 checkpoint" in out and "wrote 2 feature files" in out, out[-2000:] + err
    assert "weight-rounding correction" not in out


def test_main_dist_vb_arg_row_trains_and_evaluates(dev, tmp_path):
    """The SRL row of the same entry point (`--task_type=vb_arg`, features -> TxEncoder -> fairseq-style decoder,
    `mdl_sf_base.py:793-832`): a few eager TrainStep iterations on the synthetic SRL batch, checkpoint, evaluation by beam
    search (`EvalB_Gen`)."""
    rc, out, err = _run(["main_dist.py", "t_srl", "--task_type=vb_arg", "--mdl.mdl_name=sfpret_txe_txd_vbarg",
                         "--mdl.tx_dec_type=txdec", "--train.bs=2", "--ds.vsitu.num_ev=2", "--train.lr=1e-4",
                         "--overfit_batch=True", "--gen.beam_size=2", "--gen.max_len_b=8", f"--misc.tmp_path={tmp_path}",
                         "--steps=4"], timeout=1500)
    assert rc == 0, out[-3000:] + err
    assert "(eager" in out and "valid" in out, out
    first, last = _losses(out)
    assert last < first, out
    assert (tmp_path / "models" / "t_srl.pth").exists()
