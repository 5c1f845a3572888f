"""The bench line's schema (the driver parses it): checked on the committed lines of the current round's build
(`profiles/r06_bench_train_v1.json`, `profiles/r06_bench_feat_fwd_v1.json`, produced by `python bench.py` on an MI355X)
and on bench.py's argument surface -- no GPU needed."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_train_line_has_the_contract_fields():
    d = _line("r06_bench_train_v1.json")
    for k in ("metric", "value", "unit", "n_gpus", "rccl_ranks", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "clips/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["n_gpus"] == 1 and d["rccl_ranks"] is None  # a single process never initialises RCCL
    assert abs(d["value"] - 8 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "launches_per_step",
              "total_launches_per_step", "all_conv", "bn_all", "families"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["total_launches_per_step"] > r["launches_per_step"] > 0  # the whole step vs the dominant family
    assert 0 < r["all_conv"]["frac_of_bf16_mfma_peak"] < 1
    b = r["bn_all"]
    for k in ("ms_per_step", "launches_per_step", "algorithmic_gb_per_step", "pmc_gb_per_step", "tiny_launches_per_step"):
        assert k in b, k
    assert b["tiny_launches_per_step"] <= b["launches_per_step"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    # round 4: BASELINE configs[1] measured by the same (driver-run) process, and the arithmetic's parity on the train line
    f = d["feat_fwd"]
    for k in ("workload", "clips_per_s", "ms_per_step", "replays", "frac_of_bf16_mfma_peak", "parity"):
        assert k in f, k
    assert "configs[1]" in f["workload"] and f["clips_per_s"] > 0
    assert abs(f["clips_per_s"] - 8 / (f["ms_per_step"] * 1e-3)) < 0.02 * f["clips_per_s"]
    # round 5: the eval leg runs with calibrated BN shifts (north_star's 1e-3 on the arithmetic that is timed), and the
    # reference's own batch [B = 8, E = 5] is in the driver's record
    assert f["parity"]["eval_mode"] == "bf16_calibrated_shift" and f["parity"]["logits_rel_err_vs_fp32_oracle"] < 1e-3
    c40 = d["canonical_b8x5"]
    for k in ("workload", "clips_per_s", "ms_per_step", "replays", "frac_of_bf16_mfma_peak"):
        assert k in c40, k
    assert abs(c40["clips_per_s"] - 40 / (c40["ms_per_step"] * 1e-3)) < 0.02 * c40["clips_per_s"]
    assert r["total_launches_per_step"] <= 880 and r["bn_all"]["tiny_launches_per_step"] <= 240
    par = d["config"]["parity"]
    assert par["north_star"] == 1e-3 and 0 < par["logits_rel_err_vs_fp32_oracle"] < 1e-2
    assert "measured_at_commit" in par and "not re-measured in this run" in par["source"]
    # round 6: the training line carries the TRAIN-mode distance beside the eval one, and names the metric
    assert par["metric"].startswith("max|logit") and 0 < par["train_mode"]["logits_rel_err_vs_fp32_oracle"] < 2e-2
    assert par["train_mode"]["metric"] == par["metric"] and "configs2" in par["train_mode"]["source"]
    assert "calibrated_shift_over_clips" in f["parity"] and f["parity"]["metric"] == par["metric"]


def test_parity_numbers_come_from_the_file_the_parity_test_writes():
    """bench.py quotes no remembered numbers: `config.parity` is read from profiles/parity_eval.json (written by
    tests/test_gpu_parity_full.py, committed with the commit it was measured at) and is None without it."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "parity_eval.json" in src and "3345" not in src and "3.02e-3" not in src
    with open(os.path.join(ROOT, "profiles", "parity_eval.json")) as fh:
        rec = json.load(fh)
    e = rec["logits_rel_err_vs_fp32_oracle"]
    assert set(e) == {"bf16", "fp32_residual_stream", "split_bf16_weights", "bf16_calibrated_shift"}
    assert e["split_bf16_weights"] < rec["north_star"] < e["bf16"]
    # round 5: the mode bench.py's eval legs run (calibrated BN shifts, no cost per forward) meets north_star's 1e-3
    assert e["bf16_calibrated_shift"] < rec["north_star"]
    assert re.fullmatch(r"[0-9a-f]{12}", rec["commit"]), "parity_eval.json carries the commit it was measured at"
    assert "calibrate_eval(mdl" in src and "canonical_b8x5_leg" in src
    # round 6: a TRAINING line carries the train-mode distance beside the eval one, and every note names its metric (the
    # tolerance is relative to max |logit|); the calibrated path's spread over clips / distributions rides with eval lines
    assert rec["train_mode"]["logits_rel_err_vs_fp32_oracle"] < 2e-2 and "configs2" in rec["train_mode"]["source"]
    cases = rec["robustness"]["cases"]
    assert cases["calibrated_on_noise/noise"]["max"] < rec["north_star"]
    assert cases["calibrated_on_video/video"]["median"] < rec["north_star"] < cases["uncalibrated/video"]["max"]
    import sys
    sys.path.insert(0, ROOT)
    import bench
    tr = bench.eval_parity_note(False, train=True)
    assert tr["metric"].startswith("max|logit") and tr["eval_mode"] == "bf16"
    assert tr["train_mode"]["logits_rel_err_vs_fp32_oracle"] == rec["train_mode"]["logits_rel_err_vs_fp32_oracle"]
    assert tr["train_mode"]["metric"] == tr["metric"] and "TRAINING" in tr["note"]
    ev = bench.eval_parity_note(True)
    assert "train_mode" not in ev and ev["eval_mode"] == "bf16_calibrated_shift"
    assert set(ev["calibrated_shift_over_clips"]["calibrated_on_video/video"]) == {"max", "median"}


def test_forward_line_and_cli_surface():
    d = _line("r06_bench_feat_fwd_v1.json")
    assert d["unit"] == "clips/s" and d["value"] > 0 and "feature extractor" in d["metric"]
    assert "parity" in d["config"]  # the measured logit distance to the fp32 oracle, per eval mode
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert re.search(r'add_argument\("%s", type=int, default=\d+' % flag, src), flag
    for env in ("RANK", "LOCAL_RANK"):
        assert f'os.environ.get("{env}"' in src, env
    assert "dist_launch.world_from_env(args.gpus)" in src  # --gpus is reconciled with WORLD_SIZE, never ignored
