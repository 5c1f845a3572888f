"""The bench line's schema (the driver parses it): checked on the committed lines of the current round's build
(`profiles/r03_bench_train_v3.json`, `profiles/r03_bench_feat_fwd_v3.json`, produced by `python bench.py` on an MI355X)
and on bench.py's argument surface -- no GPU needed."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_train_line_has_the_contract_fields():
    d = _line("r03_bench_train_v3.json")
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


def test_forward_line_and_cli_surface():
    d = _line("r03_bench_feat_fwd_v3.json")
    assert d["unit"] == "clips/s" and d["value"] > 0 and "feature extractor" in d["metric"]
    assert "parity" in d["config"]  # the measured logit distance to the fp32 oracle, per eval mode
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert re.search(r'add_argument\("%s", type=int, default=\d+' % flag, src), flag
    for env in ("RANK", "LOCAL_RANK"):
        assert f'os.environ.get("{env}"' in src, env
    assert "dist_launch.world_from_env(args.gpus)" in src  # --gpus is reconciled with WORLD_SIZE, never ignored
