"""The bench line's schema (the driver parses it): checked on the committed line of the last measured build
(`profiles/r01_bench_train_v9.json`, produced by `python bench.py` on an MI355X) and on bench.py's argument
surface -- no GPU needed."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read())


def test_train_line_has_the_contract_fields():
    d = _line("r01_bench_train_v9.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "clips/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "bf16"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 8 * d["n_gpus"] / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1


def test_forward_line_and_cli_surface():
    d = _line("r01_bench_feat_fwd_v9.json")
    assert d["unit"] == "clips/s" and d["value"] > 0 and "feature extractor" in d["metric"]
    src = open(os.path.join(ROOT, "bench.py")).read()
    for flag in ("--gpus", "--steps", "--warmup"):
        assert re.search(r'add_argument\("%s", type=int, default=\d+' % flag, src), flag
    for env in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        assert f'os.environ.get("{env}"' in src, env
