"""Fold what the parity tests of a GPU session wrote under gpurun_out/ into the committed profiles/parity_eval.json:
   parity_eval.json            (eval modes, one clip: test_slowfast_r50_one_clip_224_eval_logits_fp32_residual_stream)
   parity_train.json           -> key "train_mode" (test_configs2_bench_batch_train_mode_...)
   parity_eval_robustness.json -> key "robustness" (test_calibrated_shift_parity_over_clips_and_under_a_distribution_shift)
bench.py quotes `config.parity` from the committed file only.  Usage: python tools/merge_parity.py [commit]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path):
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def main():
    dst = os.path.join(ROOT, "profiles", "parity_eval.json")
    rec = _load(dst) or {}
    new = _load(os.path.join(ROOT, "gpurun_out", "parity_eval.json"))
    if new:
        keep = {k: rec[k] for k in ("train_mode", "robustness") if k in rec}
        rec = {**new, **keep}
    for name, key in (("parity_train.json", "train_mode"), ("parity_eval_robustness.json", "robustness")):
        part = _load(os.path.join(ROOT, "gpurun_out", name))
        if part:
            rec[key] = part
    commit = sys.argv[1] if len(sys.argv) > 1 else subprocess.run(
        ["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    import re

    if not re.fullmatch(r"[0-9a-f]{12}", str(rec.get("commit", ""))):
        # (the GPU box has no .git: the test wrote the build tag there; keep it beside the commit of the tree that ran)
        if rec.get("commit"):
            rec["build_tag"] = rec["commit"]
        rec["commit"] = commit
    with open(dst, "w") as f:
        json.dump(rec, f, indent=1)
        f.write("\n")
    print(f"{dst}: keys {sorted(rec)}")


if __name__ == "__main__":
    main()
