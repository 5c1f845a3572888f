"""Repeats the local-gradient part of tests/two_rank_child.py (two processes sharing cuda:0) under switch settings and
counts failures: which switch makes the flaky head-section gradient go away?"""
import io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from vidsitu_amd import dist_launch

child = os.path.join(ROOT, "tests", "two_rank_child.py")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
configs = [a for a in sys.argv[2:]] or ["A=0"]
os.environ["VS_TWO_RANK_ONLY_LOCAL"] = "1"
for cfg in configs:
    kv = dict(x.split("=", 1) for x in cfg.split(","))
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    fails, msgs = 0, []
    for _ in range(reps):
        out, err = io.StringIO(), io.StringIO()
        rc = dist_launch.launch_ranks(2, [sys.executable, child], out=out, err=err, check_devices=False, grace_s=20.0)
        if rc != 0 or "TWO_RANK_CHILD_OK" not in out.getvalue():
            fails += 1
            m = [ln for ln in err.getvalue().splitlines() if "AssertionError" in ln]
            msgs.append(m[0][:260] if m else err.getvalue()[-300:])
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    print(f"{cfg}: {fails} / {reps} failed", flush=True)
    for m in msgs[:3]:
        print("    ", m, flush=True)
