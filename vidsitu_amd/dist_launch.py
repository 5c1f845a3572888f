"""One process per GPU: the launcher behind `bench.py --gpus N` and `main_dist.py`.

What the reference does with `torch.multiprocessing.spawn` from a parent that has already touched CUDA
(`utils/trn_dist_utils.py:32-39` `launch_job`, rendezvous `tcp://localhost:9997`, `:30`), built for the
MI355X pool's rules: the parent NEVER initialises the GPU (it only counts devices), every rank is a fresh
child interpreter started with `subprocess.Popen`, the rendezvous is 127.0.0.1 on a free port picked per
job (two jobs on one node do not collide), rank 0's stdout is relayed verbatim (the bench line), and the
parent exits non-zero as soon as any rank does -- the remaining ranks are then terminated, not left
hanging in a collective.

Pure host logic: importable and testable without a GPU (tests/test_dist_launch.py drives it with a stub
child).
"""
import os
import socket
import subprocess
import sys
import tempfile
import time

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus():
    """Number of GPUs this process could use, WITHOUT touching the HIP / HSA runtime (the launcher parent must stay
    GPU-free: it forks the ranks): the KFD topology nodes that carry SIMDs (CPU nodes have `simd_count 0`), cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when one is set.  Where the topology is not exposed (a sandbox without
    that sysfs tree) the count falls back to `torch.cuda.device_count()` -- hipGetDeviceCount: it loads the runtime but
    creates no context and opens no queue on this image."""
    import glob

    files = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not files:
        import torch

        return int(torch.cuda.device_count())
    n = 0
    for f in files:
        try:
            with open(f) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue  # a node this user may not read is not a usable GPU
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    # The topology lists every GPU of the HOST; a container / cgroup may map only some of them.  What this process can
    # open is a render node: the sysfs count is an upper bound, cut down to the /dev/dri/renderD* nodes that exist
    # here AND are openable by this user (where /dev/dri is absent altogether the bound stays as it is).
    nodes = glob.glob("/dev/dri/renderD*")
    if nodes:
        n = min(n, sum(1 for d in nodes if os.access(d, os.R_OK | os.W_OK)))
    return _apply_visible_env(n, os.environ)


def _apply_visible_env(n_physical, env):
    """Cut a physical device count down by the *_VISIBLE_DEVICES variables in the order the stack applies them:
    ROCR_VISIBLE_DEVICES indexes the PHYSICAL devices (the HSA runtime filters first); HIP_VISIBLE_DEVICES and its alias
    CUDA_VISIBLE_DEVICES re-index the set ROCR left.  (ROCR=2,3 with HIP=0,1 on an 8-GPU node is two devices.)"""
    n = n_physical
    v = env.get("ROCR_VISIBLE_DEVICES")
    if v is not None:
        n = min(n, _count_visible_tokens(v, n_physical))
    n_rocr = n
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is not None:
            n = min(n, _count_visible_tokens(v, n_rocr))
    return n


def _count_visible_tokens(value, n_devices):
    """Devices a *_VISIBLE_DEVICES value selects out of n_devices: integer tokens must be in range and distinct (the
    runtime stops at the first invalid one); anything else (a GPU-<uuid> token) is taken at face value."""
    seen = []
    for t in value.split(","):
        t = t.strip()
        if t == "":
            break
        if t.lstrip("-").isdigit():
            if int(t) < 0 or int(t) >= n_devices or str(int(t)) in seen:
                break
            t = str(int(t))
        seen.append(t)
    return len(seen)


def rank_env(rank, world, port, base=None):
    """Environment of rank `rank` of a `world`-rank single-node job."""
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    # the host driver of the pool only supports dmabuf IPC (RCCL / cross-process device memory)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def world_from_env(requested, env=None):
    """Reconcile `--gpus N` with a launcher-provided WORLD_SIZE.

    -> (world, must_spawn).  WORLD_SIZE set (torch.distributed.run or our own parent): it must equal the
    request, else ValueError -- a silent 1-rank line under `--gpus 8` is a wasted scaling run.  WORLD_SIZE
    unset: spawn `requested` ranks ourselves when that is more than one."""
    env = os.environ if env is None else env
    ws = env.get("WORLD_SIZE")
    if ws is None or ws == "":
        return int(requested), int(requested) > 1
    ws = int(ws)
    if ws != int(requested):
        raise ValueError(f"--gpus {requested} but the launcher set WORLD_SIZE={ws}: refusing to run a "
                         f"{ws}-rank job under an {requested}-GPU label")
    return ws, False


def launch_ranks(world, argv, port=None, poll_s=0.05, grace_s=10.0, out=None, err=None, check_devices=True):
    """Start `world` copies of `argv` (a full command line, e.g. [sys.executable, "bench.py", ...]) with the
    rank environment, relay rank 0's stdout to `out` (default sys.stdout) and every rank's stderr to `err`,
    and return the job's exit code: 0 only if every rank exited 0.  When a rank fails the others get
    SIGTERM, then SIGKILL after `grace_s`.  The caller must not have initialised the GPU."""
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    if check_devices:
        have = visible_gpus()
        if have < world:
            err.write(f"[dist_launch] {world} ranks requested but {have} GPU(s) visible: not launching\n")
            return 2
    port = free_port() if port is None else int(port)
    # rank 0 inherits the parent's stdout (the bench line goes straight through); a caller-provided `out` / `err`
    # object gets the text through temporary files (a pipe nobody drains would block a chatty rank)
    f_out = None if out is sys.stdout else tempfile.TemporaryFile("w+")
    f_err = [None if err is sys.stderr else tempfile.TemporaryFile("w+") for _ in range(world)]
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(
            list(argv), env=rank_env(r, world, port), stdout=f_out if r == 0 else subprocess.DEVNULL,
            stderr=f_err[r], text=True))
    rc = 0
    try:
        live = set(range(world))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    err.write(f"[dist_launch] rank {r} exited with {code}: stopping the job\n")
                    for q in live:
                        procs[q].terminate()
                    deadline = time.time() + grace_s
                    for q in sorted(live):
                        try:
                            procs[q].wait(max(0.1, deadline - time.time()))
                        except subprocess.TimeoutExpired:
                            procs[q].kill()
                    live.clear()
                    break
            if live:
                time.sleep(poll_s)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if f_out is not None:
        f_out.seek(0)
        out.write(f_out.read())
        f_out.close()
    for r, f in enumerate(f_err):
        if f is not None:
            f.seek(0)
            t = f.read()
            f.close()
            if t:
                err.write(f"--- rank {r} stderr ---\n{t}")
    return rc
