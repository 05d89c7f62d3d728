"""bench.py's multi-rank path on real hardware with the one GPU a test box has: launched exactly as the driver
launches N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 ...`), so RCCL initialises on the
device, every step issues the asynchronous `all_gather_into_tensor` of the pitch track on RCCL's stream with the
double-buffered waits, and the gathered-track check runs (bench.py `use_dist`).  The same workload is
then run without the launcher and the two `value`s are compared: the exchange overlaps the next step's kernels, so
one rank with it must be as fast as one rank without it."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(args, launcher):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", "1"] + args
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0]), r.stderr


def test_bench_under_launcher_two_minutes_rccl_world1():
    """2 minutes of audio: RCCL init on device_id, the async all-gather, works[k-2].wait(), torch.equal of the gathered
    track against the rank's own — on the hardware."""
    line, err = _bench(["--steps", "20", "--warmup", "5", "--minutes", "2", "--no-cpu-baseline", "--no-supplementary"], True)
    assert line["outputs_ok"] is True and line["n_gpus"] == 1 and line["steps"] == 20
    assert line["exchange"]["backend"] == "nccl" and line["exchange"]["collective"] == "all_gather_into_tensor"
    assert line["exchange"]["gathered_equals_local"] is True
    log = os.environ.get("MX_RCCL_LOG")
    if log:  # kept as profiles/rccl_r03_world1.log
        with open(log, "a") as f:
            f.write(json.dumps(line) + "\n")
            f.write(err[-4000:] + "\n")


def test_bench_under_launcher_matches_plain_run_full_hour():
    """The headline workload (60 min, two launches per step) with and without the launcher: `value` within 3 %."""
    common = ["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-supplementary"]
    plain, _ = _bench(common, False)
    dist, err = _bench(common, True)
    assert plain["outputs_ok"] and dist["outputs_ok"] and dist["n_gpus"] == 1
    assert "exchange" in dist and "exchange" not in plain
    ratio = dist["value"] / plain["value"]
    log = os.environ.get("MX_RCCL_LOG")
    if log:
        with open(log, "a") as f:
            f.write(json.dumps({"plain": plain["value"], "under_launcher": dist["value"], "ratio": ratio,
                                "plain_ms": plain["ms_per_step"], "launcher_ms": dist["ms_per_step"]}) + "\n")
            f.write(json.dumps(dist) + "\n")
    assert 0.97 <= ratio <= 1.03, (plain["value"], dist["value"])
