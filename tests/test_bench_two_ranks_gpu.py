"""The N > 1 path of bench.py really executed (round-3 review: the first run of `bench.py --gpus N` would
have been the driver's): two ranks on the one-GPU test box — the collectives then go over gloo and the line
says `devices_used: 1`, a dry run of the path, not a scaling measurement — for the three shardings of
SURVEY 8e: one rig per rank, left/right camera split (BASELINE C4), one stream time-sliced (C5)."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUICK = ["--steps", "4", "--warmup", "2", "--repeats", "1", "--cpu-frames", "0", "--cpu-procs", "0", "--no-profile-pass",
         "--no-sae-pass", "--no-host-pass"]


def _bench(*extra, n=2):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)  # (outside a launcher: bench.py starts its own ranks)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)] + QUICK + list(extra),
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must carry exactly one line: %r" % p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


@pytest.mark.parametrize("split", ["rigs", "camera", "time"])
def test_bench_two_ranks_on_one_gpu(split):
    import torch
    two_devices = torch.cuda.device_count() >= 2
    d, err = _bench("--split", split)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["warmup"] == 2
    assert d["devices_used"] == (2 if two_devices else 1)
    assert math.isfinite(d["ms_per_step"]) and d["ms_per_step"] > 0 and d["value"] > 0
    assert d["unit"] == "Mevents/s" and d["roofline"] is None and d["cpu_baseline"] is None
    cfg = d["config"]
    if split == "rigs":
        assert d["scaling"] == "weak" and d["scaling_note"].startswith("replicas")
        assert cfg["track_exchange"] != "none" and cfg["parallelism"] == "1 rig per GPU"
        assert cfg["events_per_step_per_gpu"] > 100000
    elif split == "camera":
        assert d["scaling"] == "strong" and "camera split" in cfg["parallelism"]
        assert cfg["track_exchange"] != "none"
    else:
        assert d["scaling"] == "strong" and "time-sliced over 2" in cfg["parallelism"]
        assert d["scaling_note"].startswith("capability, not a speed-up")
    if not two_devices:
        assert "dry run over gloo" in err
    assert d["tail_latency"]["allocs_in_timed_passes"] == 0 or split != "rigs"


@pytest.mark.parametrize("split", ["rigs", "time"])
def test_bench_eight_ranks_dry_run(split):
    """the command line the driver will issue on an 8-GPU node (`bench.py --gpus 8`, and C5's `--split time`), here with
    all eight ranks on the test box's device(s) over gloo: the launcher, each rank bound to its share of the cores,
    helper threads sized by that share (never more threads than CPUs), the track exchange / the plane exchange with
    world size 8, ONE JSON line on fd 1 that says what it is"""
    import torch
    n_dev = torch.cuda.device_count()
    d, err = _bench("--split", split, n=8)
    assert d["n_gpus"] == 8 and d["devices_used"] == min(8, n_dev) and d["steps"] == 4
    assert math.isfinite(d["ms_per_step"]) and d["ms_per_step"] > 0 and d["value"] > 0
    cfg = d["config"]
    usable = len(os.sched_getaffinity(0))
    assert 1 <= cfg["host_threads"] <= max(1, usable // 8) or cfg["host_threads"] <= 8 and usable >= 64
    if usable // 8 < 4:  # (a small share: blocking helpers, no launch thread)
        assert cfg["helper_idle_spin_us"] == 30 and not cfg["launch_thread"]
    if split == "rigs":
        assert d["scaling"] == "weak" and d["scaling_note"].startswith("replicas")
        assert cfg["parallelism"] == "1 rig per GPU" and cfg["track_exchange"] != "none"
        assert cfg["events_per_step_per_gpu"] > 100000
        # value is the whole job: eight rigs' events over the slowest rank's time
        assert d["value"] * 1e6 * d["ms_per_step"] * 1e-3 > 7.5 * cfg["events_per_step_per_gpu"]
    else:
        assert d["scaling"] == "strong" and "time-sliced over 8" in cfg["parallelism"]
        assert d["scaling_note"].startswith("capability, not a speed-up")
    if n_dev < 8:
        assert "dry run over gloo" in err
