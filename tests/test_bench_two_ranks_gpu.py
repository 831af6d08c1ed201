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


def _bench(*extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)  # (outside a launcher: bench.py starts its own ranks)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + QUICK + list(extra),
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
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
        assert d["scaling"] == "weak"
        assert cfg["track_exchange"] != "none" and cfg["parallelism"] == "1 rig per GPU"
        assert cfg["events_per_step_per_gpu"] > 100000
    elif split == "camera":
        assert d["scaling"] == "strong" and "camera split" in cfg["parallelism"]
        assert cfg["track_exchange"] != "none"
    else:
        assert d["scaling"] == "strong" and "time-sliced over 2" in cfg["parallelism"]
    if not two_devices:
        assert "dry run over gloo" in err
    assert d["tail_latency"]["allocs_in_timed_passes"] == 0 or split != "rigs"
