"""The host side of rejectWithF_event is lock-free multi-threaded code (fe_host.cpp: helper threads
take RANSAC / LMedS hypotheses in groups, the caller replays them in order; job buffers alternate):
tools/ransac_bench.cpp + fe_host.cpp built with ThreadSanitizer and with AddressSanitizer +
UndefinedBehaviorSanitizer, run over the RANSAC branch, the LMedS branch and the 15-point
boundary with 0..7 helpers.  No report may appear and every helper count must reproduce the
single-thread flags (the bench counts mismatches itself)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tools", "ransac_bench.cpp"), os.path.join(ROOT, "esvio_amd", "csrc", "fe_host.cpp")]
# (target_clones' ifunc resolvers run before the sanitizer runtimes are up: plain functions here)
BASE = ["-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-math-errno", "-DESVIO_NO_SIMD_CLONES", "-pthread",
        "-I" + os.path.join(ROOT, "include")]


def _build(tmp_path, name, flags):
    cxx = shutil.which(os.environ.get("CXX", "g++"))
    if not cxx:
        pytest.skip("no host C++ compiler")
    exe = str(tmp_path / name)
    p = subprocess.run([cxx] + BASE + flags + SRC + ["-o", exe], capture_output=True, text=True, timeout=600)
    if p.returncode != 0 and "sanitize" in p.stderr:
        pytest.skip("sanitizer runtime not installed: " + p.stderr[-200:])
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


@pytest.mark.parametrize("name,flags,markers", [
    ("tsan", ["-fsanitize=thread"], ("WARNING: ThreadSanitizer",)),
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"], ("ERROR: AddressSanitizer", "runtime error")),
])
def test_host_ransac_under_sanitizers(tmp_path, name, flags, markers):
    exe = _build(tmp_path, name, flags)
    for points, outliers, reps in ((120, 0.37, 12), (12, 0.2, 8), (15, 0.1, 12), (300, 0.6, 6)):
        p = subprocess.run([exe, str(points), str(outliers), str(reps)], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"))
        out = p.stdout + p.stderr
        assert p.returncode == 0, out[-2000:]
        for m in markers:
            assert m not in out, out[-3000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("helpers=")]
        assert len(lines) == 6 and all(l.rstrip().endswith("mismatches 0") for l in lines), p.stdout
