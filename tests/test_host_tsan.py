"""ThreadSanitizer over EVERY thread the library starts, without a GPU: the host side (fe_api / fe_track /
fe_stages / fe_image / fe_evstage / fe_host .cpp) built with g++ -fsanitize=thread against tests/hipstub — a HIP
runtime that runs everything at once on the calling thread, streams and events carrying HIP's ordering as atomics,
and a fake device whose select / LK produce enough for rejectWithF_event to run its RANSAC on the helper pool — and
driven through the C ABI with the soak's toggling pattern (tests/hipstub/drive.cpp: replay from pageable memory
with 1..4 batches announced, the launch thread switched on and off mid-stream, 0..7 helpers, lazy on / off, plain
calls, refused calls, esvio_fe_reset with batches in flight, handles created and destroyed).

No report may appear (tests/hipstub/tsan.supp suppresses the one designed race: the idempotent re-do of a staging
chunk a straggling helper holds) and the driver must finish — a hang is a failure too: on the commit before round
4's launcher fix (e4c4950^: job numbers not reset when the launch thread is restarted) this driver hangs for every
seed; profiles/r05_tsan_history.txt has that run, and the run on the commit before round 4's staging fix (f0d8679^),
whose window — a DMA still in flight when the batch is closed — a device that completes at once does not open."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "hipstub")
CSRC = os.path.join(ROOT, "esvio_amd", "csrc")
HOST_SOURCES = ["fe_api.cpp", "fe_track.cpp", "fe_stages.cpp", "fe_image.cpp", "fe_evstage.cpp", "fe_host.cpp"]


def build_driver(tmp_path, tree=ROOT, name="drive_tsan", sanitize="thread"):
    cxx = shutil.which(os.environ.get("CXX", "g++"))
    if not cxx:
        pytest.skip("no host C++ compiler")
    csrc = os.path.join(tree, "esvio_amd", "csrc")
    gen = str(tmp_path / "kernel_stubs.cpp")
    subprocess.check_call([sys.executable, os.path.join(STUB, "gen_kernel_stubs.py"), os.path.join(csrc, "fe_kernels.h"), gen],
                          stdout=subprocess.DEVNULL)
    exe = str(tmp_path / name)
    cmd = [cxx, "-O1", "-g", "-std=c++17", "-fsanitize=" + sanitize, "-ffp-contract=off", "-fno-math-errno",
           "-DESVIO_NO_SIMD_CLONES", "-pthread", "-I" + STUB, "-I" + os.path.join(tree, "include"), "-I" + csrc]
    cmd += [os.path.join(csrc, f) for f in HOST_SOURCES]
    cmd += [os.path.join(STUB, "hip_stub.cpp"), os.path.join(STUB, "fake_device.cpp"), gen, os.path.join(STUB, "drive.cpp"),
            "-ldl", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    if p.returncode != 0 and "sanitize" in p.stderr and "cannot find" in p.stderr:
        pytest.skip("sanitizer runtime not installed: " + p.stderr[-200:])
    assert p.returncode == 0, p.stderr[-3000:]
    return exe


def run_driver(exe, seed, frames, timeout, **extra_env):
    env = dict(os.environ, TSAN_OPTIONS="suppressions=%s halt_on_error=0" % os.path.join(STUB, "tsan.supp"), **extra_env)
    return subprocess.run([exe, str(seed), str(frames)], capture_output=True, text=True, timeout=timeout, env=env)


def test_every_host_thread_under_thread_sanitizer(tmp_path):
    exe = build_driver(tmp_path)
    for seed in (1, 2, 3):
        p = run_driver(exe, seed, 300, 600)  # (~10 s here; a hang runs into the timeout)
        out = p.stdout + p.stderr
        assert p.returncode == 0, out[-3000:]
        assert "WARNING: ThreadSanitizer" not in out, out[-6000:]
        assert "drive ok:" in p.stdout, out[-2000:]
        calls = int(p.stdout.split("drive ok:")[1].split()[0])
        assert calls >= 300
    # launches that fail in the middle of a call, on the calling thread or on the launch thread (whose first error
    # is sticky until esvio_fe_reset and travels to the caller's error text): every failed call is followed by a
    # reset in the driver, the stream goes on, nothing hangs, nothing races
    for every in ("97", "701", "1933"):
        p = run_driver(exe, 5, 300, 600, HIPSTUB_FAIL_EVERY=every)
        out = p.stdout + p.stderr
        assert p.returncode == 0 and "drive ok:" in p.stdout, out[-3000:]
        assert "WARNING: ThreadSanitizer" not in out, out[-6000:]
        # (how many TRACK calls an injected failure lands in depends on the threads' timing — the helpers' and the
        # launch thread's own HIP calls count too —: only the densest setting is held to a minimum)
        failed = int(p.stdout.split("handles,")[1].split()[0])
        assert failed >= (3 if every == "97" else 0), p.stdout


def test_host_side_under_address_and_ub_sanitizers(tmp_path):
    """the same driver — both LK modes, host and device batches, the launch thread switched, resets and failing exits —
    built with -fsanitize=address,undefined: no report of either (round 5: one, a memcpy from an empty vector's null
    data() in fill_tracks), no leak at exit"""
    exe = build_driver(tmp_path, name="drive_asan", sanitize="address,undefined")
    for seed in (1, 2):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1")
        p = subprocess.run([exe, str(seed), "300"], capture_output=True, text=True, timeout=600, env=env)
        out = p.stdout + p.stderr
        assert p.returncode == 0 and "drive ok:" in p.stdout, out[-3000:]
        assert "runtime error" not in out and "AddressSanitizer" not in out and "LeakSanitizer" not in out, out[-6000:]
    # ... and on the failing exits: every 97th / 701st HIP call fails (calling thread or launch thread), the driver
    # resets and goes on — error paths are where buffers are forgotten or freed twice
    for every in ("97", "701"):
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", HIPSTUB_FAIL_EVERY=every)
        p = subprocess.run([exe, "5", "300"], capture_output=True, text=True, timeout=600, env=env)
        out = p.stdout + p.stderr
        assert p.returncode == 0 and "drive ok:" in p.stdout, out[-3000:]
        assert "runtime error" not in out and "AddressSanitizer" not in out and "LeakSanitizer" not in out, out[-6000:]
