#!/usr/bin/env python3
"""Benchmark of the MI355X event front-end hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one stereo event batch: SAE update (both cameras) ->
time surfaces -> pyramids -> temporal LK fwd/back -> [published frames: F-RANSAC, mask, Arc*
detection + greedy selection] -> stereo LK fwd/back -> undistort/velocity, exactly the sequence
of FeatureTracker::trackEvent (reference feature_tracker.cpp:340-603).  Workload at N=1:
BASELINE config C3 (superset of C2): stereo 640x480 synthetic scene stream, 5 Mev/s per camera
in 30 Hz batches (~167 k events per camera per step), shipped DSEC parameters
(max_cnt 300, min_dist 10, flow_back 1, equalize 0), frames published at freq=15 Hz like
stereo_event_tracker_node.cpp:177-188.  Events are resident in HBM before the timed region.

N>1: one process per GPU (torch.distributed / RCCL); every rank runs an independent stereo rig
(weak scaling, no data-path collective); the tracked-corner records of every published frame are
merged with one asynchronous all_gather (north-star hand-off to the estimator).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


# The per-kernel timers' slots are named after the kernel FUNCTION each one brackets (esvio_fe_kernel_name =
# the name rocprofv3 prints), so `kernels` / `roofline.kernel` match profiles/*.md verbatim.  The few slots that
# bracket more than one function:
KERNEL_FUNCS = {
    "k_tile_hist": ["k_tile_hist", "k_mc_warp"], "k_clahe": ["k_clahe_lut", "k_clahe_interp", "k_normalize"],
    "k_sae_apply": ["k_sae_apply", "k_sae_apply_ev", "k_sae_apply_ev_write"], "k_arc_map": ["k_arc_map", "k_arc_mark"],
}
# kernels whose reads are per-lane gathers, not wide coalesced streams: the guide calibrates the 2x
# FETCH_SIZE correction for coalesced reads only
GATHER_KERNELS = {"k_tile_apply", "k_sae_apply", "k_lk", "k_lk_f32", "k_arc_ev", "k_arc_map", "k_select", "k_select_mw",
                  "k_select_gbm", "k_compact", "k_dedup"}


def workload_label(W, H, rate, world, split):
    """BASELINE.json's config the run's shape corresponds to (C1..C5), derived from the shape — not assumed"""
    if (W, H) == (640, 480):
        if split == "camera" and world == 2:
            return "C4 (left/right cameras on 2 GPUs)"
        return "C3 (superset of C2)" if abs(rate - 5e6) < 1 else "C3's sensor at %.1f Mev/s per camera" % (rate / 1e6)
    if (W, H) == (346, 260):
        return "C1's sensor (DAVIS346) on the GPU path"
    if (W, H) == (1280, 720):
        return ("C5 (time-sliced over %d GPUs)" % world) if split == "time" and world > 1 else \
               "C5's sensor shape on %s" % ("one GPU" if world == 1 else "%d GPUs, one rig each" % world)
    return "custom shape"


def pmc_traffic(kernel):
    """HBM bytes per launch of the kernels behind stats label `kernel` from the newest committed
    rocprofv3 PMC summary of this same command (profiles/*_bench_c3.json, made by
    tools/rocprof_summary.py from separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes).  Units
    are KB; per MI355X_MICROARCH.md §HBM the gfx950 FETCH_SIZE counter reports half of a coalesced
    read stream, hence 2*FETCH + WRITE.  Returns (bytes, source file, note)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_c3.json")))
    if not files:
        return None, None, None
    try:
        d = json.load(open(files[-1]))
        funcs = KERNEL_FUNCS.get(kernel, [kernel])

        def total(counter):
            tot, disp = 0.0, 0
            for name, v in d["pmc"].get(counter, {}).items():
                base = name.split("(")[0].split("<")[0].split("::")[-1].replace("void ", "").strip()
                if base in funcs:
                    tot += v["avg"] * v["dispatches"]
                    disp = max(disp, v["dispatches"])
            return tot, disp
        f, nf = total("FETCH_SIZE")
        w, nw = total("WRITE_SIZE")
        if not nf or not nw:
            return None, None, None
        note = ("2x FETCH_SIZE correction applied; it is calibrated for wide coalesced reads only — "
                "uncalibrated for this gather kernel (raw FETCH+WRITE = %d B)" % int((f / nf + w / nw) * 1024)
                if kernel in GATHER_KERNELS else "2x FETCH_SIZE correction (coalesced stream)")
        return int((2.0 * f / nf + w / nw) * 1024), os.path.basename(files[-1]), note
    except Exception:
        return None, None, None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--rate", type=float, default=5e6, help="events/s per camera")
    ap.add_argument("--batch-hz", type=float, default=30.0)
    ap.add_argument("--freq", type=int, default=15, help="publish rate (config freq)")
    ap.add_argument("--cpu-frames", type=int, default=40, help="oracle frames for cpu_baseline (0=skip)")
    ap.add_argument("--cpu-procs", type=int, default=-1,
                    help="processes for the all-cores CPU figure (-1 = os.cpu_count(), 0 = skip)")
    ap.add_argument("--no-profile-pass", action="store_true")
    ap.add_argument("--call-phases", action="store_true",
                    help="diagnosis: every call of the last timed pass with its begin time, wall time and the library's "
                         "phases of it (esvio_fe_latency_recent, read after the passes) to stderr")
    ap.add_argument("--no-sae-pass", action="store_true",
                    help="skip the event-proportional chain at C5's batch size (`sae_chain_c5_batch` in the line)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--stream", choices=["scene", "poisson"], default="scene",
                    help="scene (default): edges of a static scene seen from a moving stereo rig plus 7 %% Poisson "
                         "noise — corners that exist and can be tracked, so LK, RANSAC and the selection do "
                         "their real work; poisson: homogeneous Poisson events only (BASELINE's wording), "
                         "where every corner is noise and the tracker mostly re-detects")
    ap.add_argument("--lk-accum", type=int, default=2, choices=[1, 2],
                    help="2 (default, the headline): calcOpticalFlowPyrLK's float sums in the order of the reference's "
                         "x86 OpenCV build (k_lk_f32) — the arithmetic of feature_tracker.cpp:410,417,490,495; "
                         "1: exact integer sums (k_lk), reported as the side figure `exact_sum_lk`")
    ap.add_argument("--equalize", type=int, default=0, choices=[0, 1],
                    help="1: CLAHE + normalize of the time surface before LK (config/esio_DSEC ships "
                         "equalize: 1; the headline stays at 0, the other shipped configs' setting)")
    ap.add_argument("--mc", type=int, default=0, choices=[0, 1],
                    help="1: the motion-compensated overload (Do_motion_correction: 1 in 3 of the 9 shipped configs): "
                         "every batch carries a Motion_correction_value with |accel| > 5 m/s^2, so all events before "
                         "the header stamp are warped (per-event Matrix3f::exp) before the SAE update")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="do not announce the next batch (esvio_fe_set_next_batch): strictly one "
                         "batch in flight, like the reference's depth-1 queues")
    ap.add_argument("--no-chain", action="store_true",
                    help="no launch waits on the device for another one (ESVIO_FE_NO_CHAIN=1): with --no-pipeline "
                         "the schedule whose per-kernel durations are clean (profiles/*_bench_c3)")
    ap.add_argument("--no-lazy", action="store_true",
                    help="wait for the stereo LK of newly detected corners inside the call that detects "
                         "them (default in replay mode: their right-camera entries are completed by "
                         "the next call; the published PointCloud rows never contain them)")
    ap.add_argument("--ahead", type=int, default=3, choices=[1, 2, 3],
                    help="batches announced ahead in replay mode (esvio_fe_set_next_batch)")
    ap.add_argument("--host-threads", type=int, default=0,
                    help="host threads of rejectWithF_event's RANSAC (esvio_fe_set_host_threads; "
                         "the result does not depend on it); 1 = the calling thread only; "
                         "0 = min(8, usable CPUs / (2 * ranks)): the helpers spin, so all ranks' "
                         "threads together have to stay inside the CPU quota")
    ap.add_argument("--launch-thread", type=int, default=-1, choices=[-1, 0, 1],
                    help="replay mode: a thread of the handle issues the announced batches' prefetch launches "
                         "(esvio_fe_set_launch_thread); -1 = on when this rank has at least 4 CPUs")
    ap.add_argument("--split", choices=["rigs", "camera", "time"], default="rigs",
                    help="N>1 sharding: one independent stereo rig per GPU (weak scaling, default); "
                         "BASELINE C4: left/right cameras of ONE rig on 2 GPUs; BASELINE C5: ONE stream, "
                         "every batch time-sliced over the N GPUs for the SAE update")
    ap.add_argument("--repeats", type=int, default=0,
                    help="times the timed --steps region is run over the continued stream (the first "
                         "is the headline ms_per_step; all are reported); 0 = 5 up to 30 steps, else 3")
    ap.add_argument("--max-cnt", type=int, default=300)
    ap.add_argument("--torch-exchange", action="store_true",
                    help="merge the tracked corners through torch.distributed (TrackExchange) also over "
                         "RCCL, instead of the library's own communicator (A/B)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with one rank: still create the process group and run the track exchange "
                         "(all_gather over RCCL with world size 1) — exercises the N>1 code path on a 1-GPU box")
    ap.add_argument("--no-reserve", action="store_true",
                    help="do not size the handle's buffers up front (esvio_fe_reserve): they then grow inside the "
                         "calls that first need them (A/B of the first-use allocation stalls)")
    ap.add_argument("--no-host-pass", action="store_true",
                    help="skip the extra pass with the events in host memory (host_resident_events)")
    ap.add_argument("--dist-backend", default="nccl",
                    help="torch.distributed backend (nccl = RCCL; gloo only for dry runs of the N>1 path)")
    return ap.parse_args()


def usable_cpus():
    """host threads this process may actually use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(q / p + 0.5)))
        except Exception:
            pass
    return n


def cgroup_cpu_stat():
    """(nr_throttled, throttled_usec) of this process's cgroup (CFS bandwidth control), or None"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            kv = dict(l.split()[:2] for l in open(path).read().splitlines() if l.strip())
            if "throttled_usec" in kv:
                return int(kv.get("nr_throttled", 0)), int(kv["throttled_usec"])
            if "throttled_time" in kv:  # (cgroup v1: nanoseconds)
                return int(kv.get("nr_throttled", 0)), int(kv["throttled_time"]) // 1000
        except Exception:
            pass
    return None


def bind_rank_to_cores(local_rank, n_local):
    """N ranks on one node: give each rank its own contiguous block of physical cores (with all their
    hardware threads), so that one rank's spinning RANSAC helpers — which stay on the calling
    thread's L3 domain — never share a core with another rank's threads.  Returns the number of
    cores of the block, or None when the topology cannot be read."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
        cores = {}
        for c in allowed:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as f:
                cores.setdefault(f.read().strip(), []).append(c)
        keys = sorted(cores, key=lambda k: min(cores[k]))
        per = len(keys) // n_local
        if per < 1:
            return None
        mine = keys[local_rank * per:(local_rank + 1) * per]
        os.sched_setaffinity(0, {c for k in mine for c in cores[k]})
        return len(mine)
    except Exception:
        return None


def cpu_all_cores(batches, args, W, H):
    """besides the like-for-like single-core figure, one independent oracle tracker per
    host core over the same batches, all started together; value = all events / slowest process."""
    import shutil
    import subprocess
    import tempfile
    nproc = usable_cpus() if args.cpu_procs < 0 else args.cpu_procs
    base = "/dev/shm" if os.path.isdir("/dev/shm") else None
    d = tempfile.mkdtemp(prefix="esvio_cpu_", dir=base)
    try:
        arrs = {}
        for i, (L, R) in enumerate(batches):
            arrs["L%d" % i] = L.view(np.uint8).reshape(-1, 16)
            arrs["R%d" % i] = R.view(np.uint8).reshape(-1, 16)
        path = os.path.join(d, "batches.npz")
        np.savez(path, **arrs)
        go = os.path.join(d, "go")
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "cpu_worker.py")
        procs = []
        for k in range(nproc):
            procs.append(subprocess.Popen(
                [sys.executable, worker, path, str(len(batches)), str(args.freq), str(W), str(H),
                 os.path.join(d, "ready%d" % k), go, str(args.lk_accum)], stdout=subprocess.PIPE, text=True))
        t_wait = time.time()
        while sum(os.path.exists(os.path.join(d, "ready%d" % k)) for k in range(nproc)) < nproc:
            if time.time() - t_wait > 120 or any(p.poll() not in (None, 0) for p in procs):
                for p in procs:
                    p.kill()
                return dict(error="workers did not start")
            time.sleep(0.01)
        open(go, "w").close()
        ev, worst = 0, 0.0
        for p in procs:
            out, _ = p.communicate(timeout=600)
            e, t = out.split()
            ev += int(e)
            worst = max(worst, float(t))
        return dict(value=round(ev / worst / 1e6, 3), unit="Mevents/s", cores=nproc,
                    sample="%d processes x %d stereo batches each (the same batches), one oracle tracker "
                           "per process, started together; all events / slowest process" % (nproc, len(batches)))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def self_launch(args):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks ourselves — the same
    command line the driver uses for N > 1 (one process per GPU under torch.distributed.run, RCCL
    rendezvous on 127.0.0.1) — and let rank 0's JSON line through.  Does not return."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this stack)
    env.setdefault("OMP_NUM_THREADS", "1")
    os.execvpe(cmd[0], cmd, env)


def main():
    args = parse()
    if args.no_chain:
        os.environ["ESVIO_FE_NO_CHAIN"] = "1"
        os.environ["ESVIO_FE_NO_CAMSPLIT"] = "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # stdout carries exactly ONE line, the JSON: whatever libraries print through C stdio on fd 1
    # (RCCL's version banner, for one) goes to stderr instead
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    total_cpus = usable_cpus()  # (before this rank is bound to its share of the cores)
    if world > 1:
        bind_rank_to_cores(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the front-end has no CPU fallback")
    n_dev = torch.cuda.device_count()
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    if world > n_dev and args.dist_backend == "nccl":
        # more ranks than devices (a dry run of the N > 1 path on a smaller box): RCCL refuses two
        # ranks on one device, so the collectives go through gloo and the line says so
        if rank == 0:
            print("bench: %d ranks on %d device(s): dry run over gloo, not a scaling measurement" % (world, n_dev),
                  file=sys.stderr)
        args.dist_backend = "gloo"
    dist = None
    multi = world > 1 or args.force_dist  # (the collective path runs, possibly with a single rank)
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)
    xdev = "cuda" if args.dist_backend == "nccl" else "cpu"

    from esvio_amd import frontend as FE
    from esvio_amd.events import event_times
    from esvio_amd.dist import CameraSplitRig, TimeSlicedSae, TrackExchange
    from esvio_amd.node import FreqControl
    from esvio_amd.synth import PoissonStream, SceneStream

    W, H = args.width, args.height
    repeats = args.repeats if args.repeats > 0 else (5 if args.steps <= 30 else 3)
    n_frames = args.warmup + args.steps * repeats
    n_prof = 0 if args.no_profile_pass else min(args.steps, 30)
    # ---- synthetic stream (per rank: an independent rig, different seed), resident in HBM
    cam_split = args.split == "camera" and world > 1
    time_split = args.split == "time" and multi
    if cam_split and world != 2:
        raise SystemExit("--split camera needs exactly 2 ranks")
    one_rig = cam_split or time_split
    stream_seed = args.seed + (0 if one_rig else 1000 * rank)
    scene = (PoissonStream(W, H, rate=args.rate, batch_hz=args.batch_hz, seed=stream_seed) if args.stream == "poisson"
             else SceneStream(W, H, rate=args.rate, batch_hz=args.batch_hz, seed=stream_seed))
    host_batches, dev_batches = [], []
    for _ in range(n_frames):
        L, R, _ = scene.next_batch()
        host_batches.append((L, R, len(L), len(R), event_times(L)[-1]))
        tl = torch.from_numpy(L.view(np.uint8).reshape(-1)).cuda()
        tr = torch.from_numpy(R.view(np.uint8).reshape(-1)).cuda()
        dev_batches.append((tl, tr, len(L), len(R), event_times(L)[-1]))
    torch.cuda.synchronize()

    cfg = FE.make_config(W, H, device=dev_index, max_cnt=args.max_cnt, min_dist=10, flow_back=1, f_ransac=1,
                         equalize=args.equalize, lk_accum=args.lk_accum)
    # PUB_THIS_FRAME depends on the batch timestamps only (node:155-188), so the whole plan is known
    # up front; replay mode hands it to esvio_fe_set_next_batch as the PUB hint
    fc = FreqControl(args.freq)
    pub_flags = []
    for b in dev_batches:
        pub_flags.append(fc.pub_this_frame(b[4]))
        if pub_flags[-1]:
            fc.published()
    helper_spin_us = None
    if args.host_threads <= 0:
        share = total_cpus / max(world, 1)  # CPUs this rank can count on
        if share >= 4:  # room for spinning helpers: they never sleep while frames keep coming
            args.host_threads = max(1, min(8, int(share) // 2))
        else:
            # a small share (8 ranks on a 16-CPU quota: 2 each): never more threads than CPUs — a helper
            # that loses its CPU stalls the job for a scheduler quantum (measured on 2 CPUs: 4 threads
            # 0.3-4.5 ms per step, 2 threads 0.133 -> 0.123, 1 thread 0.165 -> 0.15-0.157) — and helpers that block between jobs
            # (30 us of idle spin; the wake-up at the start of a published frame's call has them back
            # before the RANSAC begins), so that they do not burn the quota while idle
            args.host_threads = max(1, int(share))
            helper_spin_us = 30
    if helper_spin_us is not None:
        os.environ["ESVIO_FE_HELPER_SPIN_US"] = str(helper_spin_us)
    if args.launch_thread < 0:  # (one more spinning thread: only where the rank has CPUs to spare)
        args.launch_thread = 1 if total_cpus / max(world, 1) >= 4 else 0

    def motion_of(mod, i):
        """the Motion_correction_value of batch i (--mc 1): header stamp = the batch's last event, a
        slowly varying angular rate and velocity, |accel| = 7.1 m/s^2 (> 5: the warp is applied)"""
        if not args.mc:
            return None
        cam = cfg.cam[0]
        w = 0.1 * np.sin(0.37 * i)
        return mod.make_motion(host_batches[i][4], v=(0.08 + w, -0.03, 0.02), v_pre=(0.07 + w, -0.025, 0.015),
                               accel=(4.0, 5.0, 3.0), omega=(0.2 + w, -0.3, 0.4 - w),
                               fx=cam.fx, fy=cam.fy, cx=cam.cx, cy=cam.cy)
    motions = [motion_of(FE, i) for i in range(n_frames)]

    class Runner:
        """one tracker + the schedule it is driven with"""

        def __init__(self, pipeline, lazy, batches, exch=None, rig=None, tsl=None, comm=None, config=None,
                     host_threads=None, launch_thread=None):
            host_threads = args.host_threads if host_threads is None else host_threads
            launch_thread = args.launch_thread if launch_thread is None else launch_thread
            self.ft = FE.FeatureTracker(config if config is not None else cfg)
            self.comm = comm is not None
            if comm is not None:
                self.ft.comm_init(comm, rank, world)
                self.ft.set_auto_exchange(True)  # every published frame's records, enqueued under the next call's wait
            self.pipeline, self.lazy, self.batches, self.exch, self.rig, self.tsl = pipeline, lazy, batches, exch, rig, tsl
            self._args = [None] * len(batches)
            for k in range(len(batches)):
                self.arg(k)
            self.announced = 0
            if lazy:
                self.ft.set_lazy_new_stereo(True)
            if host_threads > 1:
                self.ft.set_host_threads(host_threads)
            if pipeline and launch_thread:
                self.ft.set_launch_thread(True)
            # set-up, not a step: every event-proportional buffer sized for the stream's largest batch, so
            # that no timed call allocates (`tail_latency.allocs` in the line counts the ones that do)
            if not args.no_reserve:
                self.ft.reserve(max(b[2] for b in batches), max(b[3] for b in batches),
                                host_batches=isinstance(batches[0][0], np.ndarray))

        def arg(self, k):
            a = self._args[k]
            if a is None:  # (built once per batch: the timed loop does no tensor / tuple work of its own)
                b = self.batches[k]
                if isinstance(b[0], np.ndarray):  # host-resident events
                    a = b
                else:
                    a = ((b[0].data_ptr(), b[2]), (b[1].data_ptr(), b[3]), b[2], b[3], b[4])
                self._args[k] = a
            return a

        def step(self, i, exchange=True):
            L, R, nl, nr, t_last = self.arg(i)
            pub = pub_flags[i]
            ft = self.ft
            if self.rig is not None:  # C4: rank 0 = left camera + tracking, rank 1 = right camera
                self.rig.track(t_last, L, R, pub)
                if pub and exchange:
                    self.exch.submit_tracker(ft, async_op=True)
                return nl if rank == 0 else nr
            if self.tsl is not None:  # C5: every rank applies its time slice of the batch to the SAE
                self.tsl.track(t_last, L, R, pub)
                return (nl + nr) if rank == 0 else 0
            if self.pipeline:  # replay mode: the next batches are already in HBM; announce them ahead
                # (host-resident batches one further: a batch is staged through pinned chunks under the
                # call after its announcement and taken up by the prefetch stream in the call after that)
                ahead = args.ahead + (1 if isinstance(self.batches[0][0], np.ndarray) else 0)
                while self.announced < min(i + ahead, len(self.batches) - 1):
                    k = self.announced = self.announced + 1
                    L2, R2, _, _, t2 = self.arg(k)
                    ft.set_next_batch(t2, L2, R2, pub_flags[k], measurements=motions[k])
            ft.trackEvent(t_last, L, R, pub, copy=False, measurements=motions[i])
            if pub and exchange:  # merge all rigs' tracked corners (asynchronous)
                if self.comm:
                    pass  # (esvio_fe_set_auto_exchange: the library does it)
                elif self.exch is not None:
                    self.exch.submit_tracker(ft, async_op=True)
            return nl + nr

    # The merge of every rank's tracked corners (north-star hand-off).  Over RCCL it runs inside the
    # library on the handle's own communicator (esvio_fe_comm_init / esvio_fe_exchange_begin: pack,
    # upload, ncclAllGather, download on a side stream; ~20 us of host time per published frame); the
    # torch.distributed mirror (TrackExchange, ~100 us of Python / c10d per published frame) serves
    # the gloo dry runs.
    lib_exchange = multi and not time_split and not cam_split and args.dist_backend == "nccl" and not args.torch_exchange
    exch = (TrackExchange(cfg.max_cnt, world, device=xdev, dist=dist,
                          stream=torch.cuda.Stream() if xdev == "cuda" else None)
            if multi and not time_split and not lib_exchange else None)
    comm_id = None
    if lib_exchange:
        box = [None]
        if rank == 0:
            try:
                box[0] = FE.comm_unique_id()
            except FE.FrontendError as e:  # (no librccl to dlopen: every rank takes the torch.distributed path)
                print("bench: %s; exchanging through torch.distributed instead" % e, file=sys.stderr)
        dist.broadcast_object_list(box, src=0)
        comm_id = box[0]
    pipeline = not args.no_pipeline and not one_rig
    lazy = pipeline and not args.no_lazy
    try:
        main_run = Runner(pipeline, lazy, dev_batches, exch=exch, comm=comm_id)
        comm_ok = 1.0
    except FE.FrontendError as e:
        if comm_id is None:
            raise
        print("bench: rank %d: esvio_fe_comm_init failed (%s)" % (rank, e), file=sys.stderr)
        main_run = Runner(pipeline, lazy, dev_batches, exch=exch, comm=None)
        comm_ok = 0.0
    if lib_exchange:
        # the library communicator is used only if every rank has it; otherwise all ranks exchange
        # through torch.distributed (RCCL as well), so that the collectives still match up
        flag = torch.tensor([comm_ok if comm_id is not None else 0.0], device=xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 1.0:
            comm_id = None
            main_run.comm = False
            main_run.ft.set_auto_exchange(False)
            main_run.exch = exch = TrackExchange(cfg.max_cnt, world, device=xdev, dist=dist,
                                                 stream=torch.cuda.Stream() if xdev == "cuda" else None)
    ft = main_run.ft
    if cam_split:
        main_run.rig = CameraSplitRig(ft, rank, dist, device=xdev)
    if time_split:
        main_run.tsl = TimeSlicedSae(ft, rank, world, dist, device=xdev)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # The interpreter's cyclic garbage collector is kept out of the warm-up and the timed passes (what
    # timeit does): the set-up above leaves ~10^5 tracked objects (batches, arrays, ctypes wrappers) and a
    # full collection walking them takes milliseconds.  The step itself allocates no cycles, and the
    # process ends soon after.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    for i in range(args.warmup):
        main_run.step(i)
    # ---- the timed region: EXACTLY --steps steps (pass 0 is the headline); the same region is then
    # repeated over the continued stream (passes 1..R-1) so that the spread can be reported.  Every
    # step's wall time is kept (one perf_counter per step), and the library's own per-call record
    # (esvio_fe_latency_stats) names where the slowest call of each pass spent its time.
    passes = []
    ransac_passes, ransac_tails = [], []
    step_ms, lib_lat = [], []
    throttle0 = cgroup_cpu_stat()
    for r in range(repeats):
        FE.ransac_stats(reset=True)
        ft.latency_stats(reset=True)
        barrier()
        t0 = time.perf_counter()
        n_events = 0
        lo = args.warmup + r * args.steps
        marks = [t0]
        for i in range(lo, lo + args.steps):
            n_events += main_run.step(i)
            marks.append(time.perf_counter())
        if lazy:
            ft.finish(copy=False)  # the last published frame's deferred right-camera entries
        if exch is not None:
            exch.wait()
        if comm_id is not None and any(pub_flags[lo:lo + args.steps]):
            ft.exchange_end(want=False)
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        marks.append(t0 + elapsed)  # (last entry: finish() + the closing synchronize)
        step_ms.append([(b - a) * 1e3 for a, b in zip(marks, marks[1:])])
        tot = torch.tensor([float(n_events), elapsed], dtype=torch.float64, device=xdev)
        if multi:
            dist.all_reduce(tot[0:1], op=dist.ReduceOp.SUM)
            dist.all_reduce(tot[1:2], op=dist.ReduceOp.MAX)
        passes.append((float(tot[0].item()), float(tot[1].item()), n_events))
        ransac_tails.append(FE.ransac_tail())
        ransac_passes.append(FE.ransac_stats())
        lib_lat.append(ft.latency_stats())
    throttle1 = cgroup_cpu_stat()
    if args.call_phases and rank == 0:  # (read AFTER the timed passes: looking changes nothing)
        prev_end = None
        for call, pub, t_begin, ms, ph in ft.latency_recent(args.steps):
            gap = 0.0 if prev_end is None else t_begin - prev_end
            prev_end = t_begin + ms
            print("call %3d pub %d begins %9.1f us (%5.1f after the previous one's end) takes %6.1f us: %s" % (
                call, int(pub), t_begin * 1e3, gap * 1e3, ms * 1e3,
                ", ".join("%s %.0f" % (k, v * 1e3) for k, v in ph.items())), file=sys.stderr)
        print("counters:", ft.debug_counters(), file=sys.stderr)
    try:
        os_threads = len(os.listdir("/proc/self/task"))
    except Exception:
        os_threads = None
    total_events, max_elapsed, n_events = passes[0]  # (the collector stays off for the extra passes below too)
    if os.environ.get("ESVIO_BENCH_STEP_TIMES"):  # (diagnostic: every pass's per-step wall times to stderr)
        for r, sm in enumerate(step_ms):
            print("bench: pass %d per-step ms:" % r, " ".join("%.3f" % v for v in sm), file=sys.stderr)
            print("bench: pass %d library latency: %s; ransac tail %s" % (r, lib_lat[r], ransac_tails[r]), file=sys.stderr)
    n_tracks = (len(ft.ids), len(ft.ids_right))
    # The passes below are other trackers, one after the other, each alone in the process like this one was: the
    # runtime has four hardware queues per priority level and PROCESS, and an idle tracker's streams keep theirs
    # (measured: the host-resident replay pass beside the idle main tracker 0.23 ms/step, alone 0.16)
    ft.close()
    # rejectWithF_event's host RANSAC inside each timed pass (this rank)
    host_ransac = [None if not rs["calls"] else dict(
        calls=rs["calls"], mean_points=round(rs["points"] / rs["calls"], 1),
        mean_iterations=round(rs["iterations"] / rs["calls"], 1), mean_us=round(rs["us"] / rs["calls"], 2),
        lmeds_calls=rs["lmeds_calls"], lmeds_mean_us=round(rs["lmeds_us"] / max(rs["lmeds_calls"], 1), 2),
        us_per_step=round((rs["us"] + rs["lmeds_us"]) / args.steps, 2),
        max_us=round(max(tl["max_us"], tl["lmeds_max_us"]), 1), redone_iterations=tl["redone_iterations"])
        for rs, tl in zip(ransac_passes, ransac_tails)]

    # ---- per-kernel pass (rank 0): HIP events around every launch on the stream it is launched on
    # (kept out of the timed region because the records cost host time).  Two schedules over the
    # first frames of the same stream, each on its own tracker:
    #   "kernels"            strictly one batch in flight, nothing speculative or lazy: clean per-launch
    #                        durations (no k_lk launch waits for k_select or for another k_lk)
    #   "kernels_pipelined"  the replay schedule of the timed region: durations of the speculative /
    #                        chained k_lk launches include the time their waves wait for their inputs
    roof = None
    device_activity = None
    kernels, kernels_pipe = {}, {}
    prof_ms = {}

    def kernel_pass(pipe):
        if not pipe:  # clean durations: no launch of this pass waits on the device for another one,
            os.environ["ESVIO_FE_NO_CHAIN"] = "1"     # ... and none runs beside the other camera's chain
            os.environ["ESVIO_FE_NO_CAMSPLIT"] = "1"
        try:
            run = Runner(pipe, pipe and lazy, dev_batches)
        finally:
            os.environ.pop("ESVIO_FE_NO_CHAIN", None)
            os.environ.pop("ESVIO_FE_NO_CAMSPLIT", None)
        for i in range(min(args.warmup, 6)):
            run.step(i, exchange=False)
        torch.cuda.synchronize()
        run.ft.set_profiling(True)
        run.ft.reset_kernel_stats()
        tp0 = time.perf_counter()
        lo = min(args.warmup, 6)
        for i in range(lo, lo + n_prof):
            run.step(i, exchange=False)
        if pipe and lazy:
            run.ft.finish(copy=False)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - tp0) / n_prof * 1e3
        stats = run.ft.kernel_stats()
        run.ft.close()
        out = {}
        for k, st in stats.items():
            if st["launches"]:
                avg_us = st["ms"] / st["launches"] * 1e3
                gbs = (st["alg_bytes"] / st["launches"]) / (avg_us * 1e-6) / 1e9 if avg_us > 0 else 0.0
                out[k] = dict(total_ms=round(st["ms"], 4), launches=st["launches"], avg_us=round(avg_us, 3),
                              alg_bytes_per_launch=st["alg_bytes"] // st["launches"],
                              achieved_GBs=round(gbs, 2), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 5))
        return out, ms

    if n_prof and rank == 0 and not one_rig:
        kernels, prof_ms["one_batch_in_flight"] = kernel_pass(False)
        if pipeline:
            kp, prof_ms["replay"] = kernel_pass(True)
            # what the device did per step of the replay schedule (HIP-event pairs around every launch of that pass):
            # something a reader can hold against ms_per_step when a sampled gpu_busy counter saw nothing of a
            # 2 ms timed region
            device_activity = dict(
                kernel_launches_per_step=round(sum(v["launches"] for v in kp.values()) / n_prof, 1),
                kernel_ms_per_step_summed=round(sum(v["total_ms"] for v in kp.values()) / n_prof, 4),
                profiled_ms_per_step=round(prof_ms["replay"], 4),
                note="replay schedule with per-launch event pairs on (slower than the timed region); the sum over "
                     "kernels exceeds the step time where streams overlap")
            kernels_pipe = {k: dict(avg_us=v["avg_us"], launches=v["launches"]) for k, v in kp.items()}
        for k in kernels:  # PMC-measured HBM bytes per launch beside the algorithmic ones
            tb, _, note = pmc_traffic(k)
            kernels[k]["hbm_bytes_per_launch_pmc"] = tb
            if note:
                kernels[k]["pmc_note"] = note
        dom = max(kernels, key=lambda k: kernels[k]["total_ms"]) if kernels else None
        if dom:
            d = kernels[dom]
            traffic, src, note = pmc_traffic(dom)
            roof = dict(bound="hbm", kernel=dom, achieved=d["achieved_GBs"], peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=round(d["achieved_GBs"] / HBM_PEAK_GBS, 6), traffic=traffic,
                        traffic_source=src, traffic_note=note, avg_launch_us=d["avg_us"],
                        avg_launch_us_replay_schedule=kernels_pipe.get(dom, {}).get("avg_us"),
                        alg_bytes_per_launch=d["alg_bytes_per_launch"],
                        profiled_ms_per_step=round(prof_ms["one_batch_in_flight"], 4),
                        schedule="one batch in flight, no speculative / chained / lazy launches "
                                 "(clean kernel durations); the timed region uses the replay schedule")

    def side_pass(pipe, lz, batches, config=None, host_threads=None, launch_thread=None):
        """one more tracker over the same frames as pass 0 (its own warm-up, then --steps timed steps);
        reported beside `value`, never as it"""
        run = Runner(pipe, lz, batches, config=config, host_threads=host_threads, launch_thread=launch_thread)
        for i in range(args.warmup):
            run.step(i, exchange=False)
        torch.cuda.synchronize()
        run.ft.latency_stats(reset=True)
        th0 = time.perf_counter()
        ev = 0
        for i in range(args.warmup, args.warmup + args.steps):
            ev += run.step(i, exchange=False)
        if lz:
            run.ft.finish(copy=False)
        torch.cuda.synchronize()
        th = time.perf_counter() - th0
        lt = run.ft.latency_stats()
        run.ft.close()
        out = dict(value=round(ev / th / 1e6, 3), unit="Mevents/s", ms_per_step=round(th / args.steps * 1e3, 4),
                   call_ms_max=round(lt["max_ms"], 4))
        if lt["max_ms"] > 1.0:  # (an outlier: say where it went)
            out["slowest_call"] = dict(phases_ms=lt["max_phase_ms"], invol_switches=lt["max_invol_switches"],
                                       allocs=lt["max_allocs"], index=lt["max_call"])
        return out, ev, th

    extra = rank == 0 and world == 1 and not one_rig and not args.no_host_pass
    # ---- the same replay schedule with the events in HOST memory (what the drop-in binding of
    # INTEGRATION.md passes): every batch then crosses PCIe inside the call that prefetches it
    host_res = None
    if extra:
        host_res, ev, th = side_pass(pipeline, lazy, host_batches)
        host_res.update(h2d_GBs=round(ev * 16 / th / 1e9, 2),
                        note="events handed over as pageable host buffers (ESVIO_FE_HOST); never `value`")

    # ---- the reference's own call pattern: depth-1 queues (stereo_event_tracker_node.cpp:128-142), one
    # esvio_fe_track_event per message, nothing announced, nothing lazy — INTEGRATION.md section 2's plain call
    one_batch = None
    if extra:
        ob_dev, _, _ = side_pass(False, False, dev_batches)
        ob_host, ev, th = side_pass(False, False, host_batches)
        # ... and from host arrays the caller has page-locked once where they lie (esvio_fe_register_host_buffer: the
        # node's deserialisation buffer / EventArray storage): no staging copy by the CPU, the arrays cross PCIe as they are
        regs = [(FE.RegisteredEvents(b[0]), FE.RegisteredEvents(b[1])) for b in host_batches]
        ob_reg, _, _ = side_pass(False, False, [(r[0].array, r[1].array) + tuple(b[2:]) for r, b in zip(regs, host_batches)])
        for r in regs:
            r[0].free()
            r[1].free()
        one_batch = dict(device_resident_ms_per_step=ob_dev["ms_per_step"], host_pageable_ms_per_step=ob_host["ms_per_step"],
                         host_registered_ms_per_step=ob_reg["ms_per_step"], host_registered_Mev_s=ob_reg["value"],
                         device_resident_Mev_s=ob_dev["value"], host_pageable_Mev_s=ob_host["value"],
                         device_resident_call_ms_max=ob_dev["call_ms_max"], host_pageable_call_ms_max=ob_host["call_ms_max"],
                         slowest_calls={k: v["slowest_call"] for k, v in (("device", ob_dev), ("host", ob_host))
                                        if "slowest_call" in v} or None,
                         note="one batch in flight, no announcement: the reference node's depth-1 pattern; never `value`")

    # ---- the same replay schedule in the OTHER LK mode.  The headline (lk_accum 2) accumulates the LK sums in
    # float in the order of the reference's x86 OpenCV build; lk_accum 1 sums exactly (integers): faster, but
    # not that build's arithmetic (`lk_modes`: within 1e-4 px on ~97 % of the points only).  Reported beside
    # `value`, never as it.
    other_lk = None
    if extra:
        other = 1 if args.lk_accum == 2 else 2
        fcfg = FE.make_config(W, H, device=dev_index, max_cnt=args.max_cnt, min_dist=10, flow_back=1, f_ransac=1,
                              equalize=args.equalize, lk_accum=other)
        other_lk, _, _ = side_pass(pipeline, lazy, dev_batches, config=fcfg)
        other_lk["lk_accum"] = other
        other_lk["note"] = ("lk_accum 1: exact integer LK sums (k_lk; bit-exact against the oracle's exact mode, NOT the "
                            "reference build's float arithmetic); never `value`" if other == 1 else
                            "lk_accum 2: LK sums in float in the order of the reference's x86 OpenCV build "
                            "(bit-exact against the oracle's float-order mode); never `value`")

    # ---- what a small host gets (a robot's companion computer is not a 256-CPU server): the same replay schedule
    # (a) with ONE host thread — no RANSAC helpers, no launch thread — on whatever CPUs this process has, and
    # (b) confined to TWO CPUs with two threads (the caller and one helper that blocks when idle).  Never `value`.
    low_cpu = None
    if extra:
        one, _, _ = side_pass(pipeline, lazy, dev_batches, host_threads=1, launch_thread=0)
        low_cpu = dict(one_host_thread_ms_per_step=one["ms_per_step"], one_host_thread_call_ms_max=one["call_ms_max"])
        try:
            cpus = sorted(os.sched_getaffinity(0))
            if len(cpus) >= 2:
                spin = os.environ.get("ESVIO_FE_HELPER_SPIN_US")
                os.environ["ESVIO_FE_HELPER_SPIN_US"] = "30"
                os.sched_setaffinity(0, set(cpus[:2]))
                try:
                    two, _, _ = side_pass(pipeline, lazy, dev_batches, host_threads=2, launch_thread=0)
                finally:
                    os.sched_setaffinity(0, set(cpus))
                    if spin is None:
                        os.environ.pop("ESVIO_FE_HELPER_SPIN_US", None)
                    else:
                        os.environ["ESVIO_FE_HELPER_SPIN_US"] = spin
                low_cpu.update(two_cpus_two_threads_ms_per_step=two["ms_per_step"], two_cpus_call_ms_max=two["call_ms_max"],
                               two_cpus=cpus[:2])
        except OSError as e:  # (no affinity control here)
            low_cpu["two_cpus_error"] = str(e)
        low_cpu["note"] = ("replay schedule, device-resident events, lk_accum %d; the headline uses %d host threads%s"
                           % (args.lk_accum, max(1, args.host_threads), " + the launch thread" if args.launch_thread else ""))

    # ---- the HBM-bound kernels at a batch size where they are HBM-bound: createSAE_left/right of one
    # stereo batch at C5's sensor shape and rate (1280x720, 100 Mev/s per camera: 6.7 M events), device
    # resident, nothing else running; per-launch HIP-event pairs on the launching stream (the library's
    # own, ~2 us of each figure is the pair).  SURVEY.md section 8d's accounting: ingest 16 B/event
    # (k_tile_hist), SAE update 32 B/event (k_tile_apply); the partition's own traffic is extra.
    sae_chain = None
    if rank == 0 and world == 1 and not args.no_sae_pass:
        sae_chain = {"note": "createSAE_left/right alone, 1280x720 stereo, 100 Mev/s per camera, one 1/30 s batch per launch "
                             "chain; HIP-event pairs around every launch; algorithmic bytes per SURVEY 8d "
                             "(k_tile_scatter: its own 16 B in + 8 B out per event, not in SURVEY's table)"}
        for sname, cls in (("scene", SceneStream), ("uniform", PoissonStream)):
            st5 = cls(1280, 720, rate=1e8, seed=12345)
            bat = []
            for _ in range(2):
                L5, R5, _t = st5.next_batch()
                bat.append((FE.EventBuffer(L5, FE.DEVICE), FE.EventBuffer(R5, FE.DEVICE), len(L5) + len(R5)))
            ft5 = FE.FeatureTracker(FE.make_config(1280, 720))
            for b5 in bat:  # (buffers grow)
                ft5.detector.createSAE_stereo(b5[0].arg, b5[1].arg)
            ft5.set_profiling(True)
            ft5.reset_kernel_stats()
            n5, ev5 = 16, 0
            for i in range(n5):
                b5 = bat[i % len(bat)]
                ft5.detector.createSAE_stereo(b5[0].arg, b5[1].arg)
                ev5 += b5[2]
            ks5 = ft5.kernel_stats()
            ft5.close()
            for b5 in bat:
                b5[0].free()
                b5[1].free()
            per = ev5 / n5
            row, tot = {}, 0.0
            # (the partition writes 8-byte records whenever the batch's stamps allow it — these synthetic
            # streams' always do — unless ESVIO_FE_WIDE_RECORDS forces the 16-byte form)
            scatter_bpe = 32 if os.environ.get("ESVIO_FE_WIDE_RECORDS") else 24
            for func, bpe in (("k_tile_hist", 16), ("k_tile_scan", 0), ("k_tile_scatter", scatter_bpe), ("k_tile_apply", 32)):
                v = ks5.get(func)
                if not v or not v["launches"]:
                    continue
                us = v["ms"] / v["launches"] * 1e3
                tot += us
                row[func] = dict(avg_launch_us=round(us, 2), alg_bytes_per_launch=int(bpe * per),
                                 achieved_GBs=round(bpe * per / us / 1e3, 1),
                                 frac=round(bpe * per / us / 1e3 / HBM_PEAK_GBS, 4))
            row["chain"] = dict(us=round(tot, 2), events_per_batch=int(per), alg_bytes=int(48 * per),
                                achieved_GBs=round(48 * per / tot / 1e3, 1), frac=round(48 * per / tot / 1e3 / HBM_PEAK_GBS, 4))
            sae_chain[sname] = row

    # ---- the whole step at C5's sensor shape and rate on ONE GPU (1280x720, 100 Mev/s per camera: 3.3 M events per
    # camera and batch), plain calls, device-resident: where the event-proportional kernels are HBM-bound.  Three
    # timed steps; then the same three with the per-launch event pairs on, for the kernels' live durations.
    c5_leg = None
    if rank == 0 and world == 1 and not args.no_sae_pass and (W, H) != (1280, 720):
        st5 = SceneStream(1280, 720, rate=1e8, batch_hz=args.batch_hz, seed=args.seed)
        nw5, ns5 = 2, 3
        bat = []
        for _ in range(nw5 + ns5):
            L5, R5, _t = st5.next_batch()
            bat.append((FE.EventBuffer(L5, FE.DEVICE), FE.EventBuffer(R5, FE.DEVICE), len(L5) + len(R5), event_times(L5)[-1]))
        cfg5 = FE.make_config(1280, 720, device=dev_index, max_cnt=args.max_cnt, min_dist=10, flow_back=1, f_ransac=1,
                              lk_accum=args.lk_accum)

        def c5_run(profile):
            if profile:  # (as in the C3 kernel pass: one launch per kernel and frame, nothing beside it)
                os.environ["ESVIO_FE_NO_CHAIN"] = "1"
                os.environ["ESVIO_FE_NO_CAMSPLIT"] = "1"
            try:
                ft5 = FE.FeatureTracker(cfg5)
            finally:
                os.environ.pop("ESVIO_FE_NO_CHAIN", None)
                os.environ.pop("ESVIO_FE_NO_CAMSPLIT", None)
            ft5.reserve(max(b[0].n for b in bat), max(b[1].n for b in bat), host_batches=False)
            if args.host_threads > 1:
                ft5.set_host_threads(args.host_threads)
            for k in range(nw5):
                ft5.trackEvent(bat[k][3], bat[k][0].arg, bat[k][1].arg, k % 2 == 0, copy=False)
            torch.cuda.synchronize()
            if profile:
                ft5.set_profiling(True)
                ft5.reset_kernel_stats()
            t5 = time.perf_counter()
            ev = 0
            for k in range(nw5, nw5 + ns5):
                ft5.trackEvent(bat[k][3], bat[k][0].arg, bat[k][1].arg, k % 2 == 0, copy=False)
                ev += bat[k][2]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t5
            ks = ft5.kernel_stats() if profile else None
            ft5.close()
            return ev, dt, ks
        ev5, dt5, _ = c5_run(False)
        _, _, ks5 = c5_run(True)
        for b5 in bat:
            b5[0].free()
            b5[1].free()
        per5 = ev5 / ns5
        # SURVEY 8d's per-event / per-pixel accounting of the step: ingest 16 + SAE update 32 B/event, render 17 B/px/camera
        alg5 = 48 * per5 + 17 * 1280 * 720 * 2
        kk = {}
        # (bytes per launch = the step's bytes over the launches the step made: one per kernel in the profiled run)
        for name, bps in (("k_tile_hist", 16 * per5), ("k_tile_scatter", 24 * per5), ("k_tile_apply", 32 * per5),
                          ("k_time_surface4", 17 * 1280 * 720 * 2), ("k_arc_map", None), ("k_arc_ev", None),
                          ("k_lk_f32" if args.lk_accum == 2 else "k_lk", None)):
            v = ks5.get(name)
            if v and v["launches"]:
                us = v["ms"] / v["launches"] * 1e3
                kk[name] = dict(avg_launch_us=round(us, 2), launches=v["launches"])
                if bps:
                    bpl = bps * ns5 / v["launches"]
                    kk[name].update(alg_bytes_per_launch=int(bpl), achieved_GBs=round(bpl / us / 1e3, 1),
                                    frac=round(bpl / us / 1e3 / HBM_PEAK_GBS, 4))
        c5_leg = dict(workload="%s: stereo 1280x720 scene stream, 100 Mev/s per camera, %g Hz batches, plain calls, device-resident, "
                               "lk_accum %d" % (workload_label(1280, 720, 1e8, 1, "rigs"), args.batch_hz, args.lk_accum),
                      steps=ns5, warmup=nw5, events_per_step=int(per5), ms_per_step=round(dt5 / ns5 * 1e3, 4),
                      value=round(ev5 / dt5 / 1e6, 1), unit="Mevents/s",
                      alg_bytes_per_step=int(alg5), alg_GBs=round(alg5 * ns5 / dt5 / 1e9, 1),
                      alg_frac_of_hbm_peak=round(alg5 * ns5 / dt5 / 1e9 / HBM_PEAK_GBS, 4), kernels=kk,
                      note="a side figure at BASELINE C5's shape on one GPU, never `value`; kernels: live HIP-event pairs of a second "
                           "run of the same three steps with one launch per kernel and frame and nothing beside it "
                           "(ESVIO_FE_NO_CHAIN / _NO_CAMSPLIT, like `kernels`); the timed steps run the plain call's real schedule")

    # ---- CPU baseline: the oracle (single-threaded port of the reference path) on a bounded sample
    cpu = None
    if rank == 0 and world == 1 and args.cpu_frames > 0:  # (the CPU baseline is an N = 1 figure)
        from oracle import oracle as O
        ocfg = O.make_config(W, H, max_cnt=args.max_cnt, min_dist=10, flow_back=1, f_ransac=1,
                             lk_accum=args.lk_accum, equalize=args.equalize)
        tr_o = O.Tracker(ocfg)
        fco = FreqControl(args.freq)
        nfr = min(args.cpu_frames, len(host_batches))
        ev = 0
        O.lk_pair_stats(True)
        tc0 = time.perf_counter()
        for i in range(nfr):
            L, R = host_batches[i][:2]
            t_last = host_batches[i][4]
            pub = fco.pub_this_frame(t_last)
            tr_o.track_event(t_last, L, R, pub, motion=motion_of(O, i))
            if pub:
                fco.published()
            ev += len(L) + len(R)
        tc = time.perf_counter() - tc0
        st = tr_o.stage_seconds()
        cpu = dict(value=round(ev / tc / 1e6, 3), unit="Mevents/s", cores=1, kind="port",
                   sample="%d stereo batches (%d events) of the same stream, oracle/liboracle.so, 1 thread; LK loop: %s"
                          % (nfr, ev, "lk_accum 2 = the x86 SIMD128 loop on SSE2 registers (b vector and A matrix sums; the "
                                      "window extraction is scalar)" if args.lk_accum == 2 else "lk_accum 1 = scalar loop, int64 sums"),
                   ms_per_step=round(tc / nfr * 1e3, 3),
                   stage_ms_per_step={k: round(v / nfr * 1e3, 3) for k, v in st.items()},
                   host_cpus=os.cpu_count(), usable_cpus=usable_cpus())
        if roof and roof["kernel"] in ("k_lk", "k_lk_f32"):
            # Issue model of the dominant kernel: one wave per point walks its <= 30 iterations per
            # level serially, the launch lasts as long as its slowest point.  Iteration counts come
            # from the CPU baseline run above (same frames, same arithmetic); instructions per
            # iteration from the gfx950 ISA of lk_point's loop; issue rate of a lone wave from
            # tools/clock_probe.hip.
            ps = O.lk_pair_stats(True)
            # round 5's loops (gfx950 ISA of lk_point's inner loop, every instruction counted — a lone wave issues one
            # instruction of ANY kind per ~4 cycles at best): exact sums 123, float order 178 (one walking round);
            # us per iteration (a few points: no straggler statistics in it) and the launch's fixed part — the launch
            # constant (7 / 9 us) + the six level visits of a forward + backward pair at 2.0 us each — from
            # tools/lk_const_probe.py on this GPU
            instr, us_iter, const_us = (123, 0.26, 19.0) if args.lk_accum == 1 else (178, 0.41, 21.0)
            ghz = 2.4
            roof["issue_model"] = dict(
                lk_accum=args.lk_accum,
                iterations_slowest_point_per_launch=round(ps["slowest_mean"], 1),
                iterations_mean_per_point=round(ps["mean_per_point"], 1),
                instructions_per_iteration=instr, cycles_per_instruction_lone_wave=round(us_iter * ghz * 1e3 / instr, 2),
                clock_GHz=ghz, us_per_iteration=us_iter, launch_constant_us=const_us,
                modeled_us=round(ps["slowest_mean"] * us_iter + const_us, 1), measured_us=roof["avg_launch_us"],
                note="latency-bound by construction: the HBM fraction of this kernel says nothing about "
                     "its quality; the streaming kernels' fractions are in `kernels`")
        if args.cpu_procs != 0:
            cpu["all_cores"] = cpu_all_cores([b[:2] for b in host_batches[:nfr]], args, W, H)

    # ---- which LK mode carries north_star's 1e-4 px: the headline's exact sums (lk_accum 1) are not the
    # arithmetic of the reference's build; measured here on this stream's own frames (checker only: the
    # oracle's exact mode — bit-identical to the GPU default by the -m gpu tests — against the oracle's
    # x86 float order at the tracker's points, temporal and stereo calls)
    lk_modes = None
    if rank == 0 and world == 1 and args.cpu_frames > 0 and not args.mc:
        from oracle import oracle as O
        tr_l = O.Tracker(O.make_config(W, H, max_cnt=args.max_cnt, min_dist=10, flow_back=1, f_ransac=1, lk_accum=1,
                                       equalize=args.equalize))
        d_all, flips, n_pts = [], 0, 0
        prev_img, prev_pts = None, None
        for i in range(min(8, len(host_batches))):
            L, R = host_batches[i][:2]
            r = tr_l.track_event(host_batches[i][4], L, R, True)
            img = tr_l.time_surface(0)
            calls = []
            if prev_img is not None and len(prev_pts):
                calls.append((prev_img, img, prev_pts))
            cur = np.array(r.cur_pts, np.float32)
            if len(cur):
                calls.append((img, tr_l.time_surface(1), cur))
            for a_img, b_img, pts in calls:
                e_pts, e_st = O.lk(a_img, b_img, pts, pts.copy(), max_level=3, flags=0, accum=1)
                f_pts, f_st = O.lk(a_img, b_img, pts, pts.copy(), max_level=3, flags=0, accum=2)
                both = (e_st == 1) & (f_st == 1)
                d_all.append(np.abs(e_pts[both] - f_pts[both]).max(axis=1))
                flips += int((e_st != f_st).sum())
                n_pts += int(len(pts))
            prev_img, prev_pts = img, cur
        d = np.concatenate(d_all) if d_all else np.zeros(1)
        lk_modes = {
            "headline": ("lk_accum 2: float sums in the recalled order of the reference's x86 OpenCV build — bit-identical "
                         "to the oracle's float-order mode, i.e. positions equal to that build's as far as the recalled "
                         "order is its order" if args.lk_accum == 2 else
                         "lk_accum 1: exact integer sums (bit-exact against the oracle's exact mode)"),
            "other_mode": ("lk_accum 1 (exact integer sums, `exact_sum_lk` is its rate) meets north_star's 1e-4 px "
                           "against the float order only on the share of points below" if args.lk_accum == 2 else
                           "lk_accum 2 (`float_order_lk` is its rate) is the mode that is bit-identical to the "
                           "reference build's float order"),
            "exact_sums_vs_float_order": dict(
                points=n_pts, status_flips=flips, p50=float(np.percentile(d, 50)), p90=float(np.percentile(d, 90)),
                p99=float(np.percentile(d, 99)), max=float(d.max()), within_1e4=round(float((d <= 1e-4).mean()), 4),
                sample="first %d frames of this stream, temporal + stereo calls at the tracked corners, oracle "
                       "accum 1 vs accum 2" % min(8, len(host_batches))),
        }

    if rank == 0:
        ms_all = sorted(p[1] / args.steps * 1e3 for p in passes)
        pass_ms = [round(p[1] / args.steps * 1e3, 4) for p in passes]
        # pass 0's steps as the caller saw them (the last entry of step_ms[0] is finish() + the closing
        # synchronize, not a step)
        s0 = np.array(step_ms[0][:-1])
        all_steps = np.concatenate([np.array(sm[:-1]) for sm in step_ms])
        l0 = lib_lat[0]
        tail = dict(
            step_ms_max=round(float(s0.max()), 4), step_ms_argmax=int(s0.argmax()),
            step_ms_p99=round(float(np.percentile(s0, 99)), 4), step_ms_median=round(float(np.median(s0)), 4),
            all_passes_step_ms_max=round(float(all_steps.max()), 4),
            all_passes_step_ms_p99=round(float(np.percentile(all_steps, 99)), 4),
            per_pass_step_ms_max=[round(float(max(sm[:-1])), 3) for sm in step_ms],
            library_call_ms=dict(p50=round(l0["p50_ms"], 4), p99=round(l0["p99_ms"], 4), max=round(l0["max_ms"], 4),
                                 max_call=l0["max_call"]),
            per_pass_library_call_ms_max=[round(l["max_ms"], 3) for l in lib_lat],
            worst_pass=(lambda w: dict(index=w, step_ms_max=round(float(max(step_ms[w][:-1])), 3),
                                       step_argmax=int(np.argmax(step_ms[w][:-1])),
                                       library_max_call=dict(ms=round(lib_lat[w]["max_ms"], 3), call=lib_lat[w]["max_call"],
                                                             phases_ms=lib_lat[w]["max_phase_ms"],
                                                             invol_switches=lib_lat[w]["max_invol_switches"],
                                                             cpu=lib_lat[w]["max_cpu"])))(
                int(np.argmax([max(sm[:-1]) for sm in step_ms]))),
            allocs_in_timed_passes=sum(l["allocs"] for l in lib_lat),
            invol_switches_in_timed_passes=sum(l["invol_switches"] for l in lib_lat),
            helper_invol_switches_in_timed_passes=sum(t["helper_invol_switches"] for t in ransac_tails),
            ransac_jobs_on_the_other_buffer=sum(t["skipped_buffers"] for t in ransac_tails),
            ransac_jobs_without_helpers=sum(t["solo_jobs"] for t in ransac_tails),
            cgroup_throttled=(None if throttle0 is None or throttle1 is None else
                              dict(periods=throttle1[0] - throttle0[0], ms=round((throttle1[1] - throttle0[1]) / 1e3, 3))),
            os_threads_in_process=os_threads,
            note="wall time of each step of pass 0 as the calling thread saw it (set_next_batch + track_event); "
                 "library_call_ms: the esvio_fe_track_event calls of pass 0 alone; worst_pass: the pass holding the slowest "
                 "step of all, its slowest call with the phases named")
        out = {
            "metric": "Mevents/s through time-surface+detect+track @640x480",
            "value": round(total_events / max_elapsed / 1e6, 3),
            "unit": "Mevents/s",
            "n_gpus": world,  # = the communicator's size (one rank per GPU)
            "devices_used": min(world, n_dev),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(max_elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if one_rig else "weak",
            "scaling_note": ("capability, not a speed-up: ONE stream, every batch time-sliced over the ranks for the SAE update — "
                             "two all-gathers of a full plane set per batch cost more than the single-GPU update they divide "
                             "(DESIGN.md section 6)" if time_split else
                             "capability, not a speed-up: ONE rig, the right camera's update + render on rank 1, its image broadcast "
                             "each frame" if cam_split else
                             "replicas: one independent stereo rig per GPU, no data-path collective; the one exchange is the "
                             "all-gather of the tracked-corner records (19.2 KB per rank per published frame)"),
            "vs_baseline": None,
            "dtype": "f64 timestamps / u8 images / int64 LK sums" if args.lk_accum == 1 else
                     "f64 timestamps / u8 images / f32 LK sums in the reference build's order",
            "data": "synthetic",
            "config": {
                "workload": "%s: stereo %dx%d %s stream, %.1f Mev/s per camera, "
                            "%g Hz batches, full SAE+TS+pyramid+LK(temporal,stereo)+Arc*+select, "
                            "max_cnt %d min_dist 10 flow_back 1 equalize %d freq %d, lk_accum %d (%s)"
                            % (workload_label(W, H, args.rate, world, args.split), W, H, args.stream, args.rate / 1e6, args.batch_hz, args.max_cnt, args.equalize, args.freq,
                               args.lk_accum, "exact integer LK sums" if args.lk_accum == 1 else
                               "float LK sums in the reference's x86 order"),
                "events_per_step_per_gpu": int(n_events / args.steps),
                "parallelism": ("left/right camera split, 1 rig on 2 GPUs" if cam_split else
                                "one stream time-sliced over %d GPUs (SAE update), tracking on rank 0" % world
                                if time_split else "1 rig per GPU") if world > 1 else "single GPU",
                "tracks_last_frame": n_tracks,
                "motion_comp": bool(args.mc),
                "pipelined_next_batch": bool(pipeline),
                "lazy_new_corner_stereo": bool(lazy),
                "host_threads": int(max(1, args.host_threads)),
                "launch_thread": bool(pipeline and args.launch_thread),
                "helper_idle_spin_us": int(os.environ.get("ESVIO_FE_HELPER_SPIN_US", "2000")),
                "batches_announced_ahead": int(args.ahead) if pipeline else 0,
                "track_exchange": ("library (ncclAllGather on the handle's communicator)" if comm_id is not None else
                                   "torch.distributed all_gather_into_tensor" if exch is not None else "none"),
                # compact copies of what sits at the end of the line
                "pass_ms": " ".join("%.4f" % v for v in pass_ms),
                "step_ms_max": tail["step_ms_max"],
                "step_ms_p99": tail["step_ms_p99"],
                "one_batch_in_flight_ms": (None if one_batch is None else
                                           "%.4f device-resident, %.4f host-registered, %.4f host-pageable"
                                           % (one_batch["device_resident_ms_per_step"], one_batch["host_registered_ms_per_step"],
                                              one_batch["host_pageable_ms_per_step"])),
                "low_cpu_ms": (None if low_cpu is None else "1 host thread %.4f; 2 CPUs / 2 threads %s"
                               % (low_cpu["one_host_thread_ms_per_step"],
                                  ("%.4f" % low_cpu["two_cpus_two_threads_ms_per_step"]) if "two_cpus_two_threads_ms_per_step" in low_cpu else "n/a")),
                "c5_shape_one_gpu": (None if c5_leg is None else "%.4f ms/step, %.0f Mev/s, k_time_surface4 %s of the HBM peak"
                                     % (c5_leg["ms_per_step"], c5_leg["value"],
                                        ("%.0f %%" % (100 * c5_leg["kernels"]["k_time_surface4"]["frac"]))
                                        if "k_time_surface4" in c5_leg["kernels"] else "n/a")),
                "other_lk_mode_ms": None if other_lk is None else "lk_accum %d: %.4f" % (other_lk["lk_accum"], other_lk["ms_per_step"]),
                "kernel_ms_per_step_summed": None if device_activity is None else device_activity["kernel_ms_per_step_summed"],
            },
            "roofline": (dict(roof, hbm_bound_kernels="sae_chain_c5_batch: k_tile_hist %.0f %%, k_tile_apply %.0f %% of the HBM peak "
                                                      "(uniform stream) at 6.7 M events per launch chain"
                                                      % (100 * sae_chain["uniform"]["k_tile_hist"]["frac"],
                                                         100 * sae_chain["uniform"]["k_tile_apply"]["frac"]))
                         if roof and sae_chain and all(k in sae_chain.get("uniform", {}) for k in ("k_tile_hist", "k_tile_apply"))
                         else roof),
            "cpu_baseline": cpu,
            "lk_modes": lk_modes,
            "kernels": kernels,
            "kernels_replay_schedule": kernels_pipe,
            "sae_chain_c5_batch": sae_chain,
            "c5_shape_one_gpu": c5_leg,
            "device_activity": device_activity,
            "host_ransac": host_ransac,
            # ---- the driver keeps the END of the line: what a reader needs beside `value` comes last
            "tail_latency": tail,
            "one_batch_in_flight": one_batch,
            "low_cpu": low_cpu,
            ("exact_sum_lk" if args.lk_accum == 2 else "float_order_lk"): other_lk,
            "host_resident_events": host_res,
            # the timed --steps region repeated over the continued stream (pass 0 = ms_per_step above)
            "repeats": dict(passes=repeats, ms_per_step=pass_ms,
                            median=round(ms_all[len(ms_all) // 2], 4), min=round(ms_all[0], 4),
                            max=round(ms_all[-1], 4),
                            value_median=round(passes[0][0] / args.steps / ms_all[len(ms_all) // 2] / 1e3, 3)),
        }
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
