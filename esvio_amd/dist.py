"""Multi-GPU plumbing (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" for CPU tests).

The hot path shards by independent unit — one stereo rig (or one time-ordered stream) per rank,
no data-path collective.  The only exchange is the north-star's merge step: one all_gather per
published frame of the fixed-size tracked-corner records (2*max_cnt rows x 8 float32 =
19.2 KB at max_cnt 300), so that every rank / the estimator on rank 0 sees all rigs' tracks.
The payload is latency-bound; xGMI bandwidth is irrelevant.
"""
import numpy as np


def shard_units(n_units, world, rank):
    """round-robin assignment of independent rigs/streams to ranks"""
    return [u for u in range(n_units) if u % world == rank]


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class TrackExchange:
    """one (async) all_gather of track records per published frame."""

    def __init__(self, max_cnt, world, device="cpu", dist=None, stream=None):
        import torch
        self.torch = torch
        self.dist = dist
        self.world = world
        self.rows = 2 * max_cnt
        self.send = torch.zeros((self.rows, 8), dtype=torch.float32, device=device)
        # one flat receive buffer (all_gather_into_tensor: no per-rank copies afterwards)
        self.recv_flat = torch.zeros((world * self.rows, 8), dtype=torch.float32, device=device)
        self.recv = list(self.recv_flat.view(world, self.rows, 8).unbind(0))
        self.stream = stream
        self.pending = None
        # pinned staging (two, alternating) so that a tracker can pack straight into them and the
        # upload is a non-blocking copy on the side stream
        self.stage = None
        if str(device).startswith("cuda"):
            self.stage = [torch.zeros((self.rows, 8), dtype=torch.float32).pin_memory() for _ in range(2)]
            self.stage_ev = [None, None]
            self.stage_i = 0

    def submit(self, rec, async_op=True):
        """rec: (2*max_cnt, 8) float32 numpy block from node.pack_track_records"""
        t = self.torch.from_numpy(np.ascontiguousarray(rec, np.float32))
        if self.stream is not None:
            with self.torch.cuda.stream(self.stream):
                # Work.wait() of an NCCL collective orders the CURRENT stream behind it: taken
                # inside the side-stream scope, so that the copy below cannot overwrite `send`
                # while the previous all_gather is still reading it
                self.wait()
                self.send.copy_(t)
                self.pending = self.dist.all_gather_into_tensor(self.recv_flat, self.send, async_op=async_op)
        else:
            self.wait()
            self.send.copy_(t)
            self.pending = self.dist.all_gather_into_tensor(self.recv_flat, self.send, async_op=async_op)
        return self.pending

    def submit_tracker(self, ft, async_op=True):
        """like submit(ft.pack_track_records()), without the detour through pageable memory"""
        if self.stage is None:
            return self.submit(ft.pack_track_records(), async_op)
        i = self.stage_i
        self.stage_i ^= 1
        if self.stage_ev[i] is not None:
            self.stage_ev[i].synchronize()  # its previous upload has long finished
        ft.pack_track_records(self.stage[i].numpy())
        with self.torch.cuda.stream(self.stream) if self.stream is not None else _null():
            self.wait()  # (inside the scope: orders THIS stream behind the previous collective)
            self.send.copy_(self.stage[i], non_blocking=True)
            ev = self.torch.cuda.Event()
            ev.record()
            self.stage_ev[i] = ev
            self.pending = self.dist.all_gather_into_tensor(self.recv_flat, self.send, async_op=async_op)
        return self.pending

    def wait(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None

    def result(self):
        """(world, rows, 8) array; rows with id (column 3) < 0 are padding"""
        self.wait()
        return np.stack([r.detach().cpu().numpy() for r in self.recv])


def merged_point_cloud(gathered):
    """concatenate every rank's valid PointCloud rows, tagging the rig (rank) in a 9th column"""
    out = []
    for rk, block in enumerate(gathered):
        valid = block[block[:, 3] >= 0]
        out.append(np.c_[valid, np.full(len(valid), rk, np.float32)])
    return np.concatenate(out, 0) if out else np.zeros((0, 9), np.float32)


class CameraSplitRig:
    """BASELINE config C4: one stereo rig on two GPUs, rank 0 owns the left camera (SAE, time
    surface, temporal LK, detection, stereo LK), rank 1 owns the right camera's SAE + time surface
    and ships its 1 byte/pixel image every frame (W*H bytes: 307 KB at 640x480 — ~2 us of one xGMI
    link; latency-, not bandwidth-bound).  The left rank's tracks equal the single-GPU result bit
    for bit because the two cameras never share SAE state (event_detector.h:74-79) and both time
    surfaces use the LEFT batch's end time (feature_tracker.cpp:367-368)."""

    def __init__(self, tracker, rank, dist, device="cpu"):
        import torch
        assert dist.get_world_size() == 2
        self.torch = torch
        self.ft = tracker
        self.rank = rank
        self.dist = dist
        W, H = tracker.cfg.width, tracker.cfg.height
        self.img = torch.zeros((H, W), dtype=torch.uint8, device=device)
        self.on_gpu = device != "cpu"
        self._empty = None

    def track(self, cur_time, event_left, event_right, pub_this_frame):
        """cur_time = last LEFT event's stamp (node:190).  Returns the tracker on rank 0."""
        if self.rank == 1:
            self.ft.detector.createSAE_right(event_right)
            if self.on_gpu:
                self.ft._hd.check(self.ft._hd.L.esvio_fe_sae_to_time_surface(
                    self.ft._hd.h, 1, float(cur_time), None))
                self.ft.export_image(1, dst=self.img.data_ptr())
            else:
                self.img.copy_(self.torch.from_numpy(self.ft.detector.SAEtoTimeSurface_right(cur_time)))
        self.dist.broadcast(self.img, src=1)
        if self.rank == 0:
            if self.on_gpu:
                self.torch.cuda.current_stream().synchronize()
                self.ft.import_image(1, self.img.data_ptr())
            else:
                self.ft.import_image(1, self.img.numpy())
            if self._empty is None:
                import numpy as np
                from .events import EVENT_DTYPE
                self._empty = np.zeros(0, EVENT_DTYPE)
            empty = (0, 0) if isinstance(event_left, tuple) else self._empty
            self.ft.trackEvent(cur_time, event_left, empty, pub_this_frame, copy=False)
            return self.ft
        return None


def time_slice(n, world, rank):
    """[lo, hi) of slice `rank` when n stream-ordered events are cut into `world` consecutive slices"""
    return (n * rank) // world, (n * (rank + 1)) // world


class TimeSlicedSae:
    """BASELINE config C5 / SURVEY.md §8e.2: ONE stream, every batch cut into `world` consecutive time
    slices, rank r applies slice r to the SAE (createSAE_left/right, event_detector.cc:149-166), two
    all-gathers compose the planes, rank 0 (the rank the host feeds) runs the rest of trackEvent on
    them.  Exact for any timestamps, ties and duplicates across the cuts:

    * `L[p] = t` is unconditional (:158): a slice's last event time per (camera, pixel, polarity)
      does not depend on what came before -> sae_slice_last, all-gather #1;
    * the pass rule (:155) reads only L: with the exact carried-in L (planes before the batch
      overlaid with the earlier slices) every decision of a slice is the sequential loop's ->
      sae_slice_apply gives the time of the slice's last PASSING event, all-gather #2;
    * planes after the batch = planes before it overlaid with the slices in order -> sae_slice_commit
      on every rank (each needs them as the next batch's starting point).

    `engine` is anything with sae_plane_doubles / sae_slice_last / sae_slice_apply / sae_slice_commit
    and (rank 0) trackEvent: a frontend.FeatureTracker on the GPU, or the oracle-backed stand-in the
    CPU tests use.  Cost: two all-gathers of a full plane set (2 cameras x W*H x 2 doubles) per
    batch — at 1280x720 29.5 MB per rank — which is more than the SAE update of the slices saves on
    one node; it is the capability SURVEY names, not a speed-up at these sizes."""

    def __init__(self, engine, rank, world, dist, device="cpu"):
        import torch
        self.torch = torch
        self.ft = engine
        self.rank, self.world, self.dist = rank, world, dist
        nd = engine.sae_plane_doubles()
        self.nd = nd
        self.on_gpu = device != "cpu"
        kw = dict(dtype=torch.float64, device=device)
        self.mine = torch.empty(nd, **kw)
        self.last_all = torch.empty(world * nd, **kw)
        self.s_all = torch.empty(world * nd, **kw)

    def _arg(self, t, n_sets=1):
        return (t.data_ptr(), n_sets) if self.on_gpu else t.numpy()

    @staticmethod
    def _cut(ev, lo, hi):
        if isinstance(ev, tuple):  # (device pointer, count) of 16 B records
            return (ev[0] + 16 * lo, hi - lo)
        return ev[lo:hi]

    @staticmethod
    def _len(ev):
        return ev[1] if isinstance(ev, tuple) else len(ev)

    def apply_batch(self, event_left, event_right):
        """the batch's SAE update, sliced; afterwards every rank's planes hold the batch"""
        d, r, w = self.dist, self.rank, self.world
        L = self._cut(event_left, *time_slice(self._len(event_left), w, r))
        R = self._cut(event_right, *time_slice(self._len(event_right), w, r))
        self.ft.sae_slice_last(L, R, self._arg(self.mine))
        if self.on_gpu:
            self.torch.cuda.current_stream().synchronize()
        d.all_gather_into_tensor(self.last_all, self.mine)
        if self.on_gpu:
            self.torch.cuda.current_stream().synchronize()
        self.ft.sae_slice_apply(L, R, self._arg(self.last_all, r), r, self._arg(self.mine))
        d.all_gather_into_tensor(self.s_all, self.mine)
        if self.on_gpu:
            self.torch.cuda.current_stream().synchronize()
        self.ft.sae_slice_commit(self._arg(self.last_all, w), self._arg(self.s_all, w), w)

    def track(self, cur_time, event_left, event_right, pub_this_frame):
        """cur_time = last LEFT event's stamp (node:190).  Returns the tracker on rank 0."""
        self.apply_batch(event_left, event_right)
        if self.rank == 0:
            self.ft.trackEvent(cur_time, event_left, event_right, pub_this_frame, copy=False)
            return self.ft
        return None
