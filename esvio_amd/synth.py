"""Synthetic event streams for the parity tests and the benchmark.

Two value distributions:

* ``uniform_batch`` — uniform Poisson noise: x,y uniform, exponential inter-arrival,
  p ~ Bernoulli(1/2).  Right for SAE / time-surface throughput; yields almost no Arc*
  corners and an untrackable image.
* ``SceneStream`` — events drawn from the moving edges of translating rectangles
  (polarity = sign of the brightness change at the edge), a few events per edge crossing,
  plus uniform noise; the right camera sees the same scene shifted by a constant
  disparity, so stereo LK has a ground truth.

Timestamps are non-decreasing and quantised to 1 us like a DAVIS.  Everything here is the
build's own code; the reference ships no generator.
"""
import numpy as np

from .events import make_events


def uniform_batch(W, H, n, t0_us, dur_us, rng):
    """n uniform events in [t0_us, t0_us+dur_us), sorted by time."""
    t = np.sort(rng.integers(0, dur_us, size=n, dtype=np.int64)) + t0_us
    x = rng.integers(0, W, size=n)
    y = rng.integers(0, H, size=n)
    p = rng.integers(0, 2, size=n)
    return make_events(x, y, t, p)


class PoissonStream:
    """Homogeneous Poisson events on both cameras (BASELINE's "synthetic Poisson event stream"):
    uniform pixels, uniform polarity, a Poisson count per batch, times uniform in the batch — no
    structure, so the corners Arc* finds and LK follows are noise.  Same interface as SceneStream."""

    def __init__(self, W=640, H=480, rate=5e6, batch_hz=30.0, t0_us=1_000_000_000, seed=12345):
        self.W, self.H, self.rate = W, H, rate
        self.dur_us = int(round(1e6 / batch_hz))
        self.t_us = int(t0_us)
        self.rng = np.random.default_rng(seed)

    def next_batch(self):
        lam = self.rate * self.dur_us * 1e-6
        nl, nr = (max(1, int(self.rng.poisson(lam))) for _ in range(2))
        L = uniform_batch(self.W, self.H, nl, self.t_us, self.dur_us, self.rng)
        R = uniform_batch(self.W, self.H, nr, self.t_us, self.dur_us, self.rng)
        self.t_us += self.dur_us
        return L, R, self.t_us


class SceneStream:
    """Stereo scene stream.  ``next_batch()`` -> (left, right, t_end_us)."""

    def __init__(self, W=640, H=480, rate=5e6, batch_hz=30.0, n_rect=28, disparity=12,
                 noise_frac=0.07, events_per_crossing=3, t0_us=1_000_000_000, seed=12345,
                 speed=(180.0, 420.0), size=(50.0, 150.0), ego=True, mover_frac=0.15,
                 ego_flip_batches=12):
        """ego=True: a static scene seen from a laterally translating camera — every rectangle
        moves along one common direction with speed ~ 1/depth (one consistent epipolar geometry,
        what a VIO front-end actually sees), except a fraction `mover_frac` of independent movers
        (outliers for the F-RANSAC); the camera reverses every `ego_flip_batches` batches so the
        scene stays in view.  ego=False: every rectangle moves independently (no common F)."""
        self.W, self.H = W, H
        self.rate = rate
        self.dur_us = int(round(1e6 / batch_hz))
        self.disparity = disparity
        self.noise_frac = noise_frac
        self.epc = events_per_crossing
        self.t_us = int(t0_us)
        self.rng = np.random.default_rng(seed)
        r = self.rng
        self.n_rect = n_rect
        self.c = np.stack([r.uniform(0.1 * W, 0.9 * W, n_rect), r.uniform(0.1 * H, 0.9 * H, n_rect)], 1)
        ang = r.uniform(0, 2 * np.pi, n_rect)
        spd = r.uniform(speed[0], speed[1], n_rect)
        self.v = np.stack([np.cos(ang) * spd, np.sin(ang) * spd], 1)  # px/s
        self.ego = ego
        self.ego_flip = ego_flip_batches
        self.batch_idx = 0
        self.is_mover = np.ones(n_rect, bool)
        if ego:
            cam_ang = r.uniform(0, 2 * np.pi)
            cam_dir = np.array([np.cos(cam_ang), np.sin(cam_ang)])
            self.is_mover = r.random(n_rect) < mover_frac
            for k in range(n_rect):
                if not self.is_mover[k]:
                    self.v[k] = cam_dir * spd[k]  # speed ~ focal * v_cam / depth
        self.half = np.stack([r.uniform(size[0], size[1], n_rect) / 2, r.uniform(size[0], size[1], n_rect) / 2], 1)
        self.rot = r.uniform(0, np.pi / 2, n_rect)
        self.sign = r.choice([-1.0, 1.0], n_rect)
        # auto-scale events per crossing so the mean rate is close to `rate`
        sweep = 0.0
        for k in range(n_rect):
            for (A, B, nrm) in self._edges(k):
                sweep += np.linalg.norm(B - A) * abs(float(nrm @ self.v[k]))
        self._sweep_px_per_s = sweep
        target_scene = rate * (1.0 - noise_frac)
        self.epc_f = max(1.0, target_scene / max(sweep, 1.0))

    def _edges(self, k):
        ca, sa = np.cos(self.rot[k]), np.sin(self.rot[k])
        ex = np.array([ca, sa])
        ey = np.array([-sa, ca])
        hx, hy = self.half[k]
        c = self.c[k]
        P = [c - hx * ex - hy * ey, c + hx * ex - hy * ey, c + hx * ex + hy * ey, c - hx * ex + hy * ey]
        nrm = [-ey, ex, ey, -ex]
        return [(P[i], P[(i + 1) % 4], nrm[i]) for i in range(4)]

    def _edge_events(self, A, B, nrm, v, sign, dur_s, xs_off):
        """pixels swept by edge AB moving with velocity v during dur_s -> (x, y, t_rel_s, pol)."""
        vn = float(nrm @ v)
        if abs(vn) < 1e-6:
            return None
        e = B - A
        L = float(np.linalg.norm(e))
        e = e / L
        d = v * dur_s
        pts = np.stack([A, B, A + d, B + d])
        x0 = int(np.floor(pts[:, 0].min())) - 1
        x1 = int(np.ceil(pts[:, 0].max())) + 1
        y0 = int(np.floor(pts[:, 1].min())) - 1
        y1 = int(np.ceil(pts[:, 1].max())) + 1
        x0, y0 = max(x0, 0), max(y0, 0)
        x1, y1 = min(x1, self.W - 1 + abs(xs_off)), min(y1, self.H - 1)
        if x1 < x0 or y1 < y0:
            return None
        xs, ys = np.meshgrid(np.arange(x0, x1 + 1), np.arange(y0, y1 + 1))
        px = xs.ravel().astype(np.float64)
        py = ys.ravel().astype(np.float64)
        tn = ((px - A[0]) * nrm[0] + (py - A[1]) * nrm[1]) / vn  # crossing time (s)
        ok = (tn >= 0) & (tn < dur_s)
        ax = A[0] + v[0] * tn
        ay = A[1] + v[1] * tn
        s = (px - ax) * e[0] + (py - ay) * e[1]
        ok &= (s >= 0) & (s <= L)
        if not ok.any():
            return None
        pol = 1 if sign * vn > 0 else 0
        return xs.ravel()[ok], ys.ravel()[ok], tn[ok], pol

    def next_batch(self):
        dur_s = self.dur_us * 1e-6
        r = self.rng
        out = []
        for cam, xoff in ((0, 0), (1, -self.disparity)):
            X, Y, T, P = [], [], [], []
            for k in range(self.n_rect):
                for (A, B, nrm) in self._edges(k):
                    res = self._edge_events(A, B, nrm, self.v[k], self.sign[k], dur_s, xoff)
                    if res is None:
                        continue
                    x, y, tn, pol = res
                    x = x + xoff
                    keep = (x >= 0) & (x < self.W)
                    x, y, tn = x[keep], y[keep], tn[keep]
                    # events per crossing: floor(epc_f) + Bernoulli(frac)
                    base = int(np.floor(self.epc_f))
                    cnt = base + (r.random(x.shape[0]) < (self.epc_f - base)).astype(np.int64)
                    rep = np.repeat(np.arange(x.shape[0]), cnt)
                    j = np.concatenate([np.arange(c) for c in cnt]) if rep.size else np.zeros(0, np.int64)
                    tt = tn[rep] + j * r.uniform(150e-6, 400e-6, rep.shape[0])
                    X.append(x[rep]); Y.append(y[rep]); T.append(tt); P.append(np.full(rep.shape[0], pol))
            n_scene = sum(a.shape[0] for a in X)
            n_noise = int(round(self.noise_frac / max(1e-9, 1 - self.noise_frac) * n_scene))
            X.append(r.integers(0, self.W, n_noise)); Y.append(r.integers(0, self.H, n_noise))
            T.append(r.uniform(0, dur_s, n_noise)); P.append(r.integers(0, 2, n_noise))
            x = np.concatenate(X); y = np.concatenate(Y); t = np.concatenate(T); p = np.concatenate(P)
            t_us = np.clip(np.floor(t * 1e6).astype(np.int64), 0, self.dur_us - 1) + self.t_us
            order = np.argsort(t_us, kind="stable")
            out.append(make_events(x[order], y[order], t_us[order], p[order]))
        # advance scene; independent movers bounce at the borders, the camera reverses periodically
        self.c += self.v * dur_s
        self.batch_idx += 1
        if self.ego and self.batch_idx % self.ego_flip == self.ego_flip // 2:
            self.v[~self.is_mover] *= -1.0
        for k in range(self.n_rect):
            if not self.is_mover[k]:
                continue
            for a, lim in ((0, self.W), (1, self.H)):
                if self.c[k, a] < 0.05 * lim and self.v[k, a] < 0:
                    self.v[k, a] = -self.v[k, a]
                if self.c[k, a] > 0.95 * lim and self.v[k, a] > 0:
                    self.v[k, a] = -self.v[k, a]
        self.t_us += self.dur_us
        return out[0], out[1], self.t_us


class ImageStream:
    """Stereo grey-image stream for the image front-end (trackImage): a fixed random texture
    (smoothed noise + rectangles of random brightness) seen through a window that moves by an
    integer velocity per frame; the right image is the left one shifted by `disparity` pixels;
    a little per-frame sensor noise.  ``next_frame()`` -> (left, right, t_seconds)."""

    def __init__(self, W=640, H=480, velocity=(3, 2), disparity=8, noise=2.0, n_rect=60, fps=20.0,
                 t0=100.0, seed=7):
        self.W, self.H = W, H
        self.v = velocity
        self.disp = disparity
        self.noise = noise
        self.dt = 1.0 / fps
        self.t = t0
        self.k = 0
        self.rng = np.random.default_rng(seed)
        r = self.rng
        pad = 256
        TH, TW = H + 2 * pad, W + 2 * pad + disparity
        tex = r.normal(0, 1, (TH, TW))
        for _ in range(3):  # cheap separable smoothing
            tex = (tex + np.roll(tex, 1, 0) + np.roll(tex, -1, 0) + np.roll(tex, 1, 1) + np.roll(tex, -1, 1)) / 5
        tex = 128 + 60 * tex / tex.std()
        for _ in range(n_rect * (TH * TW) // (W * H)):
            w, h = r.integers(12, 70, 2)
            x, y = r.integers(0, TW - w), r.integers(0, TH - h)
            tex[y:y + h, x:x + w] = 0.35 * tex[y:y + h, x:x + w] + 0.65 * r.uniform(20, 235)
        self.tex = tex
        self.pad = pad

    def next_frame(self):
        ox = self.pad + (self.v[0] * self.k) % 128
        oy = self.pad + (self.v[1] * self.k) % 128
        r = self.rng

        def view(x0):
            img = self.tex[oy:oy + self.H, x0:x0 + self.W] + r.normal(0, self.noise, (self.H, self.W))
            return np.clip(np.rint(img), 0, 255).astype(np.uint8)

        left, right = view(ox + self.disp), view(ox)
        t = self.t
        self.t += self.dt
        self.k += 1
        return left, right, t
