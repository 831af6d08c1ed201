"""Oracle-backed stand-in for a handle's time-slice entry points (esvio_fe_sae_slice_last / _apply /
_commit), so that esvio_amd.dist.TimeSlicedSae and the composition rule can be tested on CPU."""
import numpy as np

NONE = -1.0


class OracleSliceEngine:
    def __init__(self, oracle, W, H):
        self.O, self.W, self.H, self.P = oracle, W, H, W * H
        self.det = oracle.Detector(W, H)

    def sae_plane_doubles(self):
        return 4 * self.P

    # plane set layout: [(cam * P + px) * 2 + pol], like the library's double2 arrays
    def _pack(self, det, which):
        out = np.empty((2, self.P, 2), np.float64)
        for cam in (0, 1):
            L0, L1, S0, S1 = det.get_sae(cam)
            a, b = (L0, L1) if which == "L" else (S0, S1)
            out[cam, :, 0] = a.reshape(-1)
            out[cam, :, 1] = b.reshape(-1)
        return out.reshape(-1)

    def _set(self, det, Lp, Sp):
        Lp, Sp = Lp.reshape(2, self.P, 2), Sp.reshape(2, self.P, 2)
        sh = (self.H, self.W)
        for cam in (0, 1):
            det.set_sae(cam, Lp[cam, :, 0].reshape(sh).copy(), Lp[cam, :, 1].reshape(sh).copy(),
                        Sp[cam, :, 0].reshape(sh).copy(), Sp[cam, :, 1].reshape(sh).copy())

    @staticmethod
    def _overlay(dst, src):
        m = src != NONE
        dst[m] = src[m]

    def sae_slice_last(self, left, right, out):
        d = self.O.Detector(self.W, self.H)
        none = np.full(4 * self.P, NONE)
        self._set(d, none, none)
        d.create_sae(0, left)
        d.create_sae(1, right)
        out[:] = self._pack(d, "L")

    def sae_slice_apply(self, left, right, last_before, n_before, s_out):
        Lin = self._pack(self.det, "L")
        for k in range(n_before):
            self._overlay(Lin, np.asarray(last_before).reshape(-1)[k * 4 * self.P:(k + 1) * 4 * self.P])
        d = self.O.Detector(self.W, self.H)
        self._set(d, Lin, np.full(4 * self.P, NONE))
        d.create_sae(0, left)
        d.create_sae(1, right)
        s_out[:] = self._pack(d, "S")

    def sae_slice_commit(self, last_all, s_all, n):
        Lp, Sp = self._pack(self.det, "L"), self._pack(self.det, "S")
        nd = 4 * self.P
        for k in range(n):
            self._overlay(Lp, np.asarray(last_all).reshape(-1)[k * nd:(k + 1) * nd])
            self._overlay(Sp, np.asarray(s_all).reshape(-1)[k * nd:(k + 1) * nd])
        self._set(self.det, Lp, Sp)


def adversarial_batches(W, H, n_batches, seed, n=6000):
    """per-pixel histories that straddle any cut: few hot pixels (so every slice continues another
    slice's pixel), equal stamps, 1 ms bursts inside the 10 ms refractory window, polarity flips,
    stamps that go backwards, plus uniform background; left and right differ"""
    from esvio_amd.events import make_events
    rng = np.random.default_rng(seed)
    out = []
    t0 = 7_000_000
    for b in range(n_batches):
        cams = []
        for cam in (0, 1):
            hot = rng.integers(0, [W, H], (12, 2))
            pick = rng.integers(0, 12, n)
            x = np.where(rng.random(n) < 0.7, hot[pick, 0], rng.integers(0, W, n))
            y = np.where(rng.random(n) < 0.7, hot[pick, 1], rng.integers(0, H, n))
            t = t0 + np.cumsum(rng.choice([0, 0, 300, 1000, 12_000, -2500], n, p=[.2, .2, .2, .2, .1, .1]))
            t = np.maximum(t, 1)
            p = (np.cumsum(rng.random(n) < 0.15) + rng.integers(0, 2)) % 2
            cams.append(make_events(x, y, t, p))
        out.append(tuple(cams))
        t0 += 40_000
    return out
