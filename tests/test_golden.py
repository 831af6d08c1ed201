"""Committed regression vectors (tests/golden/scene_192x144.npz and scene_192x144_f32.npz — one per LK
accumulation mode — made by tests/golden/make_golden.py
from the CPU oracle — the reference has no fixtures and cannot run here).  CPU: the oracle still
reproduces them.  GPU: the HIP path reproduces them bit for bit through the C ABI.
At BASELINE C3's own size (640x480, the bench stream) the committed fixture is a set of SHA-256 digests of the same
quantities (tests/golden/c3_640x480_digests.json, tests/golden/make_c3_digests.py)."""
import importlib.util
import json
import os

import numpy as np
import pytest

from esvio_amd.events import EVENT_DTYPE

GDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["scene_192x144_f32.npz", "scene_192x144.npz"]  # lk_accum 2 (the default mode), lk_accum 1
KEYS = ("ids", "track_cnt", "cur_pts", "cur_un_pts", "pts_velocity", "ids_right", "cur_right_pts",
        "cur_un_right_pts", "right_pts_velocity")


def _load(name):
    z = np.load(os.path.join(GDIR, name))
    kw = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg_")}
    return z, int(z["W"]), int(z["H"]), int(z["n_batches"]), kw


def _events(z, name):
    return np.ascontiguousarray(z[name]).view(EVENT_DTYPE).reshape(-1)


def _check(z, b, ts_l, ts_r, flags, res):
    assert np.array_equal(ts_l, z["tsL%d" % b]) and np.array_equal(ts_r, z["tsR%d" % b])
    assert np.array_equal(flags, z["flags%d" % b])
    for k in KEYS:
        a, e = getattr(res, k), z["%s%d" % (k, b)]
        assert a.shape == e.shape and np.array_equal(a, e), (b, k)


@pytest.mark.parametrize("fixture", FIXTURES)
def test_oracle_reproduces_golden(oracle, fixture):
    z, W, H, NB, kw = _load(fixture)
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    for b in range(NB):
        L, R = _events(z, "L%d" % b), _events(z, "R%d" % b)
        r = tr.track_event(float(z["t%d" % b]), L, R, bool(z["pub%d" % b]))
        _check(z, b, tr.time_surface(0), tr.time_surface(1), tr.detector().corner_flags(L), r)
    det = tr.detector()
    for cam in (0, 1):
        for name, p in zip(("L0", "L1", "S0", "S1"), det.get_sae(cam)):
            assert np.array_equal(p, z["sae_cam%d_%s" % (cam, name)])


@pytest.mark.gpu
@pytest.mark.parametrize("fixture", FIXTURES)
def test_gpu_reproduces_golden(fixture):
    from esvio_amd import frontend as FE
    z, W, H, NB, kw = _load(fixture)
    assert kw["lk_accum"] == (2 if "f32" in fixture else 1)
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    for b in range(NB):
        L, R = _events(z, "L%d" % b), _events(z, "R%d" % b)
        ft.trackEvent(float(z["t%d" % b]), L, R, bool(z["pub%d" % b]))
        _check(z, b, ft.gettimesurface(0), ft.gettimesurface(1), ft.detector.isCorner(L), ft)
    for cam in (0, 1):
        for name, p in zip(("L0", "L1", "S0", "S1"), ft.detector.get_sae(cam)):
            assert np.array_equal(p, z["sae_cam%d_%s" % (cam, name)])
    ft.close()


def _c3():
    spec = importlib.util.spec_from_file_location("make_c3_digests", os.path.join(GDIR, "make_c3_digests.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m, json.load(open(os.path.join(GDIR, "c3_640x480_digests.json")))


def _same_digests(got, want, tag):
    assert len(got) == len(want)
    for f, (g, w) in enumerate(zip(got, want)):
        diff = [k for k in w if g[k] != w[k]]
        assert not diff, (tag, "frame", f, diff)


@pytest.mark.parametrize("lk_accum", [2, 1])
def test_oracle_reproduces_c3_digests(oracle, lk_accum):
    """8 frames of the bench stream at 640x480: inputs, time surfaces, corner flags and every result vector hash to
    the committed digests"""
    m, ref = _c3()
    assert (ref["W"], ref["H"], ref["seed"]) == (m.W, m.H, m.SEED)
    tr = oracle.Tracker(oracle.make_config(m.W, m.H, lk_accum=lk_accum, **ref["cfg"]))
    got = m.run(lambda t, L, R, pub: tr.track_event(t, L, R, pub), tr.time_surface, lambda L: tr.detector().corner_flags(L))
    _same_digests(got, ref["modes"][str(lk_accum)], ("oracle", lk_accum))
    assert got[-1]["n_left"] > 100 and got[-1]["n_right"] > 50


@pytest.mark.gpu
@pytest.mark.parametrize("lk_accum", [2, 1])
def test_gpu_reproduces_c3_digests(lk_accum):
    from esvio_amd import frontend as FE
    m, ref = _c3()
    ft = FE.FeatureTracker(FE.make_config(m.W, m.H, lk_accum=lk_accum, **ref["cfg"]))
    got = m.run(lambda t, L, R, pub: ft.trackEvent(t, L, R, pub), ft.gettimesurface, ft.detector.isCorner)
    ft.close()
    _same_digests(got, ref["modes"][str(lk_accum)], ("gpu", lk_accum))
