"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads, exports every symbol
include/esvio_fe.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from esvio_amd import frontend as FE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "esvio_fe.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(esvio_fe_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert _declared_symbols() == sorted(FE.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol():
    L = FE.load_library()
    for s in _declared_symbols():
        assert hasattr(L, s), s
    assert b"gfx950" in L.esvio_fe_version()


def test_struct_layouts_match_header():
    # esvio_fe_event is 16 B; config/camera sizes as laid out by the C compiler (natural alignment)
    from esvio_amd.events import EVENT_DTYPE
    assert EVENT_DTYPE.itemsize == 16
    assert [EVENT_DTYPE.fields[n][1] for n in ("x", "y", "sec", "nsec", "polarity")] == [0, 2, 4, 8, 12]
    assert C.sizeof(FE.Camera) == 64
    assert C.sizeof(FE.Config) == 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 8 + 2 * 64
    assert C.sizeof(FE.Tracks) == 8 + 9 * 8


def test_create_fails_loudly_without_gpu(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    L = FE.load_library()
    h = C.c_void_p()
    rc = L.esvio_fe_create(C.byref(FE.make_config(640, 480)), C.byref(h))
    assert rc == -2 and not h  # ESVIO_FE_ENODEVICE: no CPU fallback exists
    with pytest.raises(FE.FrontendError):
        FE.FeatureTracker(FE.make_config(640, 480))


def test_create_rejects_bad_config():
    L = FE.load_library()
    h = C.c_void_p()
    for kw, rc_expected in ((dict(min_dist=2), -1), (dict(equalize=1), -4),
                            (dict(median_blur_kernel_size=1), -4), (dict(decay_ms=0.0), -1),
                            (dict(max_cnt=0), -1)):
        cfg = FE.make_config(640, 480, **kw)
        assert L.esvio_fe_create(C.byref(cfg), C.byref(h)) == rc_expected, kw
    assert L.esvio_fe_create(None, C.byref(h)) == -1
    assert L.esvio_fe_destroy(None) == -1
    assert L.esvio_fe_track_event(None, 0.0, None, 0, None, 0, 0, 1, None) == -1


def test_host_stages_run_without_gpu(oracle):
    """the two host-side stages of the boundary (no device needed) agree with the oracle"""
    cam = dict(fx=560.0, fy=555.0, cx=320.5, cy=239.0, k1=-0.31, k2=0.11, p1=4e-4, p2=-7e-4)
    rng = np.random.default_rng(0)
    for u, v in rng.uniform(0, 640, (50, 2)):
        a = FE.lift_projective(cam, u, v)
        b = oracle.lift_projective(cam, u, v)
        assert np.array_equal(a, b)
    nodist = dict(cam, k1=0.0, k2=0.0, p1=0.0, p2=0.0)
    assert np.allclose(FE.lift_projective(nodist, 320.5, 239.0), [0, 0, 1])
