"""Every parameter set the reference ships (tests/shipped_configs.py: sensor sizes, feature budgets,
equalize / motion-compensation switches, the real cameras' intrinsics and distortion — k1 down to
-0.41, tangential terms up to 0.055) through the C ABI against the oracle: the event node's
trackEvent (motion-compensated where the config says so) and, for the ESVIO configs, the image
node's trackImage at the frame camera's size with max_cnt_img / min_dist_img."""
import numpy as np
import pytest

from esvio_amd import frontend as FE
from esvio_amd.events import event_times
from esvio_amd.synth import ImageStream, SceneStream
from shipped_configs import SHIPPED
from test_parity_gpu import _compare_tracks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(SHIPPED))
def test_event_node_parameters(oracle, name):
    p = SHIPPED[name]
    W, H = p["ev"]
    kw = dict(max_cnt=p["max_cnt"], min_dist=p["min_dist"], equalize=p["equalize"], flow_back=1,
              f_threshold=1.0, f_ransac=1, cams=list(p["ev_cams"]))
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    s = SceneStream(W, H, rate=2e6 if W < 400 else 5e6, seed=len(name))
    K = p["ev_cams"][0]
    for f in range(5):
        L, R, _ = s.next_batch()
        t = event_times(L)
        pub = f % 2 == 0
        if p["mc"]:  # Do_motion_correction: 1 — a Motion_correction_value that passes the 5 m/s^2 gate
            mv = dict(t1=t[0] + 0.8 * (t[-1] - t[0]), v=(0.6, -0.2, 0.1), v_pre=(0.5, -0.15, 0.05),
                      accel=(4.0, 5.0, 3.0) if f % 2 else (0.5, 0.2, 0.1), omega=(0.3, -0.4, 0.6),
                      fx=K["fx"], fy=K["fy"], cx=K["cx"], cy=K["cy"])
            ft.trackEvent(t[-1], L, R, pub, measurements=FE.make_motion(**mv))
            r = tr.track_event(t[-1], L, R, pub, motion=oracle.make_motion(**mv))
        else:
            ft.trackEvent(t[-1], L, R, pub)
            r = tr.track_event(t[-1], L, R, pub)
        _compare_tracks(ft, r, (name, f))
    assert len(ft.ids) > p["max_cnt"] // 3
    # the undistorted coordinates really went through this camera's distortion model
    un = ft.cur_un_pts
    assert np.isfinite(un).all() and np.abs(un).max() < 5.0
    ft.close()


@pytest.mark.parametrize("name", sorted(n for n in SHIPPED if "img" in SHIPPED[n]))
def test_image_node_parameters(oracle, name):
    p = SHIPPED[name]
    W, H = p["img"]
    kw = dict(max_cnt=p["max_cnt_img"], min_dist=p["min_dist_img"], equalize=p["equalize"], flow_back=1,
              f_threshold=1.0, f_ransac=1, cams=list(p["img_cams"]))
    ft = FE.FeatureTracker(FE.make_config(W, H, **kw))
    tr = oracle.Tracker(oracle.make_config(W, H, **kw))
    s = ImageStream(W, H, velocity=(4, -2), disparity=12, seed=len(name))
    for f in range(3):
        L, R, t = s.next_frame()
        ft.trackImage(t, L, R, f != 1)
        _compare_tracks(ft, tr.track_image(t, L, R, f != 1), (name, "image", f))
    assert len(ft.ids) > p["max_cnt_img"] // 3 and len(ft.ids_right) > p["max_cnt_img"] // 6
    ft.close()
