"""CPU: the distribution of LK positions between exact accumulation (= the GPU kernel, bit for bit,
see test_parity_gpu.py::test_lk_parity) and the float accumulation orders of the reference's
OpenCV builds stays inside the measured bands of tests/lk_orders.py."""
import numpy as np

import lk_orders


def test_exact_sums_against_float_orders(oracle):
    rows = lk_orders.table(oracle, with_scene=True)
    assert len(rows) >= 8
    for name, d_simd, d_scalar, d_between in rows:
        print(name, "SIMD128:", d_simd, "scalar:", d_scalar)
        assert d_simd["n"] > 100, name
        kind = "scene" if name.startswith("scene") else "texture"
        lk_orders.assert_band(d_simd, 2, kind)
        lk_orders.assert_band(d_scalar, 0, kind)
        # the exact sums are what both float orders approximate: the bulk is never farther from the
        # SIMD order than the two float orders are from each other
        assert d_simd["p90"] <= d_between["p90"] + 1e-6, (name, d_simd, d_between)


def test_simd_order_mode_sums_the_same_terms(oracle):
    """accum=2 only reorders float additions: on an image whose derivative products are small
    integers every order is exact and the three modes agree bit for bit"""
    W, H = 96, 80
    yy, xx = np.mgrid[0:H, 0:W]
    prev = ((xx // 6 + yy // 5) % 2 * 8 + 100).astype(np.uint8)  # low-contrast checker: tiny sums
    nxt = np.roll(prev, 1, axis=1)
    pts = np.stack([np.linspace(20, W - 20, 12), np.linspace(20, H - 20, 12)], 1).astype(np.float32)
    out = [oracle.lk(prev, nxt, pts, pts.copy(), max_level=0, flags=0, accum=a) for a in (0, 1, 2)]
    for p, s in out[1:]:
        assert np.array_equal(s, out[0][1])
        assert np.array_equal(p.view(np.uint32), out[0][0].view(np.uint32))
