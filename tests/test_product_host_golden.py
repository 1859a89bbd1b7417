"""The PRODUCT's host-side bookkeeping (tigar_amd.BSplines.uniformKnots / BSpline1: rows a-1, a-2 of
SURVEY.md section 8) directly against the golden vectors generated from the reference's own source
(tests/golden/make_golden.py).  Host arithmetic only -- nothing here touches the device, so this runs in
the CPU suite; the evaluation methods (device twins) are pinned by the -m gpu tests."""
import os
import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def test_product_uniform_knots_bit_exact():
    from tigar_amd.BSplines import uniformKnots
    g = _load("golden_knots.npz")
    meta = g["meta"]
    assert meta.shape[0] >= 100
    for k in range(meta.shape[0]):
        p, a, b, N, per, drop = meta[k]
        kv = uniformKnots(int(p), float(a), float(b), int(N), bool(per), int(drop))
        ref = g["k%d" % k]
        assert isinstance(kv, list) and len(kv) == len(ref)
        assert np.array_equal(np.array(kv, dtype=np.float64), ref)        # bit-exact


def test_product_uniform_knots_rejects_excess_continuity_drop():
    import pytest
    from tigar_amd.BSplines import uniformKnots
    with pytest.raises(ValueError):
        uniformKnots(2, 0.0, 1.0, 4, False, 2)      # the reference prints an error and exits (:24-26)


def test_product_bspline1_bookkeeping_bit_exact():
    from tigar_amd.BSplines import BSpline1
    g = _load("golden_bspline1.npz")
    n = int(g["ncases"])
    assert n >= 20
    for ci in range(n):
        pre = "c%d_" % ci
        s = BSpline1(int(g[pre + "p"]), g[pre + "knots"])
        assert s.nel == int(g[pre + "nel"])
        assert s.ncp == int(g[pre + "ncp"]) == s.getNcp()
        assert np.array_equal(s.uniqueKnots, g[pre + "uniqueKnots"])
        assert np.array_equal(s.multiplicities, g[pre + "multiplicities"])
        assert np.array_equal(s.ghostKnots, g[pre + "ghostKnots"])
        assert int(s.isDiscontinuous()) == int(g[pre + "disc"])
        assert np.array_equal(np.array([s.greville(i) for i in range(s.ncp)]), g[pre + "greville"])
