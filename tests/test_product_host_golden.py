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


def test_multi_patch_side_dofs_and_mesh_file_name():
    """``MultiBSpline.getPatchSideDofs`` (tIGAr/BSplines.py:898-908): the side dofs of one patch in the global numbering,
    in the order of ``BSpline.getSideDofs``; ``generateMeshXMLFileName`` (tIGAr/common.py:88-93): the md5 name the
    reference derives from the communicator and the rank."""
    import hashlib
    from tigar_amd import BSplines as B, common as tc
    patches = [B.BSpline([2, 2], [B.uniformKnots(2, 0., 3., 3), B.uniformKnots(2, 0., 1., 2)]),
               B.BSpline([2, 3], [B.uniformKnots(2, -1., 1., 2), B.uniformKnots(3, 0., 2., 3)])]
    mb = B.MultiBSpline(patches)
    for patch in (0, 1):
        for direction in (0, 1):
            for side in (0, 1):
                for layers in (1, 2):
                    want = [mb.globalDofIndex(d, patch) for d in mb.splines[patch].getSideDofs(direction, side, layers)]
                    assert mb.getPatchSideDofs(patch, direction, side, layers) == want
    assert mb.getPatchSideDofs(1, 0, 0)[0] == patches[0].getNcp()          # patch 1 starts after patch 0's functions
    name = tc.generateMeshXMLFileName(tc.worldcomm)
    assert name == "mesh-" + hashlib.md5((repr(tc.worldcomm) + repr(tc.worldcomm.rank)).encode("utf-8")).hexdigest() + ".xml"
