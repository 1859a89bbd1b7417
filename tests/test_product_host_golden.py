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


def test_side_dof_lists_drop_their_array_when_edited():
    """``getSideDofs`` returns a list (the reference's type) that carries the numpy array it came from, so that
    ``addZeroDofs`` needs no pass over 67 000 Python integers per face; any edit through the list interface drops the
    array and the list's own content is what counts."""
    from tigar_amd import BSplines as B
    from tigar_amd.common import AbstractExtractionGenerator as G
    s = B.BSpline([2, 2], [B.uniformKnots(2, 0., 1., 4)] * 2)
    dofs = s.getSideDofs(0, 0)
    assert isinstance(dofs, list) and dofs.array is not None and G._index_array(dofs) is dofs.array
    edits = [lambda d: d.sort(reverse=True), lambda d: d.reverse(), lambda d: d.__setitem__(0, 99), lambda d: d.append(7),
             lambda d: d.extend([1, 2]), lambda d: d.pop(), lambda d: d.remove(d[1]), lambda d: d.insert(0, 5),
             lambda d: d.__delitem__(0), lambda d: d.clear(), lambda d: d.__iadd__([3])]
    for edit in edits:
        d = s.getSideDofs(1, 1)
        edit(d)
        assert d.array is None
        assert np.array_equal(G._index_array(d), np.asarray(list(d), dtype=np.int64))
    d = s.getSideDofs(1, 0)
    assert type(d + [1]) is list and type(d[1:3]) is list and d.array is not None      # copies are plain lists


def test_product_multipatch_bookkeeping_matches_the_reference():
    """MultiBSpline (tIGAr/BSplines.py:651-700, 884-908) on ten seeded random configurations computed by the reference's class
    (tests/golden/make_golden_multipatch.py): knot vectors normalised to (0, 1), dof offsets, ncp, nel, getPatchSideDofs"""
    import json
    from tigar_amd import BSplines as B
    g = _load("golden_multipatch.npz")
    for m in json.loads(str(g["meta"])):
        name, npatch, degs = m["name"], m["npatch"], m["degrees"]
        patches = [B.BSpline(degs, [[float(v) for v in g["%s_p%d_kv%d_in" % (name, k, d)]] for d in range(2)]) for k in range(npatch)]
        mb = B.MultiBSpline(patches)
        assert list(mb.doffsets) == g[name + "_doffsets"].tolist()
        assert mb.getNcp() == int(g[name + "_ncp"]) and mb.nel == int(g[name + "_nel"])
        for k in range(npatch):
            for d in range(2):
                assert np.array_equal(np.asarray(patches[k].splines[d].knots, dtype=np.float64), g["%s_p%d_kv%d_norm" % (name, k, d)])
                for side in (0, 1):
                    for nl in (1, 2):
                        assert list(mb.getPatchSideDofs(k, d, side, nl)) == g["%s_p%d_side_%d_%d_%d" % (name, k, d, side, nl)].tolist()
