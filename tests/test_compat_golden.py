"""generateFieldsCompat (tIGAr/compatibleSplines.py:21-66) against the reference's own outputs
(tests/golden/golden_compat.npz, generated through the stub import)."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class _CM(object):
    def __init__(self, spline):
        self.s = spline

    def getScalarSpline(self):
        return self.s


def test_fields_match_reference():
    from tigar_amd import BSplines as B
    from tigar_amd.compatibleSplines import generateFieldsCompat
    g = np.load(os.path.join(HERE, "golden", "golden_compat.npz"))
    meta = json.loads(str(g["meta"]))
    assert len(meta) == 29            # five hand-picked + 24 seeded random cases
    for m in meta:
        name = m["name"]
        ckv = [g["%s_ckv%d" % (name, j)] for j in range(len(m["cdeg"]))]
        cm = _CM(B.BSpline(m["cdeg"], ckv))
        fields = generateFieldsCompat(cm, m["kind"], m["degrees"], periodicities=m["periodicities"])
        assert len(fields) == m["nfields"]
        for i, f in enumerate(fields):
            assert [s.p for s in f.splines] == g["%s_f%d_deg" % (name, i)].tolist()
            assert f.getNcp() == int(g["%s_f%d_ncp" % (name, i)])
            for j, s in enumerate(f.splines):
                assert np.array_equal(np.asarray(s.knots, dtype=np.float64), g["%s_f%d_kv%d" % (name, i, j)])
