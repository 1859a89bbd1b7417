#!/usr/bin/env python3
"""
Golden fixture for the compatible (RT / N type) B-spline field construction, generated from the
REFERENCE's ``generateFieldsCompat`` (tIGAr/compatibleSplines.py:21-66) through the stub import
(in-container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_compat.py

golden_compat.npz holds, per case, the inputs (control-mesh degrees / knot vectors, type, degrees k',
periodicities) and the reference's outputs (per field: degrees and knot vectors, ncp).
"""
import os, sys, json
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_stub_import as R  # noqa: E402

B = R.import_reference()
import tIGAr.common as _TC  # noqa: E402
_TC.Constant = lambda v: v        # module-level DEFAULT_RT_PENALTY = Constant(1e1) (compatibleSplines.py:19)
import tIGAr.compatibleSplines as C  # noqa: E402


class _CM(object):
    def __init__(self, spline):
        self.s = spline

    def getScalarSpline(self):
        return self.s


def main():
    out, meta = {}, []
    cases = [
        ("rt2d", "RT", [1, 1], None, [2, 2], [B.uniformKnots(2, 0., 1., 4), B.uniformKnots(2, 0., 2., 3)]),
        ("n2d", "N", [1, 2], None, [2, 2], [B.uniformKnots(2, 0., 1., 3), B.uniformKnots(2, -1., 1., 5)]),
        ("rt3d", "RT", [1, 1, 1], None, [1, 1, 1], [B.uniformKnots(1, 0., 1., 2)] * 3),
        ("rt2d_per", "RT", [2, 1], [True, False], [2, 2], [B.uniformKnots(2, 0., 1., 5), B.uniformKnots(2, 0., 1., 4)]),
        ("n3d", "N", [2, 1, 1], None, [2, 2, 2], [B.uniformKnots(2, 0., 1., 2), B.uniformKnots(2, 0., 1., 3), B.uniformKnots(2, 0., 3., 2)]),
    ]
    # seeded random cases: type, dimension, degrees k' per direction, periodicities, control-mesh degree and knot vectors
    rng = np.random.default_rng(31)
    for i in range(24):
        d = int(rng.choice([2, 2, 3]))
        kind = str(rng.choice(["RT", "N"]))
        degs = [int(rng.integers(1, 4 if d == 2 else 3)) for _ in range(d)]
        per = [bool(rng.random() < 0.25) for _ in range(d)]
        per = per if any(per) else None
        cdeg = [int(rng.integers(1, 3))] * d
        ckv = [B.uniformKnots(cdeg[j], float(rng.choice([0., -1.])), float(rng.choice([1., 2.5])),
                              int(rng.integers(max(degs) + 2, 7 if d == 2 else 5))) for j in range(d)]
        cases.append(("rnd%02d" % i, kind, degs, per, cdeg, ckv))
    for name, kind, degs, per, cdeg, ckv in cases:
        cm = _CM(B.BSpline(cdeg, ckv))
        fields = C.generateFieldsCompat(cm, kind, degs, periodicities=per)
        meta.append({"name": name, "kind": kind, "degrees": degs, "periodicities": per, "cdeg": cdeg, "nfields": len(fields)})
        for j, kv in enumerate(ckv):
            out["%s_ckv%d" % (name, j)] = np.asarray(kv, dtype=np.float64)
        for i, f in enumerate(fields):
            out["%s_f%d_deg" % (name, i)] = np.asarray([s.p for s in f.splines], dtype=np.int64)
            out["%s_f%d_ncp" % (name, i)] = np.int64(f.getNcp())
            for j, s in enumerate(f.splines):
                out["%s_f%d_kv%d" % (name, i, j)] = np.asarray(s.knots, dtype=np.float64)
    out["meta"] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(HERE, "golden_compat.npz"), **out)
    print("wrote golden_compat.npz:", [m["name"] for m in meta])


if __name__ == "__main__":
    main()
