"""writeExtraction -> ExtractedSpline(dirname) round trip (tIGAr/common.py:435-502,
748-894): same extraction data, same solution through the general PtAP path."""
import os

import numpy as np
import pytest

from geom_util import quarter_annulus

pytestmark = pytest.mark.gpu


def test_extraction_directory_round_trip(tmp_path):
    import tigar_amd as t
    from tigar_amd import forms as F, NURBS, petscio
    kv, Pf = quarter_annulus(6)
    gen = t.EqualOrderSpline(1, NURBS.NURBSControlMesh([2, 2], [kv, kv], Pf))
    sp0 = gen.getScalarSpline(0)
    for direction in (0, 1):
        for side in (0, 1):
            gen.addZeroDofs(0, sp0.getSideDofs(direction, side))
    d = str(tmp_path / "extraction")
    gen.writeExtraction(d)
    assert sorted(os.listdir(d)) == ["extraction-data.npz", "extraction-info.txt", "extraction-mat-ctrl.dat",
                                     "extraction-mat.dat", "zero-dofs.dat"]
    info = open(os.path.join(d, "extraction-info.txt")).read().split("\n")
    assert info[:7] == ["2", "Lagrange", "1", "2", str(gen.getNcp(-1)), "2", str(gen.getNcp(0))]
    M = gen.M.to_scipy()
    Mf = petscio.read_mat(os.path.join(d, "extraction-mat.dat"))
    assert np.array_equal(M.indptr, Mf.indptr) and np.array_equal(M.indices, Mf.indices) and np.array_equal(M.data, Mf.data)
    assert petscio.read_is(os.path.join(d, "zero-dofs.dat")).tolist() == list(gen.zeroDofs)

    def solve(spline, geometry):
        solver = t.PETScKrylovSolver("cg", "jacobi")
        solver.parameters["relative_tolerance"] = 1e-13
        spline.setSolverOptions(linearSolver=solver)
        u = t.Function(spline.V)
        rhs = F.NodalLoadForm(lambda x: np.sin(x[:, 0]) + x[:, 1] ** 2, geometry)
        spline.solveLinearVariationalProblem(F.Equation(F.LaplaceForm(geometry=geometry), rhs), u)
        return u.vector().get_local()

    s_gen = t.ExtractedSpline(gen, 4)
    s_dir = t.ExtractedSpline(d, 4)
    assert s_dir.nsd == 2 and s_dir.nFields == 1 and s_dir.p == [2] and s_dir.p_control == 2
    assert s_dir.elementType == "Lagrange" and s_dir.V.dim() == s_gen.V.dim()
    for a, b in zip(s_gen.cpFuncs, s_dir.cpFuncs):
        assert np.array_equal(a.vector().get_local(), b.vector().get_local())
    assert list(s_dir.zeroDofs) == list(s_gen.zeroDofs)
    u1 = solve(s_gen, gen)
    u2 = solve(s_dir, s_dir)          # geometry from the files; M through the general hash PtAP
    assert np.max(np.abs(u1 - u2)) <= 1e-10 * np.max(np.abs(u1))
    # a directory without the data file (as written by the reference: HDF5) is reported, not guessed
    os.remove(os.path.join(d, "extraction-data.npz"))
    with pytest.raises(IOError):
        t.ExtractedSpline(d, 4)
