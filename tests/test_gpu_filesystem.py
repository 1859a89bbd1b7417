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
    assert sorted(os.listdir(d)) == ["extraction-data.h5", "extraction-info.txt", "extraction-mat-ctrl.dat",
                                     "extraction-mat.dat", "zero-dofs.dat"]
    # the HDF5 file: the groups dolfin's HDF5File writes (tIGAr/common.py:460-467), in this package's node numbering
    from tigar_amd import h5io
    with h5io.H5File(os.path.join(d, "extraction-data.h5"), "r") as h5:
        X, topo = h5.read_dataset("/mesh/coordinates"), h5.read_dataset("/mesh/topology")
        nel = 6
        assert X.shape == ((nel + 1) ** 2, 2) and topo.shape == (nel ** 2, 4)
        assert h5.read_attr("/mesh/topology", "celltype") == "quadrilateral"
        assert np.array_equal(h5.read_dataset("/mesh/cell_indices"), np.arange(nel ** 2))
        nodes = gen.V_control.grids[0].coordinates()
        for i in range(3):                                  # x, y and the weight of the NURBS geometry
            name = "/control%d" % i
            assert np.array_equal(h5.read_dataset(name + "/vector_0"), gen.cpFuncs[i].vector().get_local())
            assert h5.read_attr(name, "signature") == "FiniteElement('Q', quadrilateral, 2)"
            cd = h5.read_dataset(name + "/cell_dofs").reshape(nel ** 2, 9)
            assert np.array_equal(h5.read_dataset(name + "/x_cell_dofs"), 9 * np.arange(nel ** 2 + 1))
            for c in (0, 7, nel ** 2 - 1):                  # the nodes of a cell lie inside the cell
                lo, hi = X[topo[c]].min(0), X[topo[c]].max(0)
                assert (nodes[cd[c]] >= lo - 1e-14).all() and (nodes[cd[c]] <= hi + 1e-14).all()
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
    # a data file without this package's group (as the reference writes it: M in dolfin's dof numbering) is reported,
    # not guessed; so is a directory without the file
    with h5io.H5File(os.path.join(d, "extraction-data.h5"), "w") as h5:
        h5.write_dataset("/mesh/coordinates", X)
    with pytest.raises(IOError, match="not written by this package"):
        t.ExtractedSpline(d, 4)
    os.remove(os.path.join(d, "extraction-data.h5"))
    with pytest.raises(IOError):
        t.ExtractedSpline(d, 4)


def test_round_trip_of_generators_on_other_node_sets(tmp_path):
    """writeExtraction / ExtractedSpline(dirname) for the node sets that are not one tensor grid: the Bezier-element
    mesh of a Rhino T-spline and the multi-patch mesh of a MultiBSpline (the reference writes both through the same
    generic code, tIGAr/common.py:435-502)."""
    import tigar_amd as t
    from tigar_amd import BSplines as B
    from tigar_amd.RhinoTSplines import RhinoTSplineControlMesh
    fname = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tspline_bicubic_patch.iga")
    gens = [t.EqualOrderSpline(1, RhinoTSplineControlMesh(fname))]

    patches = [B.BSpline([2, 2], [B.uniformKnots(2, 0., 3., 3), B.uniformKnots(2, 0., 1., 2)]),
               B.BSpline([2, 2], [B.uniformKnots(2, -1., 1., 2), B.uniformKnots(2, 0., 2., 3)])]
    mb = B.MultiBSpline(patches)

    class CM(t.AbstractControlMesh):
        def getScalarSpline(self):
            return mb

        def getNsd(self):
            return 2

        def getHomogeneousCoordinate(self, node, direction):
            if direction == 2:
                return 1.0
            patch = 0 if node < mb.doffsets[1] else 1
            s = patches[patch]
            local = node - mb.doffsets[patch]
            m0 = s.splines[0].getNcp()
            idx = (local % m0, local // m0)
            return s.splines[direction].greville(idx[direction]) + (2.0 * patch if direction == 0 else 0.0)
    gens.append(t.EqualOrderSpline(1, CM()))

    for k, gen in enumerate(gens):
        gen.addZeroDofs(0, [0, 3])
        d = str(tmp_path / ("extraction%d" % k))
        gen.writeExtraction(d)
        s_dir = t.ExtractedSpline(d, 4)
        s_gen = t.ExtractedSpline(gen, 4)
        assert type(s_dir.V.grids[0]) is type(s_gen.V.grids[0]) and s_dir.V.dim() == s_gen.V.dim()
        assert np.array_equal(s_dir.V.grids[0].coordinates(), s_gen.V.grids[0].coordinates())
        Ma, Mb = s_gen.M.to_scipy(), s_dir.M.to_scipy()
        assert np.array_equal(Ma.indptr, Mb.indptr) and np.array_equal(Ma.indices, Mb.indices) and np.array_equal(Ma.data, Mb.data)
        for a, b in zip(s_gen.cpFuncs, s_dir.cpFuncs):
            assert np.array_equal(a.vector().get_local(), b.vector().get_local())
        assert list(s_dir.zeroDofs) == [0, 3]


def _petsc_mat_bytes(A):
    """PETSc's MatLoad layout (big-endian: classid 1211216, rows, cols, nnz | row lengths | column indices | values),
    encoded HERE with struct.pack -- independently of tigar_amd.petscio, as tests/test_petscio.py restates it"""
    import struct
    import scipy.sparse as sp
    A = sp.csr_matrix(A)
    A.sort_indices()
    lens = np.diff(A.indptr)
    return (struct.pack(">4i", 1211216, A.shape[0], A.shape[1], A.nnz) + struct.pack(">%di" % A.shape[0], *lens)
            + struct.pack(">%di" % A.nnz, *A.indices) + struct.pack(">%dd" % A.nnz, *A.data))


def test_directory_with_independently_encoded_petsc_files_against_the_oracle(tmp_path):
    """Known answer instead of a round trip: the PETSc binary files of the directory are byte strings
    produced by this test's own struct.pack encoder from the ORACLE's extraction matrices and zero dofs (the product
    wrote only the node-set description); ExtractedSpline(dirname) reads them and M^T A M, M^T b and the solution are
    compared with the oracle's (tIGAr/common.py:748-894 -> 1176-1204)."""
    import struct
    import tigar_amd as t
    from tigar_amd import BSplines as B
    from oracle import tigar_oracle as O
    d, p, nel = 2, 2, 8
    kv = [B.uniformKnots(p, 0., 1., nel)] * d
    gen = t.EqualOrderSpline(1, B.ExplicitBSplineControlMesh([p] * d, kv))
    dirname = str(tmp_path / "extraction")
    gen.writeExtraction(dirname)                      # (node sets, control functions, info file)
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    Mo = O.generate_M_tensor(s)
    zd = []
    for direction in range(d):
        for side in (0, 1):
            zd += s.getSideDofs(direction, side)
    with open(os.path.join(dirname, "extraction-mat.dat"), "wb") as f:
        f.write(_petsc_mat_bytes(Mo))
    with open(os.path.join(dirname, "extraction-mat-ctrl.dat"), "wb") as f:
        f.write(_petsc_mat_bytes(Mo))
    with open(os.path.join(dirname, "zero-dofs.dat"), "wb") as f:
        f.write(struct.pack(">%di" % (2 + len(zd)), 1211218, len(zd), *zd))
    spline = t.ExtractedSpline(dirname, 2 * p)
    assert spline.zeroDofs.tolist() == zd                       # duplicates (corners) kept as written
    fn = lambda x: np.sin(np.pi * x)
    A, b, _, _ = O.poisson_fe_system(s, f1d=[fn] * d)
    K = spline.extractMatrix(A).to_scipy().tocsr()
    K.sort_indices()
    Ko = O.extract_matrix(Mo, A, zd).tocsr()
    Ko.sort_indices()
    assert np.array_equal(K.indptr, Ko.indptr) and np.array_equal(K.indices, Ko.indices)
    assert abs(K - Ko).max() <= 1e-12 * abs(Ko).max()
    rhs = spline.extractVector(b)
    assert np.max(np.abs(rhs.get_local() - O.extract_vector(Mo, b, zd))) <= 1e-13 * np.max(np.abs(b))
    solver = t.PETScKrylovSolver("cg", "jacobi")
    solver.parameters["relative_tolerance"] = 1e-12
    spline.setSolverOptions(linearSolver=solver)
    u = t.Function(spline.V)
    U = spline.solveLinearSystem(spline.extractMatrix(A), rhs, u)
    Uo, uo = O.solve_linear_system(Mo, Ko, O.extract_vector(Mo, b, zd), "direct")
    assert np.max(np.abs(U.get_local() - Uo)) <= 1e-9 * np.max(np.abs(Uo))
    assert np.max(np.abs(u.vector().get_local() - uo)) <= 1e-9 * np.max(np.abs(uo))
