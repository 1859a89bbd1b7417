"""PETSc binary viewer formats of the on-disk extraction data (SURVEY.md 8f-2;
tIGAr/common.py:435-502, 748-894).  Known-answer bytes are restated from PETSc's documented
MatLoad/ISLoad/VecLoad layouts (PETSc itself is absent: parity unpinned)."""
import struct

import numpy as np
import pytest
import scipy.sparse as sp

from tigar_amd import petscio


def test_mat_known_answer_bytes(tmp_path):
    A = sp.csr_matrix(np.array([[1.5, 0.0, -2.0], [0.0, 0.0, 0.0], [0.0, 4.25, 0.0]]))
    path = tmp_path / "m.dat"
    petscio.write_mat(str(path), A)
    expected = struct.pack(">4i", 1211216, 3, 3, 3) + struct.pack(">3i", 2, 0, 1) + struct.pack(">3i", 0, 2, 1) \
        + struct.pack(">3d", 1.5, -2.0, 4.25)
    assert path.read_bytes() == expected
    B = petscio.read_mat(str(path))
    assert (B != A).nnz == 0 and B.shape == (3, 3)
    assert B.indices.dtype == np.int32 and B.data.dtype == np.float64


def test_is_and_vec_known_answer_bytes(tmp_path):
    p = tmp_path / "z.dat"
    petscio.write_is(str(p), [5, 0, 5, 7])                      # duplicates survive (corner dofs)
    assert p.read_bytes() == struct.pack(">6i", 1211218, 4, 5, 0, 5, 7)
    assert petscio.read_is(str(p)).tolist() == [5, 0, 5, 7]
    v = tmp_path / "v.dat"
    petscio.write_vec(str(v), [0.5, -1.0])
    assert v.read_bytes() == struct.pack(">2i", 1211214, 2) + struct.pack(">2d", 0.5, -1.0)
    assert petscio.read_vec(str(v)).tolist() == [0.5, -1.0]


def test_round_trip_random_and_errors(tmp_path):
    rng = np.random.default_rng(0)
    A = sp.random(57, 33, density=0.1, random_state=3, format="csr")
    A.data = rng.standard_normal(A.nnz)
    path = str(tmp_path / "a.dat")
    petscio.write_mat(path, A.tocoo())                         # any scipy format in
    B = petscio.read_mat(path)
    assert np.array_equal(B.indptr, A.indptr) and np.array_equal(B.indices, A.indices) and np.array_equal(B.data, A.data)
    empty = sp.csr_matrix((4, 6))
    petscio.write_mat(path, empty)
    assert petscio.read_mat(path).nnz == 0 and petscio.read_mat(path).shape == (4, 6)
    with pytest.raises(petscio.PetscFormatError):
        petscio.read_is(path)                                  # wrong classid
    with open(path, "wb") as f:
        f.write(struct.pack(">4i", 1211216, 2, 2, 3) + struct.pack(">2i", 2, 1))
    with pytest.raises(petscio.PetscFormatError):
        petscio.read_mat(path)                                 # truncated
