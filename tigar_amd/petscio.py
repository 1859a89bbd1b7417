"""
PETSc binary viewer formats used by the reference's on-disk extraction data
(``writeExtraction`` / ``initFromFilesystem``, tIGAr/common.py:435-502, 748-894):

    extraction-mat.dat, extraction-mat-ctrl.dat   Mat   (viewer(self.M.mat()), :469-474)
    zero-dofs.dat                                 IS    (:487-491)

Layout [ext: PETSc MatLoad/ISLoad/VecLoad manual pages; PETSc is not available in this image, so the
byte format is restated from its documentation -- parity unpinned against PETSc itself]: all
integers big-endian int32 (default PetscInt), scalars big-endian float64;

    Mat: MAT_FILE_CLASSID 1211216 | rows | cols | nnz | nnz-per-row[rows] | col[nnz] | val[nnz]
    Vec: VEC_FILE_CLASSID 1211214 | n | val[n]
    IS : IS_FILE_CLASSID  1211218 | n | idx[n]

Host-side I/O (numpy); the matrices are moved to / from HBM by the caller.
"""
import numpy as np
import scipy.sparse as sp

MAT_FILE_CLASSID = 1211216
VEC_FILE_CLASSID = 1211214
IS_FILE_CLASSID = 1211218
_I = np.dtype(">i4")
_F = np.dtype(">f8")


class PetscFormatError(ValueError):
    pass


def write_mat(path, A):
    """scipy sparse -> PETSc binary AIJ (rows in order, column indices sorted within a row)."""
    A = sp.csr_matrix(A)
    A.sort_indices()
    if A.nnz >= 2 ** 31 or max(A.shape) >= 2 ** 31:
        raise PetscFormatError("matrix too large for 32-bit PetscInt binary format (nnz = %d)" % A.nnz)
    with open(path, "wb") as f:
        np.array([MAT_FILE_CLASSID, A.shape[0], A.shape[1], A.nnz], dtype=_I).tofile(f)
        np.diff(A.indptr).astype(_I).tofile(f)
        A.indices.astype(_I).tofile(f)
        A.data.astype(_F).tofile(f)


def read_mat(path):
    with open(path, "rb") as f:
        head = np.fromfile(f, dtype=_I, count=4)
        if head.size != 4 or head[0] != MAT_FILE_CLASSID:
            raise PetscFormatError("%s is not a PETSc binary Mat (classid %s)" % (path, head[:1]))
        m, n, nnz = int(head[1]), int(head[2]), int(head[3])
        if nnz < 0:
            raise PetscFormatError("%s: dense/special Mat storage (nnz = %d) is not supported" % (path, nnz))
        lens = np.fromfile(f, dtype=_I, count=m)
        col = np.fromfile(f, dtype=_I, count=nnz)
        val = np.fromfile(f, dtype=_F, count=nnz)
        if lens.size != m or col.size != nnz or val.size != nnz or int(lens.sum()) != nnz:
            raise PetscFormatError("%s is truncated or inconsistent" % path)
    indptr = np.concatenate([[0], np.cumsum(lens.astype(np.int64))])
    return sp.csr_matrix((val.astype(np.float64), col.astype(np.int32), indptr), shape=(m, n))


def write_is(path, idx):
    idx = np.asarray(idx, dtype=np.int64)
    with open(path, "wb") as f:
        np.array([IS_FILE_CLASSID, idx.size], dtype=_I).tofile(f)
        idx.astype(_I).tofile(f)


def read_is(path):
    with open(path, "rb") as f:
        head = np.fromfile(f, dtype=_I, count=2)
        if head.size != 2 or head[0] != IS_FILE_CLASSID:
            raise PetscFormatError("%s is not a PETSc binary IS" % path)
        idx = np.fromfile(f, dtype=_I, count=int(head[1]))
        if idx.size != int(head[1]):
            raise PetscFormatError("%s is truncated" % path)
    return idx.astype(np.int32)


def write_vec(path, v):
    v = np.asarray(v, dtype=np.float64)
    with open(path, "wb") as f:
        np.array([VEC_FILE_CLASSID, v.size], dtype=_I).tofile(f)
        v.astype(_F).tofile(f)


def read_vec(path):
    with open(path, "rb") as f:
        head = np.fromfile(f, dtype=_I, count=2)
        if head.size != 2 or head[0] != VEC_FILE_CLASSID:
            raise PetscFormatError("%s is not a PETSc binary Vec" % path)
        v = np.fromfile(f, dtype=_F, count=int(head[1]))
        if v.size != int(head[1]):
            raise PetscFormatError("%s is truncated" % path)
    return v.astype(np.float64)
