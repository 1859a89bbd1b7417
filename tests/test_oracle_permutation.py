"""The oracle's restatement of generatePermutation / applyPermutation (tIGAr/common.py:407-433, 1583-1665) on a case
small enough to do by hand."""
import numpy as np
import scipy.sparse as sp

from oracle import tigar_oracle as O


def test_majority_owner_tie_and_renaming_by_hand():
    # FE rows x dofs; row I lists the functions getNodesAndEvals returns at FE node I
    rows = [[0, 1], [0, 1, 2], [1, 2], [2, 3]]
    indptr = np.cumsum([0] + [len(r) for r in rows])
    support = sp.csr_matrix((np.zeros(indptr[-1]), np.concatenate(rows), indptr), shape=(4, 4))   # values may be 0.0
    fe_owner = np.array([0, 1, 1, 0])
    # dof 0: FE rows 0,1 -> owners 0,1, tie -> 0 | dof 1: rows 0,1,2 -> 0,1,1 -> 1 | dof 2: rows 1,2,3 -> 1,1,0 -> 1
    # dof 3: row 3 -> 0 ; ranks [0,1,1,0] -> stable argsort
    perm = O.generate_permutation(support, fe_owner)
    assert perm.tolist() == [0, 3, 1, 2]
    M = sp.csr_matrix(np.arange(1.0, 17.0).reshape(4, 4))
    Mp, zd = O.apply_permutation(M, [3, 1], perm)
    assert np.array_equal(Mp.toarray(), M.toarray()[:, [0, 3, 1, 2]])
    assert zd.tolist() == [1, 2]          # old dof 3 is new dof 1, old dof 1 is new dof 2
