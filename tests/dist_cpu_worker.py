"""world_size-2 gloo worker (CPU): the distributed hot-path pattern with the oracle standing in
for the kernels.  Launched by tests/test_dist_layout.py through torch.distributed.run."""
import os
import sys
import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tigar_oracle as O                     # noqa: E402
from tigar_amd.dist import ZSlabLayout, split_range      # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    d, p, nel = 3, 2, 6
    s = O.BSpline([p] * d, [O.uniform_knots(p, 0., 1., nel)] * d)
    Mo = O.generate_M_tensor(s)
    A, b, _, _ = O.poisson_fe_system(s, f1d=[lambda x: np.sin(np.pi * x)] * d)
    zd = []
    for direction in range(d):
        for side in (0, 1):
            zd += s.getSideDofs(direction, side)
    ncps = [q.getNcp() for q in s.splines]
    nfe = [len(O.fe_nodes_1d(q, p)) for q in s.splines]
    lay = ZSlabLayout(s.splines[-1].knots, p, O.fe_nodes_1d(s.splines[-1], p), p,
                      int(np.prod(ncps[:-1])), int(np.prod(nfe[:-1])))
    k0, k1 = split_range(lay.ncp, world)[rank]
    S = lay.slab(k0, k1)
    g0, g1 = S["dofs"]
    a0, a1 = S["a_rows"]
    m0, m1 = S["m_rows"]
    hl, hh = S["halo"]
    # slab-local assembly from local row blocks only
    MTl = Mo.T.tocsr()[g0:g1][:, a0:a1]
    Kl = (MTl @ A.tocsr()[a0:a1][:, m0:m1] @ Mo[m0:m1]).tocsr()
    Kfull_rows = O.zero_rows_columns(O.sp.vstack([O.sp.csr_matrix((g0, Kl.shape[1])), Kl,
                                                  O.sp.csr_matrix((Kl.shape[1] - g1, Kl.shape[1]))]).tocsr(), zd)
    Kl = Kfull_rows[g0:g1]
    rhs = MTl @ b[a0:a1]
    zl = np.array([z - g0 for z in zd if g0 <= z < g1], dtype=int)
    rhs[zl] = 0.0
    n = g1 - g0
    Kext = Kl[:, g0 - hl:g1 + hh]                # columns of the extended local vector

    def halo_exchange(xloc):
        ext = np.zeros(hl + n + hh)
        ext[hl:hl + n] = xloc
        reqs = []
        # what the neighbours need from me = their halo sizes (symmetric stencil here)
        if rank > 0:
            reqs.append(dist.isend(torch.from_numpy(xloc[:hl_of(rank - 1, "hi")].copy()), rank - 1))
        if rank < world - 1:
            reqs.append(dist.isend(torch.from_numpy(xloc[n - hl_of(rank + 1, "lo"):].copy()), rank + 1))
        if rank > 0:
            buf = torch.zeros(hl, dtype=torch.float64)
            dist.recv(buf, rank - 1)
            ext[:hl] = buf.numpy()
        if rank < world - 1:
            buf = torch.zeros(hh, dtype=torch.float64)
            dist.recv(buf, rank + 1)
            ext[hl + n:] = buf.numpy()
        for r in reqs:
            r.wait()
        return ext

    def hl_of(r, which):
        kk0, kk1 = split_range(lay.ncp, world)[r]
        h = lay.slab(kk0, kk1)["halo"]
        return h[0] if which == "lo" else h[1]

    def allsum(*vals):
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t)
        return t.tolist()

    dinv = 1.0 / Kl.diagonal(k=0) if False else None
    diag = np.array([Kl[i, g0 + i] for i in range(n)])
    dinv = np.where(diag != 0, 1.0 / diag, 1.0)
    x = np.zeros(n)
    r = rhs.copy()
    z = dinv * r
    pvec = z.copy()
    rz, zz = allsum(float(r @ z), float(z @ z))
    tol = max(1e-10 * np.sqrt(zz), 1e-30)
    its = 0
    for its in range(1, 2000):
        Kp = Kext @ halo_exchange(pvec)
        (pKp,) = allsum(float(pvec @ Kp))
        alpha = rz / pKp
        x += alpha * pvec
        r -= alpha * Kp
        z = dinv * r
        rz_new, zz = allsum(float(r @ z), float(z @ z))
        if np.sqrt(zz) <= tol:
            break
        pvec = z + (rz_new / rz) * pvec
        rz = rz_new
    # serial reference
    Kg = O.extract_matrix(Mo, A, zd)
    Ug, ug = O.solve_linear_system(Mo, Kg, O.extract_vector(Mo, b, zd), "direct")
    err = np.linalg.norm(x - Ug[g0:g1]) / np.linalg.norm(Ug)
    # prolongation rows owned by this rank, with the upper halo of U
    u0, u1 = S["u_rows"]
    ext = halo_exchange(x)
    ul = Mo[u0:u1][:, g0 - hl:g1 + hh] @ ext
    err_u = np.linalg.norm(ul - ug[u0:u1]) / np.linalg.norm(ug)
    ok = err < 1e-8 and err_u < 1e-8 and its < 500
    flags = allsum(1.0 if ok else 0.0)[0]
    if rank == 0:
        print("its", its, "err", err, err_u)
        print("DIST_OK" if flags == world else "DIST_FAIL")
    dist.destroy_process_group()
    sys.exit(0 if flags == world else 1)


if __name__ == "__main__":
    main()
