"""world_size-2 worker (CPU): the distributed hot-path pattern -- z-slab layout, neighbour halo exchange,
ONE fused 3-scalar reduction per iteration of the single-reduction CG (the recurrence of csrc/tg_krylov.hip),
prolongation with halo -- with the oracle standing in for the kernels and the PRODUCT's host-side message
layer carrying the exchanges: ``tigar_amd.launch.SocketTransport`` (launched by the product's ``spawn_local``)
or torch.distributed/gloo adapted to the same ``Transport`` interface (launched by torch.distributed.run).
The device half of the same exchanges (tg_comm_* in csrc/tg_dist.hip) is covered by the 2-process -m gpu test."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import tigar_oracle as O                     # noqa: E402
from tigar_amd.dist import ZSlabLayout, split_range      # noqa: E402
from tigar_amd.launch import Transport, SocketTransport  # noqa: E402


class GlooTransport(Transport):
    """torch.distributed (gloo) behind the product's Transport interface"""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        dist.init_process_group("gloo")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allreduce_sum(self, a):
        t = self.torch.from_numpy(a)
        self.dist.all_reduce(t)
        return a

    def allreduce_max(self, value):
        t = self.torch.tensor([value], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def barrier(self):
        self.dist.barrier()

    def sendrecv(self, peer, send, recv, tag=0):
        reqs = []
        if send is not None and len(send):
            reqs.append(self.dist.isend(self.torch.from_numpy(np.ascontiguousarray(send)), peer))
        if recv is not None and len(recv):
            buf = self.torch.zeros(len(recv), dtype=self.torch.float64)
            self.dist.recv(buf, peer)
            recv[...] = buf.numpy()
        for r in reqs:
            r.wait()

    def close(self):
        self.dist.destroy_process_group()


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "gloo"
    tr = GlooTransport() if kind == "gloo" else SocketTransport()
    rank, world = tr.rank, tr.world
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

    def hl_of(r, which):
        kk0, kk1 = split_range(lay.ncp, world)[r]
        h = lay.slab(kk0, kk1)["halo"]
        return h[0] if which == "lo" else h[1]

    # what the neighbours need from this rank = their halo sizes (tg_comm_set_slab learns the same numbers)
    send_lo = hl_of(rank - 1, "hi") if rank > 0 else 0
    send_hi = hl_of(rank + 1, "lo") if rank < world - 1 else 0

    def halo_exchange(xloc):
        """the order of tg_comm_halo_exchange's host-staged branch: lower neighbour, then upper"""
        ext = np.zeros(hl + n + hh)
        ext[hl:hl + n] = xloc
        if rank > 0:
            tr.sendrecv(rank - 1, xloc[:send_lo].copy(), ext[:hl])
        if rank < world - 1:
            tr.sendrecv(rank + 1, xloc[n - send_hi:].copy(), ext[hl + n:])
        return ext

    def allsum(*vals):
        a = np.array(vals, dtype=np.float64)
        tr.allreduce_sum(a)
        return a.tolist()

    diag = np.array([Kl[i, g0 + i] for i in range(n)])
    dinv = np.where(diag != 0, 1.0 / diag, 1.0)
    # single-reduction CG (tg_cg in csrc/tg_krylov.hip): one fused reduction of (gamma, delta, nu) per iteration
    x = np.zeros(n)
    r = rhs.copy()
    u = dinv * r
    w = Kext @ halo_exchange(u)
    gamma, delta, nu = allsum(float(r @ u), float(w @ u), float(u @ u))
    tol2 = max(1e-10 * np.sqrt(nu), 1e-30) ** 2
    pvec, svec = np.zeros(n), np.zeros(n)
    gamma_prev = alpha_prev = None
    its, nred = 0, 1
    for its in range(1, 2000):
        if its == 1:
            beta, alpha = 0.0, gamma / delta
        else:
            beta = gamma / gamma_prev
            alpha = gamma / (delta - beta * gamma / alpha_prev)
        gamma_prev, alpha_prev = gamma, alpha
        pvec = u + beta * pvec
        svec = w + beta * svec
        x += alpha * pvec
        r -= alpha * svec
        u = dinv * r
        w = Kext @ halo_exchange(u)
        gamma, delta, nu = allsum(float(r @ u), float(w @ u), float(u @ u))
        nred += 1
        if not nu > tol2:
            break
    # serial reference
    Kg = O.extract_matrix(Mo, A, zd)
    Ug, ug = O.solve_linear_system(Mo, Kg, O.extract_vector(Mo, b, zd), "direct")
    err = np.linalg.norm(x - Ug[g0:g1]) / np.linalg.norm(Ug)
    # the textbook recurrence takes the same number of iterations (+-1) on this system
    _, its_ref, _ = O.cg_jacobi(Kg, O.extract_vector(Mo, b, zd), rtol=1e-10)
    # prolongation rows owned by this rank, with the upper halo of U
    u0, u1 = S["u_rows"]
    ext = halo_exchange(x)
    ul = Mo[u0:u1][:, g0 - hl:g1 + hh] @ ext
    err_u = np.linalg.norm(ul - ug[u0:u1]) / np.linalg.norm(ug)
    # a large exchange in both directions at once must not block (socket buffers are smaller than this)
    big = np.full(1 << 19, float(rank))
    got = np.empty(1 << 19)
    tr.sendrecv(1 - rank if world == 2 else (rank + 1 if rank % 2 == 0 else rank - 1), big, got)
    ok = err < 1e-8 and err_u < 1e-8 and abs(its - its_ref) <= 1 and nred == its + 1 \
        and bool(np.all(got == float(1 - rank))) and tr.allreduce_max(float(rank)) == float(world - 1)
    flags = allsum(1.0 if ok else 0.0)[0]
    if rank == 0:
        print("its", its, "ref", its_ref, "err", err, err_u, "reductions", nred)
        print("DIST_OK" if flags == world else "DIST_FAIL")
    tr.barrier()
    tr.close()
    sys.exit(0 if flags == world else 1)


if __name__ == "__main__":
    main()
