"""CPU tests of the z-slab layout arithmetic (tigar_amd/dist.py) against the oracle's
matrices, and a world_size-2 gloo run of the distributed Krylov pattern (halo exchange +
scalar all-reduce) with the oracle standing in for the kernels."""
import os
import subprocess
import sys
import numpy as np
import pytest

from oracle import tigar_oracle as O
from tigar_amd.dist import ZSlabLayout, split_range

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(d, p, nel, drop=0):
    kv = O.uniform_knots(p, 0., 1., nel, False, drop)
    s = O.BSpline([p] * d, [kv] * d)
    Mo = O.generate_M_tensor(s)
    A, b, _, _ = O.poisson_fe_system(s, f1d=[lambda x: np.sin(np.pi * x)] * d)
    nodes = O.fe_nodes_1d(s.splines[-1], p)
    ncps = [q.getNcp() for q in s.splines]
    nfe = [len(O.fe_nodes_1d(q, p)) for q in s.splines]
    lay = ZSlabLayout(s.splines[-1].knots, p, nodes, p, int(np.prod(ncps[:-1])), int(np.prod(nfe[:-1])))
    return s, Mo, A, b, lay


def test_split_range():
    assert split_range(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert split_range(2, 4) == [(0, 1), (1, 2), (2, 2), (2, 2)]


@pytest.mark.parametrize("d,p,nel,world,drop", [(2, 2, 9, 3, 0), (2, 3, 7, 2, 0), (3, 2, 5, 2, 0), (2, 4, 6, 4, 0),
                                                 (1, 3, 12, 5, 0), (2, 3, 6, 3, 1)])
def test_slab_extents_cover_and_are_tight(d, p, nel, world, drop):
    s, Mo, A, b, lay = _setup(d, p, nel, drop)
    K = (Mo.T @ A @ Mo).tocsr()
    MT = Mo.T.tocsr()
    A = A.tocsr()
    owned = np.zeros(Mo.shape[0], dtype=int)
    for (k0, k1) in split_range(lay.ncp, world):
        if k1 == k0:
            continue
        S = lay.slab(k0, k1)
        g0, g1 = S["dofs"]
        cols = MT[g0:g1].indices
        a0, a1 = S["a_rows"]
        assert cols.min() >= a0 and cols.max() < a1               # rows of A needed by the slab
        assert cols.min() < a0 + lay.plane_fe * lay.q + lay.plane_fe and cols.max() >= a1 - lay.plane_fe * (lay.q + 1)
        ac = A[a0:a1].indices
        m0, m1 = S["m_rows"]
        assert ac.min() >= m0 and ac.max() < m1                   # rows of M needed
        assert ac.min() < m0 + lay.plane_fe and ac.max() >= m1 - lay.plane_fe     # tight to a plane
        kc = K[g0:g1].indices
        hl, hh = S["halo"]
        assert kc.min() >= g0 - hl and kc.max() < g1 + hh         # Krylov halo
        assert kc.min() < g0 - hl + lay.plane_dofs and kc.max() >= g1 + hh - lay.plane_dofs
        u0, u1 = S["u_rows"]
        owned[u0:u1] += 1
        if u1 > u0:
            mc = Mo[u0:u1].indices
            assert mc.min() >= g0 and mc.max() < g1 + hh          # prolongation needs only the upper halo
    assert np.all(owned == 1)                                       # FE rows partitioned exactly


def test_local_ptap_blocks_reproduce_global_K():
    """K rows of a slab computed from the three local row blocks == rows of the global K."""
    import scipy.sparse as sp
    s, Mo, A, b, lay = _setup(3, 2, 4)
    Kg = (Mo.T @ A @ Mo).tocsr()
    for (k0, k1) in split_range(lay.ncp, 3):
        S = lay.slab(k0, k1)
        g0, g1 = S["dofs"]
        a0, a1 = S["a_rows"]
        m0, m1 = S["m_rows"]
        MTl = Mo.T.tocsr()[g0:g1][:, a0:a1]
        Al = A.tocsr()[a0:a1][:, m0:m1]
        Ml = Mo[m0:m1]
        Kl = (MTl @ Al @ Ml).tocsr()
        assert abs(Kl - Kg[g0:g1]).max() < 1e-13 * abs(Kg).max()
        yl = MTl @ b[a0:a1]
        assert np.max(np.abs(yl - (Mo.T @ b)[g0:g1])) < 1e-13 * np.max(np.abs(b))


def test_gloo_world2_distributed_cg_pattern():
    """Two CPU processes (gloo): slab-local K rows, halo exchange of the direction vector with
    the z-neighbour, all-reduced dot products -- the communication pattern of tg_krylov_solve --
    must reproduce the serial solution."""
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "tests", "dist_cpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DIST_OK" in out.stdout


def test_socket_transport_world2_distributed_cg_pattern():
    """The same pattern carried by the product's own launcher and TCP transport
    (tigar_amd.launch.spawn_local + SocketTransport: what bench.py --gpus N and the host-staged device
    communicator use)."""
    sys.path.insert(0, ROOT)
    from tigar_amd.launch import spawn_local
    env = {"PYTHONPATH": ROOT + os.pathsep + os.environ.get("PYTHONPATH", "")}
    rc = spawn_local(2, [os.path.join(ROOT, "tests", "dist_cpu_worker.py"), "socket"], env_extra=env, port=29877)
    assert rc == 0


def _rdv_worker(rank, world, port, q):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    from tigar_amd.launch import HostRendezvous
    r = HostRendezvous(rank, world)
    data = r.broadcast_bytes(bytes(range(128)) if rank == 0 else None, 128)
    m = r.allreduce_max(float(rank) * 1.5)
    r.barrier()
    q.put((rank, data == bytes(range(128)), m))


def test_host_rendezvous_three_ranks():
    """The launcher-side TCP transport of tigar_amd/launch.py (RCCL id broadcast, barrier, max)."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, port = 3, 29741
    procs = [ctx.Process(target=_rdv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in reversed(procs):          # start rank 0 last: the others must retry connecting
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
    assert [r[0] for r in res] == [0, 1, 2]
    assert all(r[1] for r in res)
    assert all(r[2] == 3.0 for r in res)


def test_host_rendezvous_skips_a_busy_port():
    """MASTER_PORT + 17 already taken by an unrelated listener (which answers nothing): rank 0 binds
    the next candidate, the other rank finds it through the magic-word handshake."""
    import multiprocessing as mp
    import socket
    world, port = 2, 29963
    blocker = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    blocker.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    blocker.bind(("127.0.0.1", port + 17))
    blocker.listen(4)
    try:
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_rdv_worker, args=(r, world, port, q)) for r in range(world)]
        for p in procs:
            p.start()
        res = sorted(q.get(timeout=180) for _ in range(world))
        for p in procs:
            p.join(timeout=60)
    finally:
        blocker.close()
    assert [r[0] for r in res] == [0, 1] and all(r[1] for r in res)


def _framing_worker(rank, world, port, q):
    import os
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    from tigar_amd.launch import SocketTransport
    tr = SocketTransport(rank, world)
    peer = 1 - rank
    # a matched pair, both directions at once, different lengths
    recv = np.zeros(3 if rank == 0 else 5)
    tr.sendrecv(peer, np.arange(5.0) if rank == 0 else np.arange(3.0) + 10.0, recv, tag=7)
    ok = np.array_equal(recv, np.arange(3.0) + 10.0 if rank == 0 else np.arange(5.0))
    # calls that do not pair up: rank 0 expects 2 values, rank 1 sends 4 (and the other way round the tags differ)
    try:
        tr.sendrecv(peer, np.ones(4), np.zeros(2 if rank == 0 else 4), tag=8 + rank)
        raised = False
    except RuntimeError as e:
        raised = "out of step" in str(e)
    q.put((rank, ok, raised))


def test_socket_transport_messages_are_framed():
    """sendrecv was a raw byte stream -- two ranks whose exchanges did not pair up (different numbers of ghost
    updates per assembly) read each other's bytes as data or stalled for a minute.  Every message now carries a header
    (magic, tag, length) that the receiver checks."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_framing_worker, args=(r, 2, 29411, q)) for r in range(2)]
    for p in reversed(procs):
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True, True), (1, True, True)]


def test_solver_stand_ins_are_announced_with_a_warning():
    """dolfin's solver / preconditioner names that this library has no implementation of are honoured with a stand-in
    (tIGAr/common.py:1255-1258 passes whatever the user configured) -- and say so at construction, not only in the message of
    a failed solve; names that run as requested stay silent"""
    import warnings
    import pytest
    import tigar_amd as t
    with pytest.warns(UserWarning, match="Chebyshev polynomial preconditioner stands in"):
        s = t.PETScKrylovSolver("cg", "ilu")
    assert (s.method, s.preconditioner, s.preconditioner_requested) == ("cg", "chebyshev", "ilu")
    with pytest.warns(UserWarning, match="Jacobi stands in"):
        s = t.PETScKrylovSolver("gmres", "hypre_amg")
    assert s.preconditioner == "jacobi"
    with pytest.warns(UserWarning, match="'minres' requested: gmres runs in its place"):
        s = t.PETScKrylovSolver("minres", "jacobi")
    assert s.method == "gmres" and s.method_requested == "minres"
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        t.PETScKrylovSolver("cg", "jacobi")
        t.PETScKrylovSolver("gmres", "none")
        t.PETScKrylovSolver("default", "default")
        t.PETScKrylovSolver("cg", "chebyshev")


def test_chebyshev_breakdown_falls_back_to_jacobi_cg(monkeypatch):
    """status -2 of Chebyshev-CG (an eigenvalue estimate that missed the upper end of the spectrum) is not reported as a
    breakdown of the solve: Jacobi-CG runs instead, with a warning and a note in ``last``; a breakdown of that
    solve as well (NaN in K or b) stays status -2 and raises"""
    import pytest
    import tigar_amd as t
    from tigar_amd import common as tc
    calls = []

    def fake(A, b, x, method, pc, *a, **k):
        calls.append((method, pc))
        return (7, float("nan"), -2) if pc == "chebyshev" else (41, 1e-9, 0)

    monkeypatch.setattr(tc, "_as_device_csr", lambda a: a)
    monkeypatch.setattr(tc, "_as_device_vector", lambda v: v)
    monkeypatch.setattr(tc._dev, "krylov_solve", fake)
    s = t.PETScKrylovSolver("cg", "chebyshev")
    with pytest.warns(UserWarning, match="broke down"):
        its = s.solve(object(), object(), object())
    assert its == 41 and calls == [("cg", "chebyshev"), ("cg", "jacobi")]
    assert s.last["status"] == 0 and s.last["preconditioner"] == "jacobi" and s.last["preconditioner_requested"] == "chebyshev"
    assert s.last["fallback"] == {"preconditioner": "jacobi", "after_iterations": 7}
    monkeypatch.setattr(tc._dev, "krylov_solve", lambda *a, **k: (3, float("nan"), -2))
    with pytest.warns(UserWarning), pytest.raises(RuntimeError, match="breakdown"):
        s.solve(object(), object(), object())
    assert s.last["status"] == -2
