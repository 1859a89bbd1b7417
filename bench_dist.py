"""
z-slab driver of bench.py: one process per GPU (RCCL over xGMI for the Krylov halo exchange and
the dot-product all-reduces), and -- also on a single GPU -- streaming of workloads whose M and
A do not fit in HBM through the device in sub-slabs (only K stays resident).
"""
import os
import sys
import time

import numpy as np


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class HostRendezvous(object):
    """Minimal TCP rendezvous between the ranks of one node (rank 0 listens on MASTER_ADDR at
    MASTER_PORT + 17, or the next free candidate): hands out the RCCL unique id and provides the host-side barrier / max of the
    bench contract.  Plain sockets: nothing but the Python standard library in the launcher path
    (importing torch here would also pull a second RCCL / HIP runtime into the process)."""

    def __init__(self, rank, world):
        import socket
        import struct
        self.rank, self.world = rank, world
        self.struct = struct
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(os.environ.get("MASTER_PORT", "29500"))
        # candidate ports (the first free one is used; a port taken by something else is skipped: both
        # sides check a magic word that is derived from MASTER_PORT)
        ports = [base + 17 + 101 * k for k in range(8)]
        magic = struct.pack("q", 0x7469676172000000 ^ base)
        self.peers = []
        if world == 1:
            return
        if rank == 0:
            srv = None
            for port in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, port))
                    srv.listen(world)
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError("no free rendezvous port among %s" % ports)
            conns = {}
            while len(conns) < world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    c.settimeout(10.0)
                    hello = self._recv(c, 12)
                    c.settimeout(None)
                except (OSError, ConnectionError):
                    c.close()
                    continue
                if hello[:8] != magic:
                    c.close()
                    continue
                c.sendall(magic)
                conns[struct.unpack("i", hello[8:])[0]] = c
            self.peers = [conns[r] for r in sorted(conns)]
            srv.close()
        else:
            deadline = time.time() + 180.0
            c = None
            while c is None:
                for port in ports:
                    try:
                        cand = socket.create_connection((addr, port), timeout=5.0)
                        cand.sendall(magic + struct.pack("i", rank))
                        if self._recv(cand, 8) == magic:
                            c = cand
                            break
                        cand.close()
                    except (OSError, ConnectionError):
                        pass
                if c is None:
                    if time.time() > deadline:
                        raise OSError("rendezvous with rank 0 failed on ports %s" % ports)
                    time.sleep(0.2)
            c.settimeout(None)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self.peers = [c]

    @staticmethod
    def _recv(c, n):
        buf = b""
        while len(buf) < n:
            part = c.recv(n - len(buf))
            if not part:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += part
        return buf

    def broadcast_bytes(self, data, n):
        if self.world == 1:
            return data
        if self.rank == 0:
            for c in self.peers:
                c.sendall(data)
            return data
        return self._recv(self.peers[0], n)

    def allreduce_max(self, value):
        """max over ranks of a host double (also a barrier)"""
        if self.world == 1:
            return value
        st = self.struct
        if self.rank == 0:
            vals = [value] + [st.unpack("d", self._recv(c, 8))[0] for c in self.peers]
            m = max(vals)
            for c in self.peers:
                c.sendall(st.pack("d", m))
            return m
        self.peers[0].sendall(st.pack("d", value))
        return st.unpack("d", self._recv(self.peers[0], 8))[0]

    def barrier(self):
        self.allreduce_max(0.0)


def _bootstrap_comm(rank, world):
    """RCCL unique id from rank 0 to everybody (host-side plumbing only)."""
    from tigar_amd import device as dev
    rdv = HostRendezvous(rank, world)
    if world == 1:
        return None, rdv
    uid = dev.Comm.unique_id() if rank == 0 else None
    uid = rdv.broadcast_bytes(uid, 128)
    return dev.Comm(uid, rank, world), rdv


def pick_sub_planes(d, p, nel, planes_mine, free_bytes):
    """dof planes per sub-slab so that one slab of A and the PtAP temporaries use at most about half
    of the free HBM (K needs the rest)."""
    nfe1 = nel * p + 1
    plane_fe = nfe1 ** (d - 1)
    nnzA_plane = plane_fe * ((2 * p + 1) ** d) * 0.55 * 12.0 * 1.0     # bytes per FE plane, generous
    nnzM_plane = plane_fe * ((p + 1) ** d) * 12.0
    per_dof_plane = p * (nnzA_plane + 2.2 * nnzM_plane) + (nel + p) ** (d - 1) * ((2 * p + 1) ** d) * 12.0 * 2
    fixed = (2 * p * p + 2) * (nnzA_plane + 2.2 * nnzM_plane)
    # (half of the free HBM: measured at 256^3 p=3 with the 3/4-of-HBM allocator pool, per step:
    # 5 planes 1.59 s of input+PtAP, 8: 1.53 s, 12: 1.44 s, 16: 1.41 s with 75 GB still free at the end
    # of a step, 20: allocation failures and pool trimming start, 24: 5.6 s)
    # (the sliced copy of K's values that lives during the Krylov solve, 47 GB at 256^3 p=3, is stored in
    # idle blocks of the allocator's pool -- the PtAP temporaries sized here -- so it needs no budget of
    # its own; when it still asked the driver for one block, 14 planes thrashed the allocator (3.5 s per
    # step) and 7 were the safe choice)
    budget = 0.5 * free_bytes - fixed
    n = int(max(1, min(planes_mine, budget // per_dof_plane)))
    return n


def run_distributed(args, d, p, nel, rank, world):
    from tigar_amd import device as dev
    from tigar_amd.common import TensorFunctionSpace
    from tigar_amd.BSplines import ExplicitBSplineControlMesh, uniformKnots
    from tigar_amd.forms import LaplaceForm, SeparableLoadForm
    from tigar_amd.dist import SlabHotPath

    comm, rdv = _bootstrap_comm(rank, world)
    if rank == 0:
        log("[bench] device:", dev.device_info(), "world", world)
    kvecs = [uniformKnots(p, 0.0, 1.0, nel) for _ in range(d)]
    controlMesh = ExplicitBSplineControlMesh([p] * d, kvecs)
    basis = controlMesh.getScalarSpline()
    grid = basis.generateMesh(degree=p)
    V = TensorFunctionSpace([grid], "Lagrange")
    lap = LaplaceForm()
    f1 = lambda x: np.sin(np.pi * x)
    load = SeparableLoadForm([f1] * d, scale=d * np.pi ** 2)
    zero_dofs = []
    for direction in range(d):
        for side in (0, 1):
            zero_dofs += basis.getSideDofs(direction, side)
    zero_dofs = np.asarray(zero_dofs, dtype=np.int32)

    free_b, total_b = dev.mem_info()
    probe = SlabHotPath(basis, grid, rank, world, None)
    planes_mine = probe.k1 - probe.k0
    sub = args.sub_planes or pick_sub_planes(d, p, nel, planes_mine, free_b)
    del probe
    path = SlabHotPath(basis, grid, rank, world, comm, sub_planes=sub)
    if rank == 0:
        log("[bench] z-slabs: %d dof planes on rank 0 in sub-slabs of %d; free HBM %.0f GB"
            % (planes_mine, sub, free_b / 2 ** 30))

    def barrier():
        dev.sync()
        rdv.barrier()

    stages = {}
    state = {}

    def step(record):
        timers = {}
        K, rhs = path.assemble(lambda r0, r1: lap.assemble_matrix(V, r0, r1),
                               lambda r0, r1: load.assemble_vector(V, r0, r1), zero_dofs, 1.0, timers)
        t0 = time.perf_counter()
        U, its, res, status = path.solve(K, rhs, "cg", "jacobi", rtol=args.rtol)
        dev.sync()
        timers["solve"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        u = path.prolong(U)
        dev.sync()
        timers["prolong"] = time.perf_counter() - t0
        if status < 0:
            raise RuntimeError("CG did not converge: status %d after %d iterations" % (status, its))
        if record:
            for k, v in timers.items():
                stages.setdefault(k, []).append(v)
        state.update(K=K, U=U, u=u, its=its)
        if os.environ.get("TIGAR_TRACE"):
            pb, nb, nl = dev.pool_stats()
            fr, tot = dev.mem_info()
            log("[bench] pool: %.1f GB free in %d blocks, %d blocks live; device free %.1f GB" % (pb / 2 ** 30, nb, nl, fr / 2 ** 30))

    for _ in range(args.warmup):
        state.clear()        # (the previous step's K must not stay alive beside the one being assembled)
        step(False)
    dev.prof_reset()
    barrier()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        state.clear()
        step(True)
    barrier()
    elapsed = time.perf_counter() - t_start
    elapsed = rdv.allreduce_max(elapsed)
    K = state["K"]
    nnzK_local = K.nnz
    ncp_local = K.shape[0]
    nnzK = nnzK_local
    if comm is not None:
        nnzK = int(round(comm.allreduce_sum([float(nnzK_local)])[0]))
    spmv_ms, spmv_n = dev.prof_get(0)
    # which product kernel the solver used: the sliced copy (tg_sell.hip) if K has the structure
    sell_classes, sell_padded = K.spmv_sell(True)
    K.spmv_sell(False)

    if args.check and rank == 0:
        # manufactured solution at the FE nodes this rank owns
        r0, r1 = path.mine["u_rows"]
        n0 = grid.shape()
        uh = state["u"].get_local()
        if True:
            # every node for small problems, every 97th for large ones (cfg3: 4.7 M of 455 M nodes)
            idx = np.arange(r0, r1) if uh.size <= 40e6 else np.arange(r0, r1, 97)
            uh = uh if uh.size <= 40e6 else uh[idx - r0]
            exact = np.ones(idx.size)
            stride = 1
            for k in range(d):
                exact *= np.sin(np.pi * grid.axes[k][(idx // stride) % n0[k]])
                stride *= n0[k]
            log("[bench] max nodal error vs manufactured solution (rank 0 rows): %.3e" % np.max(np.abs(uh - exact)))
    mean_stages = {k: float(np.mean(v)) for k, v in stages.items()}
    if rank == 0:
        log("[bench] stages (mean s):", {k: round(v, 5) for k, v in mean_stages.items()}, "CG iterations:",
            state["its"], "nnz(K) global:", nnzK)
    t_input = mean_stages.pop("input", 0.0)
    return {"ncp": basis.getNcp(), "nnzK": nnzK, "nnzK_local": nnzK_local, "ncp_local": ncp_local,
            "elapsed": elapsed, "spmv_ms_total": spmv_ms, "spmv_count": spmv_n, "iterations": state["its"],
            "stages": mean_stages, "t_input": t_input, "t_input_in_timed_region": True, "sub_planes": sub,
            "sell_classes": sell_classes, "sell_padded": sell_padded}
