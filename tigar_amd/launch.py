"""
Process-group plumbing of the multi-GPU path: one process per GPU (RANK / WORLD_SIZE / LOCAL_RANK /
MASTER_ADDR / MASTER_PORT as set by ``python -m torch.distributed.run``, or by ``spawn_local``).

The reference inherits its process group from MPI through dolfin (``MPI.comm_world``,
tIGAr/common.py:35-39) and every generator / spline takes a communicator argument
(tIGAr/common.py:139-153, 676-677).  Here a communicator is a ``Transport`` (host side: rendezvous,
barrier, small host reductions) plus, on top of it, a device communicator of libtigar_hip.so:

* RCCL (``device.Comm``: ncclSend/Recv halo exchange + ncclAllReduce over xGMI) -- the product path on a
  multi-GPU node; the transport only carries the 128-byte RCCL id to the ranks;
* IPC (``device.IpcComm``): halo planes pushed by a kernel into the neighbour's device mailbox (HIP IPC), flags
  and the slots of the small all-reduce in a shared-memory file, waits inside the kernels -- enqueue-only like
  RCCL; used where RCCL cannot form a group (several ranks sharing one GPU, e.g. on a 1-GPU box) and as the
  fallback when RCCL fails to initialise;
* host-staged (``device.HostComm``): the same solver code with its exchanges staged through pinned host
  memory and carried by the transport (one host wait per exchange) -- the last resort, and the reference the
  other two are compared with bit for bit.

Only the Python standard library is needed here (no torch import in the launch path: importing it would
pull a second HIP / RCCL runtime into the process).  ``tests/`` adapt ``torch.distributed`` (gloo) to the
same ``Transport`` interface.
"""
import os
import select
import socket
import struct
import subprocess
import sys
import time

import numpy as np


class Transport(object):
    """Host-side message layer between the ranks.  Interface: ``rank``, ``world``,
    ``broadcast_bytes(data, n)`` (from rank 0), ``allreduce_sum(array)`` (float64, in place),
    ``allreduce_max(value)``, ``sendrecv(peer, send, recv)`` (float64 arrays; either may be empty),
    ``barrier()``."""

    rank, world = 0, 1

    def broadcast_bytes(self, data, n):
        return data

    def allreduce_sum(self, a):
        return a

    def allreduce_max(self, value):
        return value

    def sendrecv(self, peer, send, recv, tag=0):
        """exchange with rank ``peer``: ``send`` (float64 array or None) goes out, ``recv`` (writable float64 array or None) is
        filled; ``tag`` names the exchange (transports that frame their messages check it on arrival, others ignore it)"""
        raise RuntimeError("sendrecv on a single-rank transport")

    def barrier(self):
        pass

    def close(self):
        pass


def _recv_exact(c, n):
    buf = bytearray()
    while len(buf) < n:
        part = c.recv(min(1 << 20, n - len(buf)))
        if not part:
            raise ConnectionError("transport peer closed the connection")
        buf += part
    return bytes(buf)


class SocketTransport(Transport):
    """TCP transport on one node.  Rank 0 listens on MASTER_ADDR at a port derived from MASTER_PORT (the
    first free one of a few candidates; both sides check a magic word, so a port held by something else
    is skipped); every rank also opens an ephemeral listening port, the table of which travels through
    rank 0, and connects to its upper neighbour -- halo data goes directly between neighbours."""

    def __init__(self, rank=None, world=None, addr=None, base_port=None, timeout=180.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.hub = []          # rank 0: sockets to ranks 1..world-1 (in rank order); others: [socket to rank 0]
        self.nbr = {}          # rank -> socket, for rank-1 and rank+1
        if self.world == 1:
            return
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        base = int(os.environ.get("MASTER_PORT", "29500")) if base_port is None else int(base_port)
        ports = [base + 17 + 101 * k for k in range(8)]
        magic = struct.pack("q", 0x7469676172000000 ^ base)
        # own listening socket for the lower neighbour's connection
        lst = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        lst.bind((addr, 0))
        lst.listen(2)
        my_port = lst.getsockname()[1]
        if self.rank == 0:
            srv = None
            for port in ports:
                try:
                    srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    srv.bind((addr, port))
                    srv.listen(self.world)
                    break
                except OSError:
                    srv.close()
                    srv = None
            if srv is None:
                raise OSError("no free rendezvous port among %s" % ports)
            conns = {}
            srv.settimeout(timeout)
            while len(conns) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    c.settimeout(10.0)
                    hello = _recv_exact(c, 12)
                    c.settimeout(None)
                except (OSError, ConnectionError):
                    c.close()
                    continue
                if hello[:8] != magic:
                    c.close()
                    continue
                c.sendall(magic)
                conns[struct.unpack("i", hello[8:])[0]] = c
            self.hub = [conns[r] for r in sorted(conns)]
            srv.close()
        else:
            deadline = time.time() + timeout
            c = None
            while c is None:
                for port in ports:
                    try:
                        cand = socket.create_connection((addr, port), timeout=5.0)
                        cand.sendall(magic + struct.pack("i", self.rank))
                        if _recv_exact(cand, 8) == magic:
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
            self.hub = [c]
        # neighbour links: port table through the hub, then rank r connects to rank r+1
        table = self._gather_ints(my_port)
        if self.rank + 1 < self.world:
            up = socket.create_connection((addr, table[self.rank + 1]), timeout=timeout)
            up.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            up.settimeout(None)
            self.nbr[self.rank + 1] = up
        if self.rank > 0:
            lst.settimeout(timeout)
            dn, _ = lst.accept()
            dn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            dn.settimeout(None)
            self.nbr[self.rank - 1] = dn
        lst.close()

    def _gather_ints(self, value):
        """all-gather of one int per rank through the hub"""
        if self.rank == 0:
            vals = [int(value)] + [struct.unpack("q", _recv_exact(c, 8))[0] for c in self.hub]
            blob = struct.pack("%dq" % self.world, *vals)
            for c in self.hub:
                c.sendall(blob)
            return vals
        self.hub[0].sendall(struct.pack("q", int(value)))
        return list(struct.unpack("%dq" % self.world, _recv_exact(self.hub[0], 8 * self.world)))

    def broadcast_bytes(self, data, n):
        if self.world == 1:
            return data
        if self.rank == 0:
            for c in self.hub:
                c.sendall(data)
            return data
        return _recv_exact(self.hub[0], n)

    def allreduce_sum(self, a):
        """in-place sum over ranks of a float64 array; summed in rank order on rank 0, so every rank gets
        the same bits"""
        if self.world == 1:
            return a
        a = np.ascontiguousarray(a, dtype=np.float64)
        nb = a.nbytes
        if self.rank == 0:
            acc = a.copy()
            for c in self.hub:
                acc += np.frombuffer(_recv_exact(c, nb), dtype=np.float64).reshape(a.shape)
            blob = acc.tobytes()
            for c in self.hub:
                c.sendall(blob)
            a[...] = acc
            return a
        self.hub[0].sendall(a.tobytes())
        a[...] = np.frombuffer(_recv_exact(self.hub[0], nb), dtype=np.float64).reshape(a.shape)
        return a

    def allreduce_max(self, value):
        """max over ranks of a host double (also a barrier)"""
        if self.world == 1:
            return value
        if self.rank == 0:
            vals = [value] + [struct.unpack("d", _recv_exact(c, 8))[0] for c in self.hub]
            m = max(vals)
            for c in self.hub:
                c.sendall(struct.pack("d", m))
            return m
        self.hub[0].sendall(struct.pack("d", value))
        return struct.unpack("d", _recv_exact(self.hub[0], 8))[0]

    def barrier(self):
        self.allreduce_max(0.0)

    def sendrecv(self, peer, send, recv, tag=0):
        """Exchange with a neighbour rank: ``send`` goes out while ``recv`` (a writable float64 array) fills;
        interleaved with select() so that two ranks sending large pieces to each other cannot block.  Every message is
        FRAMED: a 24-byte header (magic, ``tag``, payload bytes) travels in front of the payload and the receiver checks it
        against what it expects -- two ranks whose calls do not pair up (different numbers of exchanges, a different
        window) raise here instead of reading each other's bytes as data."""
        c = self.nbr[int(peer)]
        out = memoryview(np.ascontiguousarray(send, dtype=np.float64)).cast("B") if send is not None and len(send) else memoryview(b"")
        inn = memoryview(recv).cast("B") if recv is not None and len(recv) else memoryview(bytearray(0))
        magic = 0x7469674672616D65
        hdr_out = memoryview(struct.pack("<QqQ", magic, int(tag), len(out)))
        hdr_in = memoryview(bytearray(24))
        outs, ins = [hdr_out, out], [hdr_in, inn]
        so, ro = [0, 0], [0, 0]
        checked = False

        def pending(bufs, pos):
            for k in (0, 1):
                if pos[k] < len(bufs[k]):
                    return k
            return -1
        c.setblocking(False)
        try:
            while True:
                ko, ki = pending(outs, so), pending(ins, ro)
                if ki != 0 and not checked:
                    m, t, nb = struct.unpack("<QqQ", bytes(hdr_in))
                    if m != magic or t != int(tag) or nb != len(inn):
                        raise RuntimeError("sendrecv with rank %d out of step: expected tag %d with %d bytes, the peer sent tag "
                                           "%d with %d bytes (magic %s) -- the ranks' exchanges do not pair up"
                                           % (peer, int(tag), len(inn), t, nb, "ok" if m == magic else "bad"))
                    checked = True
                if ko < 0 and ki < 0:
                    break
                rl, wl, _ = select.select([c] if ki >= 0 else [], [c] if ko >= 0 else [], [], 60.0)
                if not rl and not wl:
                    raise TimeoutError("sendrecv with rank %d stalled" % peer)
                if wl:
                    so[ko] += c.send(outs[ko][so[ko]:so[ko] + (1 << 20)])
                if rl:
                    got = c.recv_into(ins[ki][ro[ki]:], len(ins[ki]) - ro[ki])
                    if got == 0:
                        raise ConnectionError("neighbour rank %d closed the connection" % peer)
                    ro[ki] += got
        finally:
            c.setblocking(True)

    def close(self):
        for c in list(self.hub) + list(self.nbr.values()):
            try:
                c.close()
            except OSError:
                pass
        self.hub, self.nbr = [], {}


HostRendezvous = SocketTransport          # (name used by round-1 callers)


def device_comm(transport, kind=None):
    """Device communicator over ``transport``: RCCL when every rank has its own GPU, the IPC communicator when ranks
    share devices (``TIGAR_COMM=rccl|ipc|host`` forces one; ``host`` = staged through pinned memory and the transport,
    a host wait per exchange).  A communicator that cannot be formed on some rank -- all ranks learn it through the
    transport -- falls back in the order rccl -> ipc -> host instead of failing the run.  Returns None for a single
    rank."""
    from . import device as dev
    if transport.world == 1:
        return None
    kind = kind or os.environ.get("TIGAR_COMM")
    if kind is None:
        ndev = dev.device_count()
        local = int(os.environ.get("LOCAL_WORLD_SIZE", transport.world))
        kind = "rccl" if ndev >= local else "ipc"
    requested = kind
    notes = []               # why a communicator other than the requested one serves (kept on the object: bench lines quote it)

    def done(comm):
        comm.requested_kind, comm.fallback_notes = requested, list(notes)
        return comm
    if kind == "host":
        return done(dev.HostComm(transport))
    selftest_s = float(os.environ.get("TIGAR_COMM_SELFTEST_S", "60"))

    def agreed(comm, err, what, fallback):
        bad = np.array([0.0 if comm is not None else 1.0])
        transport.allreduce_sum(bad)
        if bad[0] > 0.0:
            notes.append("%s failed on %d rank(s)%s -> %s" % (what, int(bad[0]), (" (rank %d: %s)" % (transport.rank, err)) if err
                                                             is not None else "", fallback))
            if transport.rank == 0:
                print("[tigar_amd] %s communicator could not be created on %d rank(s) (%s); using the %s communicator"
                      % (what, int(bad[0]), err, fallback), file=sys.stderr, flush=True)
            return None
        return comm

    if kind == "rccl":
        comm, err = None, None
        try:
            uid = (dev.Comm.unique_id() + dev.Comm.unique_id()) if transport.rank == 0 else None
        except Exception as e:                      # (rank 0 only)
            uid, err = None, e
        flag = np.array([1.0 if (transport.rank == 0 and uid is None) else 0.0])
        transport.allreduce_sum(flag)
        if flag[0] == 0.0:
            uid = transport.broadcast_bytes(uid, 256)
            try:
                comm = dev.Comm(uid[:128], transport.rank, transport.world, unique_id_halo=uid[128:])
                comm.selftest(selftest_s)
            except Exception as e:
                comm, err = None, e
        comm = agreed(comm, err, "RCCL", "IPC")
        if comm is not None:
            return done(comm)
    comm, err = None, None
    try:
        comm = dev.IpcComm(transport)
        comm.selftest(selftest_s)
    except Exception as e:
        comm, err = None, e
    comm = agreed(comm, err, "IPC", "host-staged")
    if comm is not None:
        return done(comm)
    return done(dev.HostComm(transport))


def spawn_local(nproc, argv, env_extra=None, port=None):
    """Launches ``nproc`` ranks of ``argv`` (a python command line without the interpreter) on this node
    with the environment ``torch.distributed.run`` would set, waits for them and returns rank 0's exit
    code (non-zero if any rank failed).  Children are started in their own process group and killed by
    pid if one of them fails."""
    port = port or (29500 + (os.getpid() * 7) % 2000)
    procs = []
    for r in range(nproc):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nproc), "LOCAL_WORLD_SIZE": str(nproc),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        if env_extra:
            env.update(env_extra)
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=env))
    rcs = [None] * nproc
    try:
        while any(rc is None for rc in rcs):
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = pr.poll()
            if any(rc not in (None, 0) for rc in rcs):
                break
            time.sleep(0.05)
    finally:
        for i, pr in enumerate(procs):
            if pr.poll() is None:
                if any(rc not in (None, 0) for rc in rcs):
                    pr.kill()
                rcs[i] = pr.wait()
    bad = [rc for rc in rcs if rc != 0]
    return bad[0] if bad else 0
