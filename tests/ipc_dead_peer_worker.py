"""Rank of test_ipc_dead_peer_is_an_error_not_a_hang: rank 1 leaves after the communicator is up, rank 0 then asks
for an all-reduce; the wait inside the kernel gives up after TIGAR_IPC_TIMEOUT_S and the call raises."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    outdir = sys.argv[1]
    from tigar_amd import common as tc
    from tigar_amd._lib import TigarHipError
    comm = tc.worldcomm
    dcomm = comm.device()
    assert dcomm.info()[2] == "ipc"
    assert list(dcomm.allreduce_sum([1.0])) == [float(comm.size)]
    comm.barrier()
    if comm.rank != 0:
        return                                   # gone before the next exchange
    t0 = time.time()
    try:
        dcomm.allreduce_sum([1.0])
        msg = "no error"
    except TigarHipError as e:
        msg = str(e)
    with open(os.path.join(outdir, "rank0.txt"), "w") as f:
        f.write("%.2f\n%s\n" % (time.time() - t0, msg))


if __name__ == "__main__":
    main()
