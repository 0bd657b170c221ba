"""One PROCESS of a multi-rank GPU rehearsal that hosts several ranks: one thread, one HIP stream and one FileDist (tests/filedist.py)
per rank.  The GPU box lets fewer than 8 processes use its card at once, and the 8-rank decompositions (2 x 2 x 2 of SURVEY.md 8e,
bench.py's 1 x 2 x 4) are what the driver's 8-GPU run will execute -- so tests/test_gpu_ranks8.py starts e.g. 4 of these with 2 ranks
each.  Ranks of one process reach each other's peer-write regions by address, ranks of other processes through HIP IPC
(csrc/examg_peer.hip).  Test infrastructure.

usage: ranks_host.py worker <proc> <nprocs> <ranks_per_proc> <bx,by,bz> <bootstrap_dir> <dir>          (tests/peer_worker.py: run_rank)
       ranks_host.py bench  <proc> <nprocs> <ranks_per_proc> <bootstrap_dir> <out.json> -- <bench.py arguments>
Rank = proc * ranks_per_proc + thread."""
import json
import os
import sys
import threading
import traceback

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def main():
    mode, proc, nprocs, per = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    from filedist import FileDist

    torch.cuda.set_device(0)
    world = nprocs * per
    failures = []

    def rank_thread(tid):
        rank = proc * per + tid
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):       # a rank's launches wait for its neighbours': never share a stream
                if mode == "worker":
                    import peer_worker

                    blocks = tuple(int(x) for x in sys.argv[5].split(","))
                    d = FileDist(sys.argv[6], rank, world)
                    peer_worker.run_rank(rank, world, blocks, sys.argv[7], d, d)
                    torch.cuda.current_stream().synchronize()
                    d.barrier()
                else:
                    import bench

                    d = FileDist(sys.argv[5], rank, world)
                    args = bench.parse(sys.argv[sys.argv.index("--") + 1:])
                    out = bench.run(args, world, rank, 0, d, injected=True)
                    torch.cuda.current_stream().synchronize()
                    d.barrier()
                    if rank == 0:
                        json.dump(out, open(sys.argv[6], "w"))
        except BaseException:      # noqa: BLE001 -- the other threads would wait for this rank until their timeouts: end the process now
            failures.append(rank)
            sys.stderr.write("rank %d failed:\n%s\n" % (rank, traceback.format_exc()))
            sys.stderr.flush()
            os._exit(7)

    threads = [threading.Thread(target=rank_thread, args=(t,)) for t in range(per)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
