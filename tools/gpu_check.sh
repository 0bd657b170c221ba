# Usage (from the repository root, on a GPU box):  gpurun -- "bash tools/gpu_check.sh"   -- the whole -m gpu suite, smoke(), and the opt-in
# rehearsals of the 8-rank job (4 processes x 2 ranks on the one GPU: tests/ranks_host.py)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gputests.log 2>&1; rc=$?; tail -8 gpurun_out/gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
EXAMG_HOSTED_RANKS=1 timeout -k 10 500 python -m pytest tests/test_gpu_peer.py -x -q -m gpu -k eight_ranks > gpurun_out/hosted_8ranks.log 2>&1; echo "8 ranks hosted (peer worker) rc=$?"; tail -3 gpurun_out/hosted_8ranks.log
