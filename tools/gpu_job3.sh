set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests.log 2>&1; rc=$?; tail -15 gpurun_out/r04_gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
bash tools/gpu_vtrace.sh
timeout -k 10 300 python3 tools/pmc_kernels.py --time --level 9 --no27 2>&1 | tail -20
timeout -k 10 300 python3 tools/sweep_rowmarch.py 256 2>&1 | tail -30
EXAMG_HOSTED_RANKS=1 timeout -k 10 500 python -m pytest tests/test_gpu_peer.py -x -q -m gpu -k eight_ranks > gpurun_out/r04_hosted_peer.log 2>&1; echo "hosted peer rc=$?"; tail -5 gpurun_out/r04_hosted_peer.log
