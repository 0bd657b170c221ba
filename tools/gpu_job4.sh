set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests.log 2>&1; rc=$?; tail -8 gpurun_out/r04_gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 tools/ab_small.py > gpurun_out/r04_ab_small.json 2> gpurun_out/r04_ab_small.err; cat gpurun_out/r04_ab_small.json; tail -3 gpurun_out/r04_ab_small.err
EXAMG_HOSTED_RANKS=1 timeout -k 10 500 python -m pytest tests/test_gpu_peer.py -x -q -m gpu -k eight_ranks > gpurun_out/r04_hosted_peer.log 2>&1; echo "hosted peer rc=$?"; tail -5 gpurun_out/r04_hosted_peer.log
