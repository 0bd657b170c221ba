set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_multi.py tests/test_gpu_peer.py tests/test_gpu_shim.py -x -q -m gpu > gpurun_out/r04_gputests2.log 2>&1; rc=$?; tail -15 gpurun_out/r04_gputests2.log
timeout -k 10 120 tools/barrier_probe.bin > gpurun_out/r04_barrier_probe.txt 2>&1; cat gpurun_out/r04_barrier_probe.txt
exit $rc
