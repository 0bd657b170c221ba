set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "two_stage or prolong" > gpurun_out/t11.log 2>&1; rc=$?; tail -15 gpurun_out/t11.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python tools/sweep_prolong_fold.py 2>&1 | tee gpurun_out/s11.log
