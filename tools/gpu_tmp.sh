set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "27_entry" > gpurun_out/t27.log 2>&1; rc=$?; tail -5 gpurun_out/t27.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/time_sf27.py 512 --dbg > gpurun_out/time_sf27.txt 2>&1; rc=$?; cat gpurun_out/time_sf27.txt | tail -30
exit $rc
