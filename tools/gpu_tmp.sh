set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/ab_rr.py > gpurun_out/ab_rr.txt 2>&1 || { tail -5 gpurun_out/ab_rr.txt; exit 1; }; cat gpurun_out/ab_rr.txt
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_multi.py tests/test_gpu_solver.py -m gpu -x -q > gpurun_out/t.log 2>&1; rc=$?; tail -5 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
for n in 256 128; do timeout -k 10 300 python tools/time_sf27.py $n --dbg > gpurun_out/time_sf27_$n.txt 2>&1 || exit 1; grep "one pass" gpurun_out/time_sf27_$n.txt; done
