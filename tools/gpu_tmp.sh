set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "27 or entry_fastest or stencil_field" > gpurun_out/t27.log 2>&1; rc=$?; tail -5 gpurun_out/t27.log
[ $rc -ne 0 ] && exit $rc
for n in 512 256 128; do timeout -k 10 400 python tools/time_sf27.py $n --dbg > gpurun_out/time_sf27_$n.txt 2>&1 || { tail -5 gpurun_out/time_sf27_$n.txt; exit 1; }; grep "one step\|residual" gpurun_out/time_sf27_$n.txt; done
