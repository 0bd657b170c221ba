set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/ab_nt.py 64 128 192 256 > gpurun_out/ab_nt2.txt 2>&1 || { tail -5 gpurun_out/ab_nt2.txt; exit 1; }; cat gpurun_out/ab_nt2.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_gputests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/vcycle_trace.py > /dev/null 2>&1; bash tools/gpu_vtrace.sh > /dev/null 2>&1; tail -3 gpurun_out/vtrace.txt
