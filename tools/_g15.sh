set -o pipefail
R=$PWD
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -x -q > gpurun_out/t15.log 2>&1; rc=$?; tail -8 gpurun_out/t15.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/vtrace
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vtrace -- python3 $R/tools/vcycle_trace.py > $R/gpurun_out/vtrace.log 2>&1; rc=$?
if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/vtrace.log; exit $rc; fi
cd $R && python3 tools/vcycle_trace_reduce.py > gpurun_out/vtrace.txt; tail -3 gpurun_out/vtrace.txt
timeout -k 10 300 python tools/vcycle_by_level.py 2>&1 | tail -6
