set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_solver.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q -k "fmg or FMG or faces or expression or config4" > gpurun_out/t.log 2>&1; rc=$?; tail -5 gpurun_out/t.log
[ $rc -ne 0 ] && exit $rc
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fmgtrace2 -- python3 $R/tools/fmg_trace.py > $R/gpurun_out/fmgtrace.log 2>&1; rc=$?
tail -2 $R/gpurun_out/fmgtrace.log
[ $rc -ne 0 ] && exit $rc
cd $R && python3 tools/vcycle_trace_reduce.py gpurun_out/fmgtrace2 > gpurun_out/fmgtrace2.txt; python3 tools/vcycle_trace_reduce.py gpurun_out/fmgtrace2 summary | tail -4
python tools/fmg_time.py
