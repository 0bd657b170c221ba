# gpurun -- "bash tools/gpu_vtrace.sh": kernel timeline of one V(3,3) cycle at 512^3 (rocprofv3 --kernel-trace) -> gpurun_out/vtrace.txt
set -o pipefail
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vtrace -- python3 $R/tools/vcycle_trace.py > $R/gpurun_out/vtrace.log 2>&1; rc=$?
tail -3 $R/gpurun_out/vtrace.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd $R && python3 tools/vcycle_trace_reduce.py > gpurun_out/vtrace.txt; tail -5 gpurun_out/vtrace.txt
