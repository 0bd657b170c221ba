# gpurun -- "bash tools/gpu_fmgtrace.sh [key=value ...]": kernel timeline of one FMG solve at 512^3 -> gpurun_out/fmgtrace.txt
set -o pipefail
R=$PWD
rm -rf $R/gpurun_out/fmgtrace
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fmgtrace -- python3 $R/tools/fmg_trace.py "$@" > $R/gpurun_out/fmgtrace.log 2>&1; rc=$?
tail -3 $R/gpurun_out/fmgtrace.log
if [ $rc -ne 0 ]; then exit $rc; fi
cd $R && python3 tools/vcycle_trace_reduce.py gpurun_out/fmgtrace summary > gpurun_out/fmgtrace.txt; cat gpurun_out/fmgtrace.txt
rm -rf $R/gpurun_out/fmgtrace
