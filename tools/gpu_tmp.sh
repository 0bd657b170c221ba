set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests.log 2>&1; rc=$?; tail -4 gpurun_out/r04_gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
ROUND=r04 bash tools/gpu_profiles.sh
