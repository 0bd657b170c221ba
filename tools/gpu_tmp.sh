set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_gputests.log
[ $rc -ne 0 ] && exit $rc
ROUND=r04 bash tools/gpu_profiles.sh
