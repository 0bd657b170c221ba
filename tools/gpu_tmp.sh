set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; rc=$?; tail -4 gpurun_out/r04_gputests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/vcycle_neighbours.py 9 > gpurun_out/vcycle_neighbours.json 2> gpurun_out/vcycle_neighbours.err || { tail -5 gpurun_out/vcycle_neighbours.err; exit 1; }
cat gpurun_out/vcycle_neighbours.json
