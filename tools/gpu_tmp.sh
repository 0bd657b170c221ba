set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_peer.py tests/test_gpu_multi.py -m gpu -x -q -k "multi_process_on_one_gpu or bench" > gpurun_out/t.log 2>&1; rc=$?; tail -25 gpurun_out/t.log | cut -c1-300
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python tools/vcycle_neighbours.py 9 lone_block neighbours_z neighbours_y_z neighbours_x_y_z > gpurun_out/vcycle_neighbours.json 2> gpurun_out/vcycle_neighbours.err || { tail -5 gpurun_out/vcycle_neighbours.err; exit 1; }
cat gpurun_out/vcycle_neighbours.json
