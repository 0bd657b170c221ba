set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/lab/regrow_probe.py > gpurun_out/regrow.txt 2>&1; rc=$?
grep "ok:\|Error\|error" gpurun_out/regrow.txt | head -30
echo rc=$rc
EXAMG_HOSTED_RANKS=1 timeout -k 10 500 python -m pytest tests/test_gpu_peer.py -x -q -m gpu -k eight_ranks > gpurun_out/hosted_8ranks.log 2>&1; echo "8 ranks hosted rc=$?"; tail -3 gpurun_out/hosted_8ranks.log
