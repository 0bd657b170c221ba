# gpurun -- "bash tools/gpu_rehearse_multi.sh [level]": bench.py at N = 2 (weak, strong) and N = 4 (strong) on the ONE GPU of the box -- halo traffic
# device to device through the peer-write transport, gloo only for the bootstrap and the timing collectives
L=${1:-7}
set -o pipefail
for sc in weak strong; do
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 2 --level $L --backend gloo --scaling $sc --no-cpu-baseline > gpurun_out/reh_n2_$sc.log 2>&1; rc=$?; tail -1 gpurun_out/reh_n2_$sc.log | cut -c1-600; echo rc=$rc
if [ $rc -ne 0 ]; then tail -20 gpurun_out/reh_n2_$sc.log; exit $rc; fi
done
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 4 --steps 10 --warmup 2 --level $L --backend gloo --scaling strong --no-cpu-baseline > gpurun_out/reh_n4_strong.log 2>&1; rc=$?; tail -1 gpurun_out/reh_n4_strong.log | cut -c1-600; echo rc=$rc
exit $rc
