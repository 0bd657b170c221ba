R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest tests/test_gpu_transport.py tests/test_gpu_solver.py tests/test_gpu_multi.py -x -q > gpurun_out/t4.log 2>&1; rc=$?; tail -15 gpurun_out/t4.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
for sc in weak strong; do
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 10 --warmup 2 --level 7 --backend gloo --check-duplicates --scaling $sc --no-cpu-baseline > gpurun_out/reh_n2_$sc.log 2>&1; rc=$?; tail -c 1500 gpurun_out/reh_n2_$sc.log; echo rc=$rc
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
done
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 4 --steps 10 --warmup 2 --level 7 --backend gloo --check-duplicates --scaling strong --no-cpu-baseline > gpurun_out/reh_n4_strong.log 2>&1; rc=$?; tail -c 1500 gpurun_out/reh_n4_strong.log; echo rc=$rc
