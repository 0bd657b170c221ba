R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_transport.py tests/test_gpu_shim.py -x -q > gpurun_out/t3.log 2>&1; rc=$?; tail -30 gpurun_out/t3.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/bench_r02a.log 2>&1; tail -c 6000 gpurun_out/bench_r02a.log
