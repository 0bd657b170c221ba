R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py -x -q > gpurun_out/t2.log 2>&1; rc=$?; tail -15 gpurun_out/t2.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python3 tools/pmc_kernels.py --time --no27 > gpurun_out/pmcK_times_b.log 2>&1; cat gpurun_out/pmcK_times_b.log
timeout -k 10 300 python3 tools/pmc_kernels.py --time --no27 --align 16 > gpurun_out/pmcK_times_a16.log 2>&1; cat gpurun_out/pmcK_times_a16.log
