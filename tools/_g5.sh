R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_solver.py tests/test_gpu_exa4.py tests/test_gpu_fullsize.py -x -q > gpurun_out/t5.log 2>&1; rc=$?; tail -5 gpurun_out/t5.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python3 tools/pmc_kernels.py --time --no27 2>&1 | grep -v amdgpu
timeout -k 10 600 python bench.py --no-cpu-baseline --no-kernel-table 2>&1 | grep -v amdgpu | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','vcycle_ms','totalTimeSolve_ms','jacobi_256cube_two_step_kernel_ms','jacobi_256cube_single_step_kernel_ms')}, d['roofline']['frac'], d['roofline']['kernel_ms'])"
