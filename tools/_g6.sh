R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t6.log 2>&1; rc=$?; tail -6 gpurun_out/t6.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python3 tools/pmc_kernels.py --time 2>&1 | grep -v amdgpu
