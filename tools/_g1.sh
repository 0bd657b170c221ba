R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -q > gpurun_out/fullsize.log 2>&1; rc=$?; tail -5 gpurun_out/fullsize.log
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
timeout -k 10 300 python3 tools/pmc_kernels.py --time > gpurun_out/pmcK_times.log 2>&1 || { tail -20 gpurun_out/pmcK_times.log; exit 1; }
cat gpurun_out/pmcK_times.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcK_FETCH -- python3 $R/tools/pmc_kernels.py > $R/gpurun_out/pmcK_F.log 2>&1 || { tail -20 $R/gpurun_out/pmcK_F.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcK_WRITE -- python3 $R/tools/pmc_kernels.py > $R/gpurun_out/pmcK_W.log 2>&1 || { tail -20 $R/gpurun_out/pmcK_W.log; exit 1; }
cd $R && python3 tools/pmc_reduce.py --out gpurun_out/r02_pmc_kernels.json
