R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/pmcK_FETCH* gpurun_out/pmcK_WRITE*
for AL in 0 16; do
TAG=""; [ $AL -ne 0 ] && TAG="_a$AL"
timeout -k 10 300 python3 tools/pmc_kernels.py --time --align $AL > gpurun_out/pmcK_times$TAG.log 2>&1 || { tail -20 gpurun_out/pmcK_times$TAG.log; exit 1; }
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcK_FETCH$TAG -- python3 $R/tools/pmc_kernels.py --align $AL > $R/gpurun_out/pmcK_F$TAG.log 2>&1 ) || { tail -20 gpurun_out/pmcK_F$TAG.log; exit 1; }
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcK_WRITE$TAG -- python3 $R/tools/pmc_kernels.py --align $AL > $R/gpurun_out/pmcK_W$TAG.log 2>&1 ) || { tail -20 gpurun_out/pmcK_W$TAG.log; exit 1; }
python3 tools/pmc_reduce.py --tag "$TAG" --out gpurun_out/r02_pmc_kernels$TAG.json
done
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/prof_r02_bench.log 2>&1 ) || { tail -20 gpurun_out/prof_r02_bench.log; exit 1; }
tail -c 600 gpurun_out/prof_r02_bench.log; find gpurun_out/prof_r02 -name "*stats*" | head
