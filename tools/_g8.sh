R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02s -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-vcycle > $R/gpurun_out/prof_r02s_bench.log 2>&1 ) || { tail -20 gpurun_out/prof_r02s_bench.log; exit 1; }
grep "^{" gpurun_out/prof_r02s_bench.log | tail -1 | cut -c1-1500
cat gpurun_out/prof_r02s/*/*kernel_stats.csv | cut -c1-60,200-330 | head -8
