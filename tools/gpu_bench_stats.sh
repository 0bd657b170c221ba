# gpurun -- "bash tools/gpu_bench_stats.sh": rocprofv3 --kernel-trace --stats of the driver's bench command (with and without the
# V-cycle extras) -> gpurun_out/benchstats_{full,smoother}/..._kernel_stats.csv + the JSON lines; copy them to profiles/
set -o pipefail
R=$PWD; mkdir -p gpurun_out; rm -rf gpurun_out/benchstats_*
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/benchstats_smoother -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-vcycle --no-kernel-table > $R/gpurun_out/benchstats_smoother.json 2> $R/gpurun_out/benchstats_smoother.err || { tail -5 $R/gpurun_out/benchstats_smoother.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/benchstats_full -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/gpurun_out/benchstats_full.json 2> $R/gpurun_out/benchstats_full.err || { tail -5 $R/gpurun_out/benchstats_full.err; exit 1; }
cd $R
for d in smoother full; do f=$(ls -t gpurun_out/benchstats_$d/*/*kernel_stats.csv | head -1); echo "== $d"; head -6 $f | cut -c1-160; tail -1 gpurun_out/benchstats_$d.json | cut -c1-300; done
rm -rf gpurun_out/vtrace; bash tools/gpu_vtrace.sh
