# gpurun -- "bash tools/gpu_profiles.sh": the rocprofv3 kernel statistics of the default bench command and of its smoother-only form, and the
# kernel timeline of one V-cycle.  Results under gpurun_out/prof_r03*/ and gpurun_out/vtrace.txt -> copy the *_kernel_stats.csv into profiles/.
set -o pipefail
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_smoother -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-vcycle --no-kernel-table --sustained-seconds 0 > $R/gpurun_out/prof_r03_smoother.json 2> $R/gpurun_out/prof_r03_smoother.err; rc=$?
tail -c 600 $R/gpurun_out/prof_r03_smoother.json; echo
if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_r03_smoother.err; exit $rc; fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 1 > $R/gpurun_out/prof_r03_bench.json 2> $R/gpurun_out/prof_r03_bench.err; rc=$?
tail -c 300 $R/gpurun_out/prof_r03_bench.json; echo
if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_r03_bench.err; exit $rc; fi
cd $R && bash tools/gpu_vtrace.sh
