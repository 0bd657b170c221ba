# gpurun -- "ROUND=r04 bash tools/gpu_profiles.sh": everything profiles/<ROUND>_* comes from, on ONE box -- the rocprofv3 kernel statistics of the
# default bench command and of its smoother-only form, the kernel timeline of one V-cycle, the counter passes at 512^3 and 256^3
# (tools/gpu_pmc.sh), the default bench line, face-exchange costs, the small-level A/B and the interpreter's cycle.  Results under gpurun_out/:
# copy <ROUND>_* and the *_kernel_stats.csv into profiles/ (profiles/README.md lists what is what).
set -o pipefail
ROUND=${ROUND:-r04}
R=$PWD
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${ROUND}_smoother -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-vcycle --no-kernel-table --sustained-seconds 0 > $R/gpurun_out/${ROUND}_bench_smoother_only.json 2> $R/gpurun_out/prof_${ROUND}_smoother.err; rc=$?
tail -c 600 $R/gpurun_out/${ROUND}_bench_smoother_only.json; echo
if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_${ROUND}_smoother.err; exit $rc; fi
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${ROUND}_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 1 > $R/gpurun_out/${ROUND}_bench_under_rocprof.json 2> $R/gpurun_out/prof_${ROUND}_bench.err; rc=$?
tail -c 300 $R/gpurun_out/${ROUND}_bench_under_rocprof.json; echo
if [ $rc -ne 0 ]; then tail -5 $R/gpurun_out/prof_${ROUND}_bench.err; exit $rc; fi
cd $R
for d in smoother bench; do f=$(ls gpurun_out/prof_${ROUND}_$d/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f gpurun_out/${ROUND}_bench$([ $d = smoother ] && echo _smoother_only)_kernel_stats.csv; done
bash tools/gpu_vtrace.sh && cp gpurun_out/vtrace.txt gpurun_out/${ROUND}_vcycle_timeline.txt
( cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fmgtrace -- python3 $R/tools/fmg_trace.py > $R/gpurun_out/fmgtrace.log 2>&1 ) && python3 tools/vcycle_trace_reduce.py gpurun_out/fmgtrace > gpurun_out/${ROUND}_fmg_timeline.txt && tail -1 gpurun_out/${ROUND}_fmg_timeline.txt
LEVEL=9 ROUND=$ROUND bash tools/gpu_pmc.sh || exit 1
LEVEL=8 ROUND=$ROUND bash tools/gpu_pmc.sh || exit 1
timeout -k 10 200 python tools/exchange_faces.py > gpurun_out/${ROUND}_exchange_faces.json 2> /dev/null; cat gpurun_out/${ROUND}_exchange_faces.json
timeout -k 10 300 python3 tools/ab_small.py > gpurun_out/${ROUND}_ab_small.json 2> /dev/null; cat gpurun_out/${ROUND}_ab_small.json
timeout -k 10 200 python tools/exa4_time.py > gpurun_out/${ROUND}_exa4_time.json 2> /dev/null; cat gpurun_out/${ROUND}_exa4_time.json
timeout -k 10 600 python bench.py > gpurun_out/${ROUND}_bench.json 2> gpurun_out/${ROUND}_bench.err || { tail -5 gpurun_out/${ROUND}_bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/${ROUND}_bench.json'))
print({k: d[k] for k in ('value','ms_per_step','vcycle_ms','totalTimeSolve_ms','jacobi_single_step_frac','jacobi_256cube_single_step_frac','jacobi_256cube_two_step_frac','helmholtz27_vcycle_ms','fmg_solve_ms','shim_vcycle_ms_plain','shim_vcycle_ms_deferred') if k in d}); print(d['roofline'])"
