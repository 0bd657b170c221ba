set -o pipefail
mkdir -p gpurun_out
EXAMG_HOSTED_RANKS=1 timeout -k 10 900 python -m pytest tests/test_gpu_multi.py -x -q -m gpu -k eight_ranks > gpurun_out/r04_hosted_bench.log 2>&1; echo "hosted bench rc=$?"; tail -5 gpurun_out/r04_hosted_bench.log
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_mid.json 2> gpurun_out/r04_bench_mid.err || { tail -5 gpurun_out/r04_bench_mid.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_mid.json'))
print({k: d[k] for k in ('value','ms_per_step','vcycle_ms','totalTimeSolve_ms','jacobi_single_step_frac','jacobi_256cube_single_step_frac','jacobi_256cube_two_step_frac','helmholtz27_vcycle_ms','fmg_solve_ms','shim_vcycle_ms_plain','shim_vcycle_ms_deferred','shim_launches_per_cycle_plain','shim_launches_per_cycle_deferred','shim_error') if k in d}); print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline'].get('loop_shape'))"
timeout -k 10 200 python tools/exa4_time.py 2>&1 | tail -3
