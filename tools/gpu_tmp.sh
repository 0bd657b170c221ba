set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err || { tail -5 gpurun_out/r04_bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r04_bench.json'))
print({k: d[k] for k in ('value','ms_per_step','vcycle_ms','totalTimeSolve_ms','jacobi_single_step_frac','jacobi_256cube_single_step_frac','helmholtz27_vcycle_ms','fmg_solve_ms','shim_vcycle_ms_deferred') if k in d}); print(d['roofline'])"
