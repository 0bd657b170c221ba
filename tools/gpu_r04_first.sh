# round 4: the whole -m gpu suite, face-exchange costs, level-8 counters, a bench line, the V-cycle timeline; last (opt-in, may fail): the
# 8-rank decompositions hosted as 4 processes x 2 ranks
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputests.log 2>&1; rc=$?; tail -15 gpurun_out/r04_gputests.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 200 python tools/exchange_faces.py > gpurun_out/r04_exchange_faces.json 2> gpurun_out/r04_exchange_faces.err || { tail -5 gpurun_out/r04_exchange_faces.err; exit 1; }
cat gpurun_out/r04_exchange_faces.json
LEVEL=8 ROUND=r04 bash tools/gpu_pmc.sh || exit 1
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_first.json 2> gpurun_out/r04_bench_first.err || { tail -5 gpurun_out/r04_bench_first.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_first.json'))
print({k: d[k] for k in ('value','ms_per_step','vcycle_ms','jacobi_single_step_frac','jacobi_256cube_single_step_frac','jacobi_256cube_two_step_frac','helmholtz27_vcycle_ms','fmg_solve_ms','shim_vcycle_ms_plain','shim_vcycle_ms_deferred','shim_launches_per_cycle_plain','shim_launches_per_cycle_deferred','shim_error') if k in d}); print(d['roofline'])"
bash tools/gpu_vtrace.sh
EXAMG_HOSTED_RANKS=1 timeout -k 10 500 python -m pytest tests/test_gpu_peer.py -x -q -m gpu -k eight_ranks > gpurun_out/r04_hosted_peer.log 2>&1; echo "hosted peer rc=$?"; tail -5 gpurun_out/r04_hosted_peer.log
