set -o pipefail
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for halo in deep shell; do
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 50 --warmup 10 --no-cpu-baseline --backend gloo --halo $halo > gpurun_out/bench_n2_$halo.json 2> gpurun_out/bench_n2.err; rc=$?
[ $rc -ne 0 ] && { tail -5 gpurun_out/bench_n2.err; exit $rc; }
python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_n2_$halo.json') if l.startswith('{')][-1])
print('$halo', {k: d.get(k) for k in ('value','ms_per_step','sustained_ms_per_step','vcycle_ms','totalTimeSolve_ms','solve_iterations','duplicate_planes_bit_identical','vcycle_duplicate_planes_bit_identical','transport')})"
done
