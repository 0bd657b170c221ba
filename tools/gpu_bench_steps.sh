# gpurun -- "bash tools/gpu_bench_steps.sh": the headline at 20 / 50 / 200 timed steps (they agree once the clock-settle phase has run)
for K in 20 50 200; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-kernel-table --steps $K --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('steps', r['steps'], 'vcycle', r.get('vcycle_ms'), 'solve', r.get('totalTimeSolve_ms'), 'ms_per_step %.4f' % r['ms_per_step'], 'kernel_ms/2 %.4f' % (r['roofline']['kernel_ms']/2), 'value %.4g' % r['value'])"
done
