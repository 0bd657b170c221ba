R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for ARGS in "--gpus 1 --steps 20 --warmup 5" "--steps 50 --warmup 5 --align 16 --no-cpu-baseline" "--steps 50 --warmup 5 --no-temporal-blocking --no-cpu-baseline --no-vcycle"; do
T0=$(date +%s.%N); timeout -k 10 600 python bench.py $ARGS > gpurun_out/b10.log 2> gpurun_out/b10.err; rc=$?; echo "wall $(echo "$(date +%s.%N) - $T0" | bc) s"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/b10.log') if l.startswith('{')][-1])
r=d['roofline']
print(d['config']['align'], d['temporal_blocking'], 'value %.3e' % d['value'], 'frac %.3f' % r['frac'], 'kernel_ms %.4f' % r['kernel_ms'], 'traffic', r['traffic'], 'vcycle', d.get('vcycle_ms'), 'solve', d.get('totalTimeSolve_ms'), 'cpu', ('%.3e' % d['cpu_baseline']['value']) if 'cpu_baseline' in d else None)
if 'roofline_kernels' in d:
    for k in d['roofline_kernels']:
        print('   %-24s %.4f ms frac %.3f' % (k['case'], k['ms'], k['frac']))
PY
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc
done
