# gpurun -- "bash tools/gpu_bench_variants.sh": bench.py as the driver calls it, on the padded layout, and one step per launch
for ARGS in "--gpus 1 --steps 20 --warmup 5" "--steps 50 --warmup 5 --align 16 --no-cpu-baseline" "--steps 50 --warmup 5 --no-temporal-blocking --no-cpu-baseline --no-kernel-table"; do
timeout -k 10 600 python bench.py $ARGS > gpurun_out/bv.log 2> gpurun_out/bv.err; rc=$?
python - <<PY
import json
r = json.loads(open("gpurun_out/bv.log").read().strip().splitlines()[-1])
print("$ARGS -> rc=$rc value %.4g frac %.3f kernel_ms %.4f traffic %s vcycle %s solve %s" % (r["value"], r["roofline"]["frac"], r["roofline"]["kernel_ms"], r["roofline"]["traffic"], r.get("vcycle_ms"), r.get("totalTimeSolve_ms")))
PY
done
