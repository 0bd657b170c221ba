set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_solver.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -x -q > gpurun_out/t18.log 2>&1; rc=$?; tail -8 gpurun_out/t18.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --no-cpu-baseline --steps 50 --warmup 5 > gpurun_out/b18.log 2> gpurun_out/b18.err; rc=$?; tail -3 gpurun_out/b18.err
python - <<'PY'
import json
r = json.loads(open("gpurun_out/b18.log").read().strip().splitlines()[-1])
print("value %.4g frac %.3f vcycle %s solve %s its %s" % (r["value"], r["roofline"]["frac"], r.get("vcycle_ms"), r.get("totalTimeSolve_ms"), r.get("solve_iterations")))
for k in r.get("roofline_kernels", []):
    print("   %-26s %.4f ms frac %.3f" % (k["case"], k["ms"], k["frac"]))
PY
exit $rc
