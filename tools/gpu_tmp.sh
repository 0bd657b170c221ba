set -o pipefail
mkdir -p gpurun_out
python - <<'PY'
import subprocess, __graft_entry__ as ge
exe = ge.build_example(name="poisson3d_fast_host")
for rep in range(2):
    out = subprocess.run([exe, "9", "4", "10000000"], capture_output=True, text=True, timeout=300)
    print(out.returncode, [l for l in out.stdout.splitlines() if not l.startswith("# ")][-6:], out.stderr[-300:])
PY
