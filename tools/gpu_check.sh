# Usage (from the repository root, on a GPU box):  gpurun -- "bash tools/gpu_check.sh"   -- the whole -m gpu suite and smoke()
set -o pipefail
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/t14.log 2>&1; rc=$?; tail -8 gpurun_out/t14.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
