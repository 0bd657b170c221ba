# gpurun -- "bash tools/gpu_fast_host.sh": the C++ host on the one-pass entry points at the benchmark's size (levels 4..9) and its test
set -o pipefail
timeout -k 10 300 ./examples/poisson3d_fast_host 9 4 | tail -12
timeout -k 10 300 python -m pytest tests/test_gpu_solver.py -x -q -k "cpp" 2>&1 | tail -3
