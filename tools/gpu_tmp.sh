set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "27_entry or two_stage_kernel_bit" > gpurun_out/t27.log 2>&1; rc=$?; tail -8 gpurun_out/t27.log
exit $rc
