# scratch: the command line of the moment (tools/gpu_check.sh and tools/gpu_profiles.sh are the kept ones)
set -o pipefail
mkdir -p gpurun_out
ROUND=r04 bash tools/gpu_profiles.sh
