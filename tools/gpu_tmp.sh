set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/ab_ntload.py > gpurun_out/ab_ntload.txt 2>&1 || { tail -5 gpurun_out/ab_ntload.txt; exit 1; }; cat gpurun_out/ab_ntload.txt
