# gpurun -- "bash tools/gpu_pmc.sh": timing pass, FETCH_SIZE / WRITE_SIZE passes and the four why-counter sets of tools/pmc_kernels.py for the verbatim and
# the align=16 layout; afterwards, locally: cp gpurun_out/r02_pmc_kernels*.json profiles/ (see the file names in the script); python3 tools/pmc_why.py [--tag _a16]
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/pmcK_FETCH* gpurun_out/pmcK_WRITE* gpurun_out/pmcY_*
for AL in 0 16; do
TAG=""; [ $AL -ne 0 ] && TAG="_a$AL"
timeout -k 10 300 python3 tools/pmc_kernels.py --time --align $AL > gpurun_out/pmcK_times$TAG.log 2>&1 || { tail -20 gpurun_out/pmcK_times$TAG.log; exit 1; }
cat gpurun_out/pmcK_times$TAG.log
for C in FETCH WRITE; do
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc ${C}_SIZE --output-format csv -d $R/gpurun_out/pmcK_$C$TAG -- python3 $R/tools/pmc_kernels.py --align $AL > $R/gpurun_out/pmcK_$C$TAG.log 2>&1 ) || { tail -20 gpurun_out/pmcK_$C$TAG.log; exit 1; }
done
python3 tools/pmc_reduce.py --tag "$TAG" --out gpurun_out/r02_pmc_kernels$TAG.json
for S in A B C D; do
CTRS=$(python3 tools/pmc_why.py sets | grep "^$S " | cut -d' ' -f2-)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/pmcY_$S$TAG -- python3 $R/tools/pmc_kernels.py --align $AL > $R/gpurun_out/pmcY_$S$TAG.log 2>&1 ) || { tail -20 gpurun_out/pmcY_$S$TAG.log; exit 1; }
echo "set $S align $AL done"
done
python3 tools/pmc_why.py --tag "$TAG"
done
