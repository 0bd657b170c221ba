# gpurun -- "LEVEL=8 ROUND=r04 bash tools/gpu_pmc.sh": per-kernel fabric traffic and the why-counters of the hot kernels (tools/pmc_kernels.py),
# verbatim layout, at 2^LEVEL cells per dimension (default 9).  Counter passes run on their own (--pmc only, no trace options: pool rule).
# Results: gpurun_out/<ROUND>_pmc_kernels[_L<level>].json, <ROUND>_pmc_why[_L<level>].json -> copy into profiles/ (tools/pmc_reduce.py records
# the commit; bench.py reads profiles/<round>_pmc_kernels*.json as roofline.traffic).
LEVEL=${LEVEL:-9}; ROUND=${ROUND:-r04}; SUF=""; if [ "$LEVEL" != "9" ]; then SUF="_L$LEVEL"; fi
NO27=""; if [ "$LEVEL" != "9" ]; then NO27="--no27"; fi
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; rm -rf gpurun_out/pmcK_FETCH* gpurun_out/pmcK_WRITE* gpurun_out/pmcY_*
timeout -k 10 300 python3 tools/pmc_kernels.py --time --level $LEVEL $NO27 > gpurun_out/pmcK_times.log 2>&1 || { tail -20 gpurun_out/pmcK_times.log; exit 1; }
cat gpurun_out/pmcK_times.log
for C in FETCH WRITE; do
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc ${C}_SIZE --output-format csv -d $R/gpurun_out/pmcK_$C -- python3 $R/tools/pmc_kernels.py --level $LEVEL $NO27 > $R/gpurun_out/pmcK_$C.log 2>&1 ) || { tail -20 gpurun_out/pmcK_$C.log; exit 1; }
done
python3 tools/pmc_reduce.py --level $LEVEL --out gpurun_out/${ROUND}_pmc_kernels$SUF.json || exit 1
for S in A B C D; do
CTRS=$(python3 tools/pmc_why.py sets | grep "^$S " | cut -d' ' -f2-)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $CTRS --output-format csv -d $R/gpurun_out/pmcY_$S -- python3 $R/tools/pmc_kernels.py --level $LEVEL $NO27 > $R/gpurun_out/pmcY_$S.log 2>&1 ) || { tail -20 gpurun_out/pmcY_$S.log; exit 1; }
echo "set $S done"
done
python3 tools/pmc_why.py --round ${ROUND}$SUF --level $LEVEL
