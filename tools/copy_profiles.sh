# after "ROUND=r04 bash tools/gpu_profiles.sh" on the GPU box: copy what profiles/README.md lists from gpurun_out/ into profiles/
ROUND=${ROUND:-r04}
for f in ab_small.json bench.json bench_kernel_stats.csv bench_smoother_only.json bench_smoother_only_kernel_stats.csv bench_under_rocprof.json \
         exa4_time.json exchange_faces.json fmg_timeline.txt pmc_kernels.json pmc_kernels_L8.json pmc_why.json vcycle_timeline.txt; do
  cp gpurun_out/${ROUND}_$f profiles/${ROUND}_$f
done
cp gpurun_out/${ROUND}_L8_pmc_why.json profiles/${ROUND}_pmc_why_L8.json
