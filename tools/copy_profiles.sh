#!/bin/bash
# Copies what tools/collect_profiles.sh / collect_config_profiles.sh left under gpurun_out/ into profiles/ (run in the build container):
#   bash tools/copy_profiles.sh r02
set -eo pipefail
TAG=${1:-r02}
O=gpurun_out/prof_$TAG; C=gpurun_out/cfgprof_$TAG
cp $O/bench.json profiles/${TAG}_bench.json
cp $O/bench_20_5.json profiles/${TAG}_bench_steps20_warmup5.json
cp $O/bench_under_rocprof.json profiles/${TAG}_bench_under_rocprof.json
cp "$(ls -t $O/stats/runc/*_kernel_stats.csv | head -1)" profiles/${TAG}_bench_kernel_stats.csv   # (the newest: an earlier collection of the round may have left its own)
for f in pmc_traffic.json pmc_traffic.txt pmc_sq.json pmc_sq.txt gemm_tile_sweep.txt; do cp $O/$f profiles/${TAG}_$f; done
python3 tools/kernel_by_grid.py "$(ls -t $O/stats/runc/*_kernel_trace.csv | head -1)" "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-other-configs: average duration per (kernel, grid size)" > profiles/${TAG}_bench_kernel_by_grid.txt
if [ -d "$C" ]; then
  for cfg in ml100k ml1m_b160 adm; do
    { cat $C/${cfg}_kernels.txt; echo; echo "# HBM traffic per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_summary.py)"; cat $C/${cfg}_pmc_traffic.txt; } > profiles/${TAG}_${cfg}_kernels_and_traffic.txt
  done
fi
