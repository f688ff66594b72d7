#!/bin/bash
# Per-configuration evidence (SURVEY section 8d asks for the launch count and the per-kernel times of the latency-bound configs):
#   gpurun --timeout 1200 -- 'bash tools/collect_config_profiles.sh r02'
set -eo pipefail
TAG=${1:-r02}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/cfgprof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for cfg in ml100k ml1m_b160 adm; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${cfg}_stats" -- python3 "$ROOT/tools/config_profile.py" $cfg > "$OUT/${cfg}_launches.txt" 2> "$OUT/${cfg}.err"
  echo "$cfg stats done"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${cfg}_fetch" -- python3 "$ROOT/tools/config_profile.py" $cfg 20 > /dev/null 2>> "$OUT/${cfg}.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${cfg}_write" -- python3 "$ROOT/tools/config_profile.py" $cfg 20 > /dev/null 2>> "$OUT/${cfg}.err"
  echo "$cfg pmc done"
done
cd "$ROOT"
for cfg in ml100k ml1m_b160 adm; do
  { cat "$OUT/${cfg}_launches.txt"; python3 tools/kernel_by_grid.py "$OUT/${cfg}_stats" "rocprofv3 --kernel-trace -- python3 tools/config_profile.py $cfg (5 + 50 train steps, one full-resolution and one multi-resolution sampling call)"; } > "$OUT/${cfg}_kernels.txt"
  python3 tools/pmc_summary.py "$OUT/${cfg}_fetch" "$OUT/${cfg}_write" "$OUT/${cfg}_pmc_traffic.json" > "$OUT/${cfg}_pmc_traffic.txt"
done
find "$OUT" -name "*.csv" -size +4M -delete || true
ls "$OUT"
