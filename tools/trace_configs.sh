#!/bin/bash
# Kernel traces of the given configs (no PMC passes): gpurun -- 'bash tools/trace_configs.sh TAG adm ml1m_b160 ...'
set -eo pipefail
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for cfg in "$@"; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/${cfg}_stats" -- python3 "$ROOT/tools/config_profile.py" $cfg > "$OUT/${cfg}_launches.txt" 2> "$OUT/${cfg}.err"
  python3 "$ROOT/tools/kernel_by_grid.py" "$OUT/${cfg}_stats" "$cfg" > "$OUT/${cfg}_kernels.txt"
  head -16 "$OUT/${cfg}_kernels.txt"
done
find "$OUT" -name "*.csv" -size +4M -delete || true
