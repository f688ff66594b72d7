#!/bin/bash
# one config, kernel trace, print lines matching a pattern: bash tools/trace_one.sh TAG CFG PATTERN
set -eo pipefail
TAG=$1; CFG=$2; PAT=$3
ROOT=$(pwd); OUT=$ROOT/gpurun_out/trace_$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/${CFG}_stats" -- python3 "$ROOT/tools/config_profile.py" $CFG > "$OUT/${CFG}_launches.txt" 2> "$OUT/${CFG}.err"
python3 "$ROOT/tools/kernel_by_grid.py" "$OUT/${CFG}_stats" "$CFG" | grep -E "$PAT" || true
rm -rf "$OUT/${CFG}_stats"
