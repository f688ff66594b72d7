#!/bin/bash
# Runs on the GPU box (gpurun): the bench line, the rocprofv3 kernel stats of the same command, the two PMC passes
# (HBM traffic) and the GEMM tile sweep.  Everything lands under gpurun_out/prof_$TAG/; copy what is to be judged
# into profiles/ afterwards (tools/pmc_summary.py turns the PMC passes into the per-kernel table).
set -eo pipefail
# usage (from the build container; GIT_HEAD is expanded there, the box has no .git):
#   gpurun --timeout 1200 -- "GIT_HEAD=$(git rev-parse --short HEAD) bash tools/collect_profiles.sh r02"
TAG=${1:-r02}
export GIT_HEAD=${GIT_HEAD:-unknown}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
: > "$OUT/bench.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$OUT/bench_20_5.json" 2>> "$OUT/bench.err"
echo "bench 20/5 done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-other-configs > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --steps 93 --warmup 5 --windows 1 > /dev/null 2> "$OUT/pmc_fetch.err"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --steps 93 --warmup 5 --windows 1 > /dev/null 2> "$OUT/pmc_write.err"
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-other-configs --steps 93 --warmup 5 --windows 1 > /dev/null 2> "$OUT/pmc_sq.err"
echo "sq done"
cd "$ROOT"
python3 tools/pmc_sq.py "$OUT/pmc_sq" "$OUT/pmc_sq.json" > "$OUT/pmc_sq.txt"
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.txt"
# the headline line last: it replays `traffic` / MFMA-busy from the PMC summaries just made (bench.py reads profiles/*pmc_*.json
# and drops them unless they carry the hash of the library it loaded)
cp "$OUT/pmc_traffic.json" "$ROOT/profiles/${TAG}_pmc_traffic.json"
cp "$OUT/pmc_sq.json" "$ROOT/profiles/${TAG}_pmc_sq.json"
python3 bench.py > "$OUT/bench.json" 2>> "$OUT/bench.err"
echo "bench done"
CFGS=0,1,2,4 python3 tools/gemm_tune.py > "$OUT/gemm_tile_sweep.txt" 2>&1
# keep the merge small: the per-dispatch PMC CSVs are large
find "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" -name "*.csv" -size +8M -delete || true
ls -la "$OUT"
