#!/bin/bash
# A/B of two builds of libsdrm_hip.so on the whole bench line, same box, alternating runs:
#   gpurun -- 'bash tools/ab_bench.sh tools/libsdrm_prev.so sdrm_amd/libsdrm_hip.so'
# (tools/libsdrm_prev.so: `git archive <commit> sdrm_amd/csrc include | tar -x -C /tmp/prev && hipcc ... -o tools/libsdrm_prev.so`)
A=$1; B=$2; ROUNDS=${3:-3}
for r in $(seq $ROUNDS); do
  for lib in "$A" "$B"; do
    SDRM_LIB=$lib python3 bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], 'steps/s  train', d['train_steps_per_s'], ' sample', d['sample_steps_per_s'], ' ms/step', d['ms_per_step'])"
  done
done
