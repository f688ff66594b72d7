#!/usr/bin/env python3
"""Pretty-prints the one-line JSON of bench.py (value, rates, per-kernel-class table)."""
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['value']} {d['unit']}  train {d.get('train_steps_per_s')}/s  sample {d.get('sample_steps_per_s')}/s  job {d.get('whole_job_tflops')} TF")
r = d.get("roofline") or {}
print(f"dominant: {r.get('kernel')}  {r.get('achieved')} {r.get('unit')}  frac {r.get('frac')}  avg {r.get('avg_launch_us')} us  traffic {r.get('traffic')}")
for k, v in (r.get("all_kernels") or {}).items():
    print(f"  {v['ms']:8.3f} ms {v['launches']:5d} x {v['ms'] * 1e3 / max(v['launches'], 1):7.2f} us {v['tflops']:7.2f} TF  {k}")
cb = d.get("cpu_baseline")
if cb:
    print(f"cpu_baseline {cb['value']:.2f} {cb['unit']} on {cb['cores']} cores ({cb['kind']}); x{d.get('speedup_vs_cpu_baseline')}")
