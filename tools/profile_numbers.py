#!/usr/bin/env python3
"""Prints the figures the docs quote from the committed round profiles: python tools/profile_numbers.py r02"""
import csv, json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
d = json.load(open(f"profiles/{tag}_bench.json")); r = d["roofline"]
e = json.load(open(f"profiles/{tag}_bench_steps20_warmup5.json"))
print(f"value {d['value']}  20/5 {e['value']} (frac {e['roofline']['frac']})  ms/step {d['ms_per_step']}  windows {min(d['window_ms']):.1f}-{max(d['window_ms']):.1f} ms")
print(f"train {d['train_steps_per_s']}/s = {1e3 / d['train_steps_per_s']:.3f} ms  sample {d['sample_steps_per_s']}/s = {1e6 / d['sample_steps_per_s']:.1f} us  TF {d['whole_job_tflops']}  launches/step {d['launches_per_step']}")
print(f"dominant: {r['avg_launch_us']} us by events -> {r['achieved']} TF frac {r['frac']}; all classes bracketed {r['avg_launch_us_all_classes_bracketed']} us; traffic {r['traffic'] / 1e6:.1f} MB; hash {r['traffic_source']}")
c = d["cpu_baseline"]; print(f"cpu {c['value']:.1f} steps/s ({d['value'] / c['value']:.0f}x), {c['cores']} cores")
for k, v in d["other_configs"].items():
    print(k, "| train", v["train"]["step_us"], "us", v["train"]["launches_per_step"], "launches", v["train"]["steps_per_s"], "/s roof", v["train"]["roofline_us"], v["train"]["frac"],
          "| full", v["sample_full"]["step_us"], v["sample_full"]["steps_per_s"], v["sample_full"]["roofline_us"], v["sample_full"]["frac"], "| multires", v["sample_multires"]["step_us"])
FL = {"1, 1, 0, 1, 4>": 18.349e9, "0, 0, 0, 0, 3>": 5.682e9, "0, 0, 0, 0, 9>": 5.682e9, "0, 0, 1, 0, 1>": 5.682e9, "0, 0, 1, 0, 0>": 5.682e9}
for row in csv.DictReader(open(f"profiles/{tag}_bench_kernel_stats.csv")):
    n = row["Name"]
    if "at::" in n or "rocclr" in n: continue
    avg = float(row["AverageNs"]) / 1e3
    fl = next((v for k, v in FL.items() if k in n), None)
    if "0, 0, 0, 0, 10>" in n or ("0, 0, 0, 0, 1>" in n): fl = 1.2552e9
    print(f"{n[:46]:46s} ...{n[-34:]:34s} calls {row['Calls']:>5s} avg {avg:7.2f} us" + (f"  {fl / avg / 1e6:6.1f} TF" if fl else ""))
