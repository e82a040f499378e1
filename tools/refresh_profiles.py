#!/usr/bin/env python3
"""Copies a tools/prof_round.sh run (gpurun_out/<tag>_*) into the committed profiles of a round:
   profiles/<round>_pmc_summary.txt (header kept when the file exists, counters replaced), profiles/pmc_traffic.json
   (tools/pmc_traffic.py), profiles/<round>_kernel_stats.{csv,md} (the qoa kernels' rows + the bench line of the same run).
   python tools/refresh_profiles.py r03 r03a"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
os.chdir(ROOT)

summary = f"profiles/{rnd}_pmc_summary.txt"
head = ""
if os.path.exists(summary):
    old = open(summary).read()
    head = old[:old.index("== encode")] if "== encode" in old else old
body = open(f"gpurun_out/{tag}_pmc/summary.txt").read()
open(summary, "w").write(head + body)
subprocess.check_call([sys.executable, "tools/pmc_traffic.py", f"gpurun_out/{tag}_pmc/summary.txt", summary], stdout=subprocess.DEVNULL)
mem = f"gpurun_out/{tag}_pmc_mem/summary.txt"
if os.path.exists(mem):
    shutil.copy(mem, f"profiles/{rnd}_pmc_mem_path.txt")

stats = glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, f"profiles/{rnd}_kernel_stats.csv")
rows = [r for r in csv.DictReader(open(stats)) if "qoa::" in r["Name"]]
sys.path.insert(0, os.path.join(ROOT, "tools"))
from gen_results_md import load  # noqa: E402  (the full record: the bench_extras line)
line = open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1]   # the contract's line (<= 6 KB), quoted below
d = load(f"gpurun_out/{tag}_bench.json")
td = load(f"gpurun_out/{tag}_trace_bench.json")
md = (f"# {rnd} -- rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 3 --warmup 1 --no-cpu --no-extras` (tools/prof_round.sh {tag})\n\n"
      "The kernel averages below are rocprof's over every launch of that command (warm-up and counted-plan launches included);\n"
      f"bench.py's own HIP-event figures in the same process: K1 {td['roofline']['kernel_ms_avg']:.3f} ms, "
      f"decode {td['decode']['kernel_ms_avg']:.4f} ms.\n\n"
      "| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n" + "".join(
          f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |\n" for r in rows)
      + "\nThe bench line of a plain run on the same box (`python3 bench.py --steps 5 --warmup 1`, no profiler):\n\n```json\n" + line + "\n```\n")
open(f"profiles/{rnd}_kernel_stats.md", "w").write(md)
enc = [r for r in rows if "encode" in r["Name"]][0]
print(f"K1 rocprof avg {float(enc['AverageNs']) / 1e6:.2f} ms; bench: {d['ms_per_step']:.2f} ms/step, traffic {d['roofline'].get('traffic')}, "
      f"decode {d['decode']['kernel_ms_avg']:.3f} ms frac {d['decode']['roofline']['frac']:.3f}, "
      f"full chip {d['k1_extras']['full_chip']['samples_per_s'] / 1e9:.1f} G samples/s")
