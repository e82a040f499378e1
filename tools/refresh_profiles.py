#!/usr/bin/env python3
"""Copies a tools/prof_r02.sh run (gpurun_out/<tag>_*) into the committed round-2 profiles:
   profiles/r02_pmc_summary.txt (header kept, counters replaced), profiles/pmc_traffic.json (tools/pmc_traffic.py),
   profiles/r02_kernel_stats.{csv,md} (qoa kernels' rows + the bench line of the same run).
   python tools/refresh_profiles.py r02g"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
os.chdir(ROOT)

old = open("profiles/r02_pmc_summary.txt").read()
body = open(f"gpurun_out/{tag}_pmc/summary.txt").read()
open("profiles/r02_pmc_summary.txt", "w").write(old[:old.index("== encode")] + body)
subprocess.check_call([sys.executable, "tools/pmc_traffic.py", f"gpurun_out/{tag}_pmc/summary.txt", "profiles/r02_pmc_summary.txt"],
                      stdout=subprocess.DEVNULL)

stats = glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_stats.csv", recursive=True)[0]
shutil.copy(stats, "profiles/r02_kernel_stats.csv")
rows = [r for r in csv.DictReader(open(stats)) if "qoa::" in r["Name"]]
line = open(f"gpurun_out/{tag}_bench.json").read().strip().splitlines()[-1]
json.loads(line)
md = open("profiles/r02_kernel_stats.md").read()
head = md[:md.index("| kernel |")]
tbl = "| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n" + "".join(
    f"| `{r['Name'][:80]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} |\n" for r in rows)
mid = md[md.index("\nThe decode leg enqueues"):md.index("```json")]
open("profiles/r02_kernel_stats.md", "w").write(head + tbl + mid + "```json\n" + line + "\n```\n")
d = json.loads(line)
enc = [r for r in rows if "encode" in r["Name"]][0]
print(f"K1 rocprof avg {float(enc['AverageNs']) / 1e6:.2f} ms; bench: {d['ms_per_step']:.2f} ms/step, traffic {d['roofline'].get('traffic')}, "
      f"decode {d['decode']['kernel_ms_avg']:.3f} ms frac {d['decode']['roofline']['frac']:.3f}, "
      f"full chip {d['k1_extras']['full_chip']['samples_per_s'] / 1e9:.1f} G samples/s ({d['k1_extras']['full_chip']['step_form']})")
