#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and prints the decode kernels' start / end times relative to the first (do the general
kernel's side-stream launches really run beside K2u?): python tools/k2lab/trace_overlap.py <kernel_trace.csv> [last N]"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "qoa_decode" in r["Kernel_Name"]]
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -24:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = "K2u" if "uniform" in r["Kernel_Name"] else "K2 "
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{name} queue {r.get('Queue_Id', '?'):>3} start {s / 1e3:9.1f} us  end {e / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  grid {r.get('Grid_Size', '?')}")
