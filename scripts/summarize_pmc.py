#!/usr/bin/env python3
"""Summarise every rocprofv3 output directory under <dir>: trace_* -> mean / min / median duration per
kernel (steady state: the last two thirds of the launches; mean_all_us: all of them), pmc_* -> mean counter value per launch
and kernel.  Prints one JSON object."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict

root = sys.argv[1]
KEYS = ("csr_vector_window_kernel", "csr_vector_sweep_kernel", "csr_vector_kernel", "light_rows_window_kernel",
        "light_rows_sweep_kernel", "light_rows_kernel", "merge_rows_kernel", "merge_tile_kernel", "merge_search_kernel", "merge_fixup_kernel",
        "merge_small_kernel", "giant_")


def short(name):
    n = name.split("(")[0]
    for k in KEYS:
        if k in n:
            return k
    return None


out = {}
for d in sorted(glob.glob(os.path.join(root, "trace_*"))):
    if not os.path.isdir(d):
        continue
    fs = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not fs:
        continue
    dur = defaultdict(list)
    for row in csv.DictReader(open(fs[0])):
        k = short(row["Kernel_Name"])
        if k:
            dur[k].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    # mean_us: over the last two thirds of the launches (in trace order) — an HBM-bound kernel runs its first ~60 executes
    # from a fresh plan 5-10 % slower (profiles/r02_warmup_curve.txt); mean_all_us: every launch
    def stats(v):
        w = v[len(v) // 3:] if len(v) >= 6 else v
        return {"launches": len(v), "mean_us": sum(w) / len(w), "mean_all_us": sum(v) / len(v), "min_us": min(v),
                "median_us": statistics.median(w)}
    out[os.path.basename(d)] = {k: stats(v) for k, v in dur.items()}
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        k = short(row["Kernel_Name"])
        if k:
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out[os.path.basename(d)] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1, sort_keys=True))
