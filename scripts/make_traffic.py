#!/usr/bin/env python3
"""profiles/traffic_latest.json from one run of scripts/gpu_profiles.sh:
    python scripts/make_traffic.py gpurun_out/r03_prof profiles/r03_traces <commit>
Copies the small files of the run (summary.json, the per-trace kernel_stats tables, the bench lines, the vendor column)
to the profiles/ directory and writes the entries bench.py may quote as roofline.traffic: per kernel and workload,
hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — rocprofv3 reports KiB, and on gfx950 FETCH_SIZE counts 64 B
per 128-B request of a wide streaming read (MI355X_MICROARCH.md, HBM) — from PMC passes taken in their own runs."""
import glob
import json
import os
import shutil
import sys

src, dst, commit = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
for f in glob.glob(os.path.join(src, "*")):
    if os.path.isfile(f) and (f.endswith("_kernel_stats.csv") or os.path.basename(f) in ("summary.json", "vendor_rocsparse_same_box.txt")
                              or os.path.basename(f).startswith("bench_") and f.endswith(".json")):
        shutil.copy(f, dst)
summary = json.load(open(os.path.join(src, "summary.json")))


def bench_line(tag):
    p = os.path.join(src, tag + ".json")
    try:
        return json.loads(open(p).read().strip().splitlines()[-1])
    except Exception:
        return None


entries = []
PAIRS = [("s32_vector", "csr_vector_window_kernel"), ("s32_light", "light_rows_window_kernel"), ("s32_merge", "merge_rows_kernel"),
         ("wide32k_vector", "csr_vector_sweep_kernel"), ("c4_vector", "csr_vector_window_kernel")]
for tag, kernel in PAIRS:
    fetch = summary.get("pmc_%s_fetch" % tag, {}).get(kernel, {}).get("FETCH_SIZE")
    write = summary.get("pmc_%s_write" % tag, {}).get(kernel, {}).get("WRITE_SIZE")
    line = bench_line("trace_" + tag)
    trace = summary.get("trace_" + tag, {}).get(kernel)
    if fetch is None or write is None or line is None:
        continue
    alg = int(line["roofline"]["algorithmic_bytes"])
    hbm = (2.0 * fetch + write) * 1024.0
    entries.append({"kernel": kernel, "workload": line["config"]["workload"], "algorithmic_bytes": alg,
                    "hbm_bytes_per_launch": hbm, "ratio_to_algorithmic": hbm / alg,
                    "mean_us": trace["mean_us"] if trace else None,
                    "source": os.path.join(dst, "summary.json"), "commit": "PMC taken at " + commit})
out = {"note": "PMC summaries (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, scripts/gpu_profiles.sh -> summarize_pmc.py "
               "-> make_traffic.py) that bench.py may quote as roofline.traffic; matched on kernel AND algorithmic bytes. "
               "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 half-count of wide streaming reads). bench.py reports "
               "the entry's source + commit as roofline.traffic_source: a recorded figure, not one measured in the bench run.",
       "entries": entries}
json.dump(out, open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic_latest.json"), "w"), indent=1)
for e in entries:
    print("%-28s %-60s %.4fx  %s us" % (e["kernel"], e["workload"][:60], e["ratio_to_algorithmic"], e["mean_us"]))
