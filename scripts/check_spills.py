#!/usr/bin/env python3
"""hipcc -Rpass-analysis=kernel-resource-usage over every translation unit of the library: one line per kernel
(VGPRs, SGPRs, scratch bytes per lane, occupancy, static LDS); exit status 1 if any kernel uses scratch."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "spmv-samples_amd", "csrc")
TUS = ["csr_vector.hip", "csr_vector_f64.hip", "light_rows.hip", "light_rows_f64.hip", "merge_path.hip", "merge_path_f64.hip",
       "merge_path_i32.hip", "analyze.hip",
       "dist.hip"]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def analyse(tu):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-function",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(SRC, tu), "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"tu": tu, "name": m.group(1)}
            rows.append(cur)
            continue
        if cur is None:
            continue
        for key, pat in (("sgpr", r"SGPRs: (\d+)"), ("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m:
                cur[key] = int(m.group(1))
    return rows


def short(name):
    name = re.sub(r"^void mi355::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else None
    with ThreadPoolExecutor(max_workers=4) as ex:
        rows = [r for rs in ex.map(analyse, TUS) for r in rs]
    rows = [r for r in rows if "vgpr" in r]
    names = demangle([r["name"] for r in rows])
    lines = []
    bad = 0
    for r, n in sorted(zip(rows, names), key=lambda t: (t[0]["tu"], t[1])):
        flag = ""
        if r.get("scratch", 0) > 0:
            bad += 1
            flag = "   <-- SCRATCH"
        lines.append("%-20s %-110s vgpr %3d sgpr %3d scratch %4d occ %d lds %6d%s" % (
            r["tu"], short(n)[:110], r["vgpr"], r.get("sgpr", 0), r.get("scratch", 0), r.get("occ", 0), r.get("lds", 0), flag))
    text = "\n".join(lines) + "\n%d kernels, %d with scratch\n" % (len(lines), bad)
    sys.stdout.write(text)
    if tag:
        with open(os.path.join(ROOT, "profiles", "%s_kernel_resources.txt" % tag), "w") as f:
            f.write("# scripts/check_spills.sh %s  (hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage)\n" % tag)
            f.write(text)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
