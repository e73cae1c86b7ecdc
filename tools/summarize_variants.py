#!/usr/bin/env python3
"""Condense gpurun_out/r01_variants (tools/profile_variants.sh) into profiles/r01_variants_kernel_stats.csv:
the garlic kernels of the kernel-trace stats, torch's data-generation kernels dropped."""
import csv
import glob
import os

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "r01_variants", "**", "*_kernel_stats.csv"), recursive=True),
             key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(src)) if "garlic::" in r["Name"]]
dst = os.path.join(ROOT, "profiles", "r01_variants_kernel_stats.csv")
with open(dst, "w", newline="") as f:
    f.write('"# rocprofv3 --kernel-trace --stats -- python3 tools/bench_variants.py --modes ld,feed,lod,tgls,wlod,wlodgl '
            '--steps 5   (200k SNPs x 1000 individuals, W=100; tools/profile_variants.sh)"\n')
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs")])
print(f"wrote {os.path.normpath(dst)}: {len(rows)} kernels")
