#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_r01.sh (gpurun_out/<tag>_*) into the small files
kept under profiles/: the kernel-trace stats, the per-launch HBM traffic of the garlic kernels from
the two PMC passes (FETCH_SIZE, WRITE_SIZE), and the bench line printed under the profiler.

usage: tools/summarize_profiles.py [tag]        (default tag r01)
"""
import csv
import glob
import json
import os
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.environ.get("GARLIC_PROF_OUT") or os.path.join(ROOT, "profiles")


def one(pattern):
    hits = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True), key=os.path.getmtime)   # newest run
    if not hits:
        sys.exit(f"missing {pattern} under gpurun_out/ -- run tools/profile_round.sh on the GPU box first")
    return hits[-1]


def pmc(counter_dir, counter):
    """mean counter value per launch for every garlic kernel (values are KiB for *_SIZE)"""
    acc = {}
    with open(one(f"{counter_dir}/**/*_counter_collection.csv")) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter or "garlic::" not in row["Kernel_Name"]:
                continue
            name = row["Kernel_Name"].split("(")[0]
            acc.setdefault(name, []).append(float(row["Counter_Value"]))
    return {k: {"launches": len(v), "mean_per_launch_KiB": sum(v) / len(v)} for k, v in acc.items()}


def main():
    os.makedirs(PROF, exist_ok=True)
    # 1. kernel-trace stats: garlic kernels in full, everything else (torch's synthetic-data
    #    kernels) folded into one line
    rows = list(csv.DictReader(open(one(f"{TAG}_trace/**/*_kernel_stats.csv"))))
    keep = [r for r in rows if "garlic::" in r["Name"]]
    other = [r for r in rows if "garlic::" not in r["Name"]]
    with open(os.path.join(PROF, f"{TAG}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in keep:
            w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage",
                                       "MinNs", "MaxNs", "StdDev")])
        w.writerow(["(torch kernels generating the synthetic panel: %d names)" % len(other),
                    sum(int(r["Calls"]) for r in other), sum(int(r["TotalDurationNs"]) for r in other),
                    "", "%.2f" % sum(float(r["Percentage"]) for r in other), "", "", ""])
    chain = [r for r in keep if "lod_chain_kernel" in r["Name"]][0]
    # the timed passes alone: bench.py first tries a few score buffers (placement changes the kernel time),
    # so the all-dispatch average above mixes buffers; the last `steps` dispatches are the timed region
    timed_avg_ns = float(chain["AverageNs"])
    trace_csv = glob.glob(os.path.join(OUT, f"{TAG}_trace/**/*_kernel_trace.csv"), recursive=True)
    if trace_csv:
        tr = [r for r in csv.DictReader(open(sorted(trace_csv, key=os.path.getmtime)[-1])) if "lod_chain_kernel" in r["Kernel_Name"]]
        tr.sort(key=lambda r: int(r["Start_Timestamp"]))
        bench_line = json.loads(open(os.path.join(OUT, f"{TAG}_trace_bench.json")).read().strip().splitlines()[-1])
        last = tr[-int(bench_line["steps"]):]
        timed_avg_ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / len(last)
        with open(os.path.join(PROF, f"{TAG}_kernel_stats.csv"), "a", newline="") as f:
            csv.writer(f).writerow(["garlic::lod_chain_kernel -- the %d dispatches of the timed region only" % len(last), len(last),
                                    int(timed_avg_ns * len(last)), "%.1f" % timed_avg_ns, "", "", "", ""])

    # 2. bench line printed by the traced run
    bench = json.loads(open(os.path.join(OUT, f"{TAG}_trace_bench.json")).read().strip().splitlines()[-1])
    with open(os.path.join(PROF, f"{TAG}_bench_under_rocprof.json"), "w") as f:
        json.dump(bench, f, indent=1)

    # 3. HBM traffic
    fetch = pmc(f"{TAG}_pmc_fetch", "FETCH_SIZE")
    write = pmc(f"{TAG}_pmc_write", "WRITE_SIZE")
    ck = [k for k in fetch if "lod_chain_kernel" in k][0]
    fetch_kib, write_kib = fetch[ck]["mean_per_launch_KiB"], write[ck]["mean_per_launch_KiB"]
    traffic = (2.0 * fetch_kib + write_kib) * 1024.0
    doc = {
        "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py (the arguments of tools/profile_round.sh, or of tools/profile_r01.sh for tag r01)",
        "workload": bench["config"]["workload"],
        "kernel": ck.replace("void ", ""),
        "FETCH_SIZE_KiB_per_launch": fetch_kib,
        "WRITE_SIZE_KiB_per_launch": write_kib,
        "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM) -> doubled; "
                      "WRITE_SIZE is exact for 16-B-per-lane streaming stores",
        "hbm_bytes_per_launch": traffic,
        "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
        "kernel_trace_avg_ns": timed_avg_ns,
        "kernel_trace_avg_ns_all_dispatches": float(chain["AverageNs"]),
        "bench_kernel_ms_hip_events": bench["roofline"]["kernel_ms"],
        "all_garlic_kernels": {k: {"FETCH_SIZE": fetch.get(k), "WRITE_SIZE": write.get(k)}
                               for k in sorted(set(fetch) | set(write))},
    }
    with open(os.path.join(PROF, f"{TAG}_pmc_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(f"chain kernel: trace avg over the timed region {timed_avg_ns / 1e6:.3f} ms, bench (HIP events) "
          f"{bench['roofline']['kernel_ms']:.3f} ms, HBM traffic {traffic / 1e9:.3f} GB "
          f"(fetch {2 * fetch_kib * 1024 / 1e9:.3f} + write {write_kib * 1024 / 1e9:.3f}), "
          f"algorithmic {doc['algorithmic_bytes_per_launch'] / 1e9:.3f} GB")


if __name__ == "__main__":
    main()
