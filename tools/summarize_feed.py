#!/usr/bin/env python3
"""Condense gpurun_out/<tag>_feed* (tools/profile_feed.sh) into profiles/<tag>_feed_*:
  <tag>_feed_kernel_stats.csv    garlic kernels of the kernel trace: four window sizes at 5M x 5000, single calls and one multi call
  <tag>_feed_pmc.json            HBM traffic per launch (FETCH_SIZE x 2 + WRITE_SIZE) and the SQ instruction mix at W = 100
usage: tools/summarize_feed.py [tag]"""
import csv, glob, json, os, sys, collections

TAG = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.environ.get("GARLIC_PROF_OUT") or os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)


def newest(pattern):
    hits = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True), key=os.path.getmtime)
    if not hits:
        sys.exit(f"missing {pattern} under gpurun_out/: run tools/profile_feed.sh {TAG} on the GPU box first")
    return hits[-1]


def counters(d):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(newest(f"{d}/**/*_counter_collection.csv"))):
        if "lod_feed_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, max(len(v) for v in acc.values())


def main():
    rows = [r for r in csv.DictReader(open(newest(f"{TAG}_feed_trace/**/*_kernel_stats.csv"))) if "garlic::" in r["Name"]]
    with open(os.path.join(PROF, f"{TAG}_feed_kernel_stats.csv"), "w", newline="") as f:
        f.write('"# rocprofv3 --kernel-trace --stats -- python3 tools/exp/feed_multi_time.py   (5M SNPs x 5000 individuals, W = 50 100 200 300: '
                'two rounds of four garlic_lod_feed calls, then four garlic_lod_feed_multi calls; tools/profile_feed.sh)"\n')
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs")])
    tr = [r for r in csv.DictReader(open(newest(f"{TAG}_feed_trace/**/*_kernel_trace.csv"))) if "lod_feed_kernel" in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last multi call: its four kernels and the span they cover together
    last = tr[-4:]
    span_ms = (max(int(r["End_Timestamp"]) for r in last) - min(int(r["Start_Timestamp"]) for r in last)) / 1e6
    each_ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in last]
    fetch, n1 = counters(f"{TAG}_feed_fetch")
    write, n2 = counters(f"{TAG}_feed_write")
    sq = {}
    for sub in ("a", "b"):
        c, _ = counters(f"{TAG}_feed_sq/{sub}")
        sq.update(c)
    win = 5_000_000 * 5000
    windows_waves = win / 64.0
    plain = [json.loads(l) for l in open(os.path.join(OUT, f"{TAG}_feed_plain.json")) if l.startswith("{")][-1]
    doc = {
        "command": "tools/profile_feed.sh: rocprofv3 --kernel-trace / --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc SQ_* (separate passes)",
        "workload": "5M SNPs x 5000 individuals; PMC passes: --winsize 100, thinning step 100", "kernel": "garlic::lod_feed_kernel",
        "multi_call_last": {"kernel_ms_each": each_ms, "span_ms_of_the_four": span_ms, "sum_ms": sum(each_ms)},
        "plain_run": plain,
        "FETCH_SIZE_KiB_per_launch": fetch.get("FETCH_SIZE"), "WRITE_SIZE_KiB_per_launch": write.get("WRITE_SIZE"), "launches": [n1, n2],
        "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact",
        "hbm_bytes_per_launch": (2.0 * fetch.get("FETCH_SIZE", 0) + write.get("WRITE_SIZE", 0)) * 1024.0,
        "algorithmic_bytes_per_launch": (0.25 + 8.0 / 100) * win,
        "sq_counters_per_launch": sq,
        "instructions_per_window_and_wave": {k: sq[k] / windows_waves for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_SALU",
                                                                               "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_VMEM") if k in sq},
    }
    with open(os.path.join(PROF, f"{TAG}_feed_pmc.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("multi_call_last", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "instructions_per_window_and_wave")}, indent=1))


if __name__ == "__main__":
    main()
