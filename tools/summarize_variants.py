#!/usr/bin/env python3
"""Condense gpurun_out/<tag>_variants* (tools/profile_variants.sh) into profiles/:
  <tag>_variants_kernel_stats.csv   the garlic kernels of the kernel-trace stats (torch's data-generation kernels dropped)
  <tag>_variants_bench.jsonl        the JSON lines tools/bench_variants.py printed, plainly and under the profiler
  <tag>_tgls_pmc_traffic.json       HBM traffic per launch of the TGLS chain kernel from the two PMC passes
usage: tools/summarize_variants.py [tag]      (default r02)"""
import csv
import glob
import json
import os
import sys

TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.environ.get("GARLIC_PROF_OUT") or os.path.join(ROOT, "profiles")


def newest(pattern):
    hits = sorted(glob.glob(os.path.join(OUT, pattern), recursive=True), key=os.path.getmtime)
    if not hits:
        sys.exit(f"missing {pattern} under gpurun_out/: run tools/profile_variants.sh {TAG} on the GPU box first")
    return hits[-1]


def lines(path):
    return [json.loads(l) for l in open(path) if l.startswith("{")]


def pmc(counter_dir, counter, kernel):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(newest(f"{counter_dir}/**/*_counter_collection.csv")))
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


def main():
    os.makedirs(PROF, exist_ok=True)
    plain, traced = lines(os.path.join(OUT, f"{TAG}_variants_plain.json")), lines(os.path.join(OUT, f"{TAG}_variants_bench.json"))
    shape = f"{plain[0]['snps']} SNPs x {plain[0]['inds']} individuals, W={plain[0]['winsize']}"
    rows = [r for r in csv.DictReader(open(newest(f"{TAG}_variants/**/*_kernel_stats.csv"))) if "garlic::" in r["Name"]]
    dst = os.path.join(PROF, f"{TAG}_variants_kernel_stats.csv")
    with open(dst, "w", newline="") as f:
        f.write(f'"# rocprofv3 --kernel-trace --stats -- python3 tools/bench_variants.py --modes ld,feed,lod,tgls,wlod,wlodgl --steps 3   ({shape}; tools/profile_variants.sh)"\n')
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs")])
    with open(os.path.join(PROF, f"{TAG}_variants_bench.jsonl"), "w") as f:
        for tag, ls in (("plain", plain), ("under rocprofv3 --kernel-trace", traced)):
            for l in ls:
                f.write(json.dumps(dict(l, run=tag)) + "\n")
    kern = "lod_chain_ring_kernel"
    fetch_kib, n1 = pmc(f"{TAG}_tgls_fetch", "FETCH_SIZE", kern)
    write_kib, n2 = pmc(f"{TAG}_tgls_write", "WRITE_SIZE", kern)
    tg = [l for l in plain if l["mode"] == "tgls"][0]
    win = tg["snps"] * tg["inds"]
    trace = [r for r in rows if kern in r["Name"]][0]
    doc = {
        "command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/bench_variants.py --modes tgls (tools/profile_variants.sh)",
        "workload": shape + ", TGLS --gl-type GQ", "kernel": "garlic::" + kern,
        "FETCH_SIZE_KiB_per_launch": fetch_kib, "WRITE_SIZE_KiB_per_launch": write_kib, "launches": [n1, n2],
        "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B (MI355X_MICROARCH.md, HBM) -> doubled; WRITE_SIZE exact",
        "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
        "terms_bytes_once": 8.0 * win, "scores_bytes": 8.0 * win, "algorithmic_bytes_per_launch": 16.25 * win,
        "fetch_over_terms_once": 2.0 * fetch_kib * 1024.0 / (8.0 * win),
        "kernel_trace_avg_ns": float(trace["AverageNs"]), "bench_kernel_ms_hip_events_plain_run": tg["kernel_ms"],
        "roofline_frac_plain_run": tg["roofline"]["frac"],
    }
    with open(os.path.join(PROF, f"{TAG}_tgls_pmc_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    # GL-weighted wLOD, strip form: the term rows enter the CUs once per strip
    kern2 = "wlod_strip_gl_kernel"
    if glob.glob(os.path.join(OUT, f"{TAG}_wlodgl_fetch/**/*_counter_collection.csv"), recursive=True):
        f2, n3 = pmc(f"{TAG}_wlodgl_fetch", "FETCH_SIZE", kern2)
        wg = [l for l in plain if l["mode"] == "wlodgl"][0]
        tr2 = [r for r in rows if kern2 in r["Name"]][0]
        doc2 = {
            "command": "rocprofv3 --pmc FETCH_SIZE -- python3 tools/bench_variants.py --modes wlodgl (tools/profile_variants.sh)",
            "workload": shape + ", --weighted with per-genotype likelihoods", "kernel": "garlic::" + kern2,
            "FETCH_SIZE_KiB_per_launch": f2, "launches": n3,
            "correction": "gfx950: FETCH_SIZE counts 128-B read requests as 64 B -> doubled",
            "fetch_bytes_per_launch": 2.0 * f2 * 1024.0,
            "terms_bytes_once": 8.0 * win, "weights_bytes_once": 8.0 * wg["snps"] * wg["winsize"],
            "tile_form_row_bytes_into_cus": 8.0 * win * (wg["winsize"] + 15) / 16.0,
            "fetch_over_terms_once": 2.0 * f2 * 1024.0 / (8.0 * win),
            "kernel_trace_avg_ns": float(tr2["AverageNs"]), "bench_kernel_ms_hip_events_plain_run": wg["kernel_ms"],
            "roofline_frac_plain_run": wg["roofline"]["frac"],
        }
        with open(os.path.join(PROF, f"{TAG}_wlodgl_pmc_traffic.json"), "w") as f:
            json.dump(doc2, f, indent=1)
        print(f"GL-weighted wLOD strip kernel: trace {float(tr2['AverageNs']) / 1e6:.2f} ms, fetch x2 = {doc2['fetch_over_terms_once']:.3f} x terms once")
    print(f"wrote {os.path.normpath(dst)}: {len(rows)} kernels; TGLS ring kernel: trace {float(trace['AverageNs']) / 1e6:.2f} ms, "
          f"fetch x2 = {doc['fetch_over_terms_once']:.3f} x terms once, HBM {doc['hbm_bytes_per_launch'] / 1e9:.1f} GB "
          f"vs algorithmic {doc['algorithmic_bytes_per_launch'] / 1e9:.1f} GB")


if __name__ == "__main__":
    main()
