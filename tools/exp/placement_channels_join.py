#!/usr/bin/env python3
"""Groups rocprofv3 json counter records of lod_chain_kernel dispatches by score allocation and prints, per
counter, the per-instance (L2 channel x XCD) distribution: sum, max/mean, and the busiest instances."""
import json, glob, sys, collections
import numpy as np
PASSES = 6
compact = []
for d in sys.argv[1:]:
    fs = sorted(glob.glob(d + "/**/*_results.json", recursive=True))
    if not fs:
        print(d, "no json"); continue
    j = json.load(open(fs[-1]))["rocprofiler-sdk-tool"][0]
    names = {}
    for k in j.get("kernel_symbols", []):
        names[k["kernel_id"]] = k.get("formatted_kernel_name", k.get("kernel_name", ""))
    cnames = {}
    for c in j.get("counters", []):
        cnames[c["id"]["handle"] if isinstance(c["id"], dict) else c["id"]] = c["name"]
    disp = []
    for rec in j["callback_records"]["counter_collection"]:
        di = rec["dispatch_data"]["dispatch_info"]
        if "lod_chain_kernel" not in names.get(di["kernel_id"], ""):
            continue
        per = collections.defaultdict(list)
        for r in rec["records"]:
            cid = r["counter_id"]["handle"] if isinstance(r["counter_id"], dict) else r["counter_id"]
            per[cnames.get(cid, str(cid))].append(r["value"])
        t = rec["dispatch_data"]
        ms = (t.get("end_timestamp", 0) - t.get("start_timestamp", 0)) / 1e6
        disp.append((ms, per))
    print(d, len(disp), "chain dispatches")
    for g in range(0, len(disp), PASSES):
        grp = disp[g:g + PASSES][1:]          # first pass into a fresh buffer left out
        if not grp: continue
        ms = np.mean([x[0] for x in grp])
        line = "  alloc %d: ms %.3f" % (g // PASSES, ms)
        for cn in grp[0][1]:
            v = np.mean([np.array(x[1][cn], dtype=np.float64) for x in grp], axis=0)
            srt = np.sort(v)[::-1]
            line += "\n      %-36s n=%d sum %.4g  max/mean %.2f  min/mean %.2f  cv %.3f  top4 %s" % (
                cn, v.size, v.sum(), v.max() / max(v.mean(), 1e-30), v.min() / max(v.mean(), 1e-30),
                v.std() / max(v.mean(), 1e-30), np.array2string(srt[:4], precision=3))
        print(line)
        compact.append({"dir": d, "alloc": g // PASSES, "ms": float(ms),
                        "per_instance": {cn: np.mean([np.array(x[1][cn], dtype=np.float64) for x in grp], axis=0).tolist()
                                         for cn in grp[0][1]}})
json.dump(compact, open("gpurun_out/chan_compact.json", "w"))
