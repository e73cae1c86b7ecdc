#!/usr/bin/env python3
import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "lod_chain_kernel" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        by[r["Dispatch_Id"]]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    disp = list(by.values())
    print(d, len(disp), "dispatches")
    for g in range(0, len(disp), 12):
        grp = disp[g:g + 12]
        keys = [k for k in grp[0] if k != "ms"]
        print("  alloc %d: ms %.3f  " % (g // 12, sum(x["ms"] for x in grp) / len(grp)) +
              "  ".join("%s %.4g" % (k, sum(x[k] for x in grp) / len(grp)) for k in keys))
