"""mean FETCH_SIZE / WRITE_SIZE (KiB per launch) of every garlic:: kernel in a rocprofv3 --pmc counter_collection csv
   usage: pmc_kernels.py <dir> <counter>"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True))[-1]
acc = {}
for row in csv.DictReader(open(f)):
    if row["Counter_Name"] != sys.argv[2] or "garlic::" not in row["Kernel_Name"]:
        continue
    acc.setdefault(row["Kernel_Name"].split("(")[0], []).append(float(row["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(f"{k[:60]:60s} {sys.argv[2]:10s} launches {len(v):4d}  mean KiB per launch {sum(v) / len(v):14.1f}  max {max(v):14.1f}")
