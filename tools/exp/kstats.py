"""print the garlic:: rows of a rocprofv3 kernel_stats csv: name, calls, average ms"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if "garlic::" in n or "fillBuffer" in n or "copyBuffer" in n:
        print(f'{n.split("(")[0][:60]:60s} calls {r["Calls"]:>5s}  avg ms {float(r["AverageNs"]) / 1e6:9.3f}  max ms {float(r["MaxNs"]) / 1e6:9.3f}')
