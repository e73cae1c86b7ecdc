#!/bin/bash
# per-kernel times of one LD call (garlic_panel_compute_ld) at the 10M x 1250 shard, for a list of window sizes
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/r3
for w in ${WS:-10 100}; do
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/ldprof_w$w -- python3 $R/tools/bench_variants.py --snps ${SNPS:-10000000} --inds ${INDS:-1250} --winsize $w --modes ld --steps 3 > $R/gpurun_out/r3/ldprof_w$w.log 2>&1
  cd $R
  python3 - $w <<'PY'
import csv, glob, sys, collections
w = sys.argv[1]
f = sorted(glob.glob(f"gpurun_out/r3/ldprof_w{w}/**/*_kernel_trace.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
tot = 0
print(f"W={w}: per LD call (4 calls profiled)")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    per_call = sum(v) / 4
    tot += per_call
    if per_call > 0.05: print(f"   {k:60s} {len(v)//4:4d} launches/call  {per_call:8.3f} ms/call")
print(f"   kernels total {tot:.2f} ms/call;", open(f"gpurun_out/r3/ldprof_w{w}.log").read().strip().splitlines()[-1][:200])
PY
done
