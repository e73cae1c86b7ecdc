#!/bin/bash
# ld_hr2_tile_kernel: what its time is made of (variants built into build/abl/: no stores / no bit expansion)
export TMPDIR=/tmp
R=$PWD
cp garlic_amd/libgarlic_hip.so /tmp/shipped.so
for f in build/abl/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ldm_prof -- python3 $R/tools/bench_variants.py --snps 10000000 --inds 1250 --winsize 100 --modes ld --steps 3 > /tmp/ldm.log 2>&1
  cd $R
  python3 - $(basename $f) <<'PY'
import csv, glob, sys
f = sorted(glob.glob("/tmp/ldm_prof/**/*_kernel_trace.csv", recursive=True))[-1]
v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "ld_hr2_tile" in r["Kernel_Name"]]
print(sys.argv[1], "ld_hr2_tile_kernel ms per call:", round(sum(v) / 4, 2))
PY
  rm -rf /tmp/ldm_prof
done
cp /tmp/shipped.so garlic_amd/libgarlic_hip.so
