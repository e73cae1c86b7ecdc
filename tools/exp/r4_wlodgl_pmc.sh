#!/bin/bash
# SQ counters of the GL strip kernel (2M x 1280, W = 100): how busy the vector ALUs are, and the clock the run really had
# (GRBM_GUI_ACTIVE / kernel time).  One counter group per pass.
O=gpurun_out/r4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"; do
  tag=$(echo $grp | tr ' ' '+')
  STEPS=3 WS=100 timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $R/$O/pmc_$tag -- python3 $R/tools/exp/r4_wlodgl_time.py > $R/$O/pmc_$tag.out 2> $R/$O/pmc_$tag.err || { echo "failed $grp"; tail -3 $R/$O/pmc_$tag.err; }
  f=$(find $R/$O/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "wlod_strip" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k:28s} {v / max(n,1):.6g} per launch ({n} launches)")
PY
  rm -rf $R/$O/pmc_$tag
done
