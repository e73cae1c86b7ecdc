#!/bin/bash
# LDS counters of the thinned-feed chain (5M x 5k, four sizes): is the CU's LDS what the chains share?
O=gpurun_out/r4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS"; do
  tag=$(echo $grp | tr ' ' '+')
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $R/$O/fpmc_$tag -- python3 $R/tools/exp/r4_feed_time.py > $R/$O/fpmc_$tag.out 2> $R/$O/fpmc_$tag.err || { echo "failed $grp"; tail -3 $R/$O/fpmc_$tag.err; }
  f=$(find $R/$O/fpmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    if "lod_feed_kernel" in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
for k, (v, n) in acc.items():
    print(f"{k:28s} {v / max(n,1):.6g} per launch ({n} launches)")
PY
  rm -rf $R/$O/fpmc_$tag
done
