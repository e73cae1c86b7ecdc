#!/bin/bash
# HBM traffic of the narrow-window wLOD kernels (W = 10, 2M x 1280): tools/exp/r4_small_time.py under FETCH_SIZE / WRITE_SIZE
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/exp/r4_small_time.py 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d /tmp/sp_$c -- python3 $R/tools/exp/r4_small_time.py > /dev/null 2> $O/small_pmc_$c.err || { echo failed $c; tail -3 $O/small_pmc_$c.err; }
  python3 $R/tools/exp/pmc_kernels.py /tmp/sp_$c $c | grep -i "small\|wlod"
done
