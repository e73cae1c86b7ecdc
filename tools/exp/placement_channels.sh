#!/bin/bash
# Per-channel view of the score-buffer placement effect: the TCC write/read request and
# DRAM-credit-stall counters WITHOUT the _sum reduction, so every L2 channel instance of every XCD is a
# record of its own (json output keeps the dimensions).  tools/exp/placement_channels_join.py groups them.
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/chan
mkdir -p $OUT
rocprofv3 -L > $OUT/avail.txt 2>&1
for set in "TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL" "TCC_EA0_RDREQ TCC_EA0_WRREQ_LEVEL" "TCC_REQ TCC_TAG_STALL"; do
  tag=$(echo $set | tr ' ' '+')
  rocprofv3 --pmc $set --output-format json csv -d $OUT/$tag -- python3 tools/exp/variance_pmc.py 6 > $OUT/$tag.log 2>&1 || echo "set $tag failed" >> $OUT/failed.txt
done
python3 tools/exp/placement_channels_join.py $OUT/*/ > $OUT/join.txt 2>&1
cp $OUT/join.txt $PWD/gpurun_out/chan_join.txt
rm -rf $OUT
tail -5 $PWD/gpurun_out/chan_join.txt
