#!/bin/bash
# times every build/abl/feed_*.so: the four C3 feeds (tools/exp/r4_feed_time.py) and the final pass at 10M x 1250
# (tools/exp/roh_segments_time.py)
O=gpurun_out/r4; mkdir -p $O
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in ${LIBS:-build/abl/feed_*.so}; do
  cp $f garlic_amd/libgarlic_hip.so
  echo "== $(basename $f .so)" | tee -a $O/feedv_${TAG:-a}.txt
  timeout -k 10 300 python tools/exp/r4_feed_time.py 2>> $O/feedv_${TAG:-a}.err | tee -a $O/feedv_${TAG:-a}.txt || break
  [ -n "$NOSEG" ] || timeout -k 10 300 python tools/exp/roh_segments_time.py 2>> $O/feedv_${TAG:-a}.err | tail -4 | tee -a $O/feedv_${TAG:-a}.txt || break
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
