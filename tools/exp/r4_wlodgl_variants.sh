#!/bin/bash
# times every build/abl/wlodgl_*.so (tools/exp/r4_build_wlod_variants.sh) with tools/exp/r4_wlodgl_time.py
O=gpurun_out/r4; mkdir -p $O
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in ${LIBS:-build/abl/wlodgl_*.so}; do
  cp $f garlic_amd/libgarlic_hip.so
  VARIANT=$(basename $f .so) timeout -k 10 300 python tools/exp/r4_wlodgl_time.py 2>> $O/wlodgl_${TAG:-a}.err | tee -a $O/wlodgl_${TAG:-a}.jsonl || break
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
