#!/bin/bash
# the counts inside lod_bits_kernel's queue: what the overlap costs the chains (variants built into build/abl/)
R=$PWD
cp garlic_amd/libgarlic_hip.so /tmp/shipped.so
for f in build/abl/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  echo "== $(basename $f)"
  timeout -k 10 300 python3 tools/exp/cov_overlap_time.py 2>&1 | grep -v amdgpu.ids
done
cp /tmp/shipped.so garlic_amd/libgarlic_hip.so
