#!/bin/bash
# lod_chain_ring_kernel in bits mode (garlic_roh_coverage_fused with likelihoods): ring rows and loader rounds in flight
# (variants built into build/abl/ with -DGARLIC_TG_DEPTH=: 4 rounds instead of 3 -> 7.04 against 7.16 ms at 2M x 1280, the
# loads are not what paces the bits mode.  Do NOT build -DGARLIC_TG_TILE_ROWS=1 variants for this script: its first call is
# the score path, which then writes past its tiles and never finishes)
cp garlic_amd/libgarlic_hip.so /tmp/shipped.so
for f in build/abl/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  echo "== $(basename $f)"
  timeout -k 10 300 python3 tools/exp/tgls_cov_time.py 2>&1 | grep -E "fused|chain ms"
done
cp /tmp/shipped.so garlic_amd/libgarlic_hip.so
