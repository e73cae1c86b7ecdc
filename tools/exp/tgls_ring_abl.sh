#!/bin/bash
# lod_chain_ring_kernel in bits mode (garlic_roh_coverage_fused with likelihoods): ring rows and loader rounds in flight
# (variants built into build/abl/ with -DGARLIC_TG_DEPTH=4 / -DGARLIC_TG_ABL_NOWAIT / _NOREADS / _NOCOMPUTE: none moves the call
# by more than 12 %: the bits mode reads 8 B of terms per window and runs at 0.8 of that stream's rate.  Do NOT build -DGARLIC_TG_TILE_ROWS=1 variants for this script: its first call is
# the score path, which then writes past its tiles and never finishes)
cp garlic_amd/libgarlic_hip.so /tmp/shipped.so
for f in build/abl/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  echo "== $(basename $f)"
  timeout -k 10 300 python3 tools/exp/tgls_cov_time.py 2>&1 | grep -E "fused|chain ms"
done
cp /tmp/shipped.so garlic_amd/libgarlic_hip.so
