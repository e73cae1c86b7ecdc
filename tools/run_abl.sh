#!/bin/bash
# time each ablation library on the 64-individual panel (one item per CU, no HBM pressure)
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in build/abl/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  r=$(python bench.py --steps 5 --warmup 2 --no-cpu --inds ${INDS:-64} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])")
  echo "$(basename $f) $r"
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
