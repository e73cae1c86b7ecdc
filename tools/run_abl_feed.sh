#!/bin/bash
# time each ablation library on the thinned-feed path (chain kernel with the thinned write-out)
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in build/abl/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  r=$(python tools/bench_variants.py --snps 1000000 --inds ${INDS:-1000} --modes feed --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['chain_kernel_ms'])")
  echo "$(basename $f) $r"
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
