#!/bin/bash
# time every library in build/var/ on the LD-weights call (same box, same call)
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in build/var/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  r=$(python tools/bench_variants.py --modes ld --snps ${SNPS:-2000000} --inds ${INDS:-1280} --steps 3 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['call_ms'])")
  echo "$(basename $f) call_ms=$r"
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
