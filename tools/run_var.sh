#!/bin/bash
# time every library in build/var/ on the C2 bench (same box, same call): box-to-box variance is larger than most effects
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in build/var/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  for k in ${WORKERS:-256}; do
  r=$(GARLIC_WORKERS=$k python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu --inds ${INDS:-1000} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'])")
  echo "$(basename $f) K=$k $r"
  done
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
