#!/bin/bash
# plain wLOD kernel time of library variants in build/abl (2M x 1280)
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for v in orig "$@"; do
  if [ $v != orig ]; then cp build/abl/$v.so garlic_amd/libgarlic_hip.so; else cp /tmp/orig.so garlic_amd/libgarlic_hip.so; fi
  for w in ${WS:-50 100 200}; do
    r=$(python tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $w --modes wlod --steps 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if d.get('mode')=='wlod': print(round(d['kernel_ms'],3), round(d['roofline']['frac'],3))")
    echo "$v W=$w wlod ms/frac $r"
  done
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
