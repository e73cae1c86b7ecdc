#!/bin/bash
# write-out of the two-block wLOD kernel: variants built by build_wlod_wo_abl.sh into build/abl/, timed at 2M x 1280
run() { python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $1 --modes wlod --steps 3 2>/dev/null | python3 -c "import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if d.get('mode')=='wlod': print(round(d['kernel_ms'],2), round(d['roofline']['frac'],3))"; }
cp garlic_amd/libgarlic_hip.so /tmp/shipped.so
for f in build/abl/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  for w in ${WS:-100}; do echo "$(basename $f) W=$w: $(run $w)"; done
done
cp /tmp/shipped.so garlic_amd/libgarlic_hip.so
