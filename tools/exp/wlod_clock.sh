#!/bin/bash
# Is the FP64-bound wLOD kernel power / clock limited?  Samples rocm-smi (shader clock, socket power) while the kernel
# loops for ~10 s, with and without the score stores (build/abl/*.so from the write-out ablation).
cp garlic_amd/libgarlic_hip.so /tmp/shipped.so
for f in build/abl/e_new.so build/abl/f_new_nowrite.so; do
  cp $f garlic_amd/libgarlic_hip.so
  python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize ${W:-100} --modes wlod --steps 600 > /tmp/bv.json 2>/dev/null &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    echo "$(basename $f .so) $(date +%s.%N | cut -c1-14) $(rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E 'sclk|Socket' | sed 's/.*: //' | tr '\n' ' ')"
    sleep 0.3
  done
  wait $pid
  python3 -c "import json; [print('   ', d['mode'], round(d['kernel_ms'],2), round(d['roofline']['frac'],3)) for d in map(json.loads, open('/tmp/bv.json')) if d.get('mode')=='wlod']"
done
cp /tmp/shipped.so garlic_amd/libgarlic_hip.so
rocm-smi -d 0 --showclocks --showpower 2>/dev/null | head -30
