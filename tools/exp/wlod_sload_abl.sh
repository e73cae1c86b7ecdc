#!/bin/bash
# What four blocks per wave could buy the plain wLOD kernel: the two-block loop with every other scalar weight load
# left out (GARLIC_WLOD_ABLATE=halfsload: the loads per flop of a four-block loop at the two-block loop's occupancy),
# with none (nosload), without the look-up address arithmetic (noint).  Results are wrong under an ablation: timing only.
cp garlic_amd/libgarlic_hip.so /tmp/shipped.so
for f in build/abl/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize 100 --modes wlod --steps 300 > /tmp/bv.json 2>/dev/null &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    echo "$(rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E 'sclk|Socket' | sed 's/.*: //' | tr '\n' ' ')"
    sleep 0.3
  done | awk -v tag=$(basename $f .so) '{gsub(/[()Mhz]/,"",$1); if ($1+0 > 1500) {n++; c+=$1; p+=$2}} END {if (n) printf "%s: sclk %.0f MHz, %.0f W; ", tag, c/n, p/n}'
  wait $pid
  python3 -c "import json; [print('kernel_ms', round(d['kernel_ms'],2), 'frac', round(d['roofline']['frac'],3)) for d in map(json.loads, open('/tmp/bv.json')) if 'kernel_ms' in d]"
done
cp /tmp/shipped.so garlic_amd/libgarlic_hip.so
