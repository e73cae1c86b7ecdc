#!/bin/bash
# Shader clock and socket power (rocm-smi) while one kernel of the path loops for ~8 s: which kernels run power-capped.
#   usage: clock_under_load.sh "MODE:W:STEPS ..."      MODE = lod | tgls | wlod | wlodgl (tools/bench_variants.py)
ls /sys/class/drm/ 2>/dev/null | head -3
for h in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "$h: $(cat $h/freq1_input 2>&1 | head -1) Hz, $(cat $h/power1_average 2>&1 | head -1) uW"; done 2>/dev/null | head -4
for spec in ${1:-"wlod:100:400"}; do
  IFS=: read mode W steps <<< "$spec"
  python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes $mode --steps $steps > /tmp/bv.json 2>/dev/null &
  pid=$!
  while kill -0 $pid 2>/dev/null; do
    echo "$mode W=$W $(rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E 'sclk|Socket' | sed 's/.*: //' | tr '\n' ' ')"
    sleep 0.3
  done | awk '{gsub(/[()Mhz]/,"",$3); if ($3+0 > 1500) {n++; c+=$3; p+=$4}} END {if (n) printf "%s %s: %d samples under load, sclk %.0f MHz, socket power %.0f W\n", $1, $2, n, c/n, p/n}'
  wait $pid
  python3 -c "import json; [print('   ', d['mode'], 'kernel_ms', round(d['kernel_ms'],2), 'frac', round(d['roofline']['frac'],3)) for d in map(json.loads, open('/tmp/bv.json')) if 'kernel_ms' in d]"
done
