#!/bin/bash
for cfg in "0 3" "4 3" "4 2" "2 3" "8 2"; do
  set -- $cfg
  GARLIC_WLOD_PFW=$1 GARLIC_WLOD_PFW_LOADS=$2 python3 tools/gen_wlod_asm.py > /dev/null && make -s -C garlic_amd/csrc 2>&1 | grep -E "error"
  for W in 50 100 200 400; do
    r=$(python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes wlod --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
    echo "PFW=$1 LOADS=$2 W=$W | wlod 2M x 1280: $r"
  done
done
