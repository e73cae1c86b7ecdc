#!/bin/bash
for k in 4; do
  GARLIC_WLOD_PFW=$k python3 tools/gen_wlod_asm.py > /dev/null && make -s -C garlic_amd/csrc 2>&1 | grep -E "error"
  for v in 0 33000 40000 48000 54000; do
    for W in 100 400; do
    r=$(GARLIC_WLOD_LDS_MIN=$v python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes wlod --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
    echo "PFW=$k LDS_MIN=$v W=$W | wlod 2M x 1280: $r"
    done
  done
done
