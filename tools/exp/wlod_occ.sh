#!/bin/bash
for v in 0 40000 54000 80000; do
  r=$(GARLIC_WLOD_LDS_MIN=$v python3 tools/bench_variants.py --snps 2000000 --inds 1280 --modes wlod --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
  echo "LDS_MIN=$v | wlod 2M x 1280 W=100: $r"
done
