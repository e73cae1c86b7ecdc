#!/bin/bash
for k in "$@"; do
  GARLIC_WLOD_PFW=$k python3 tools/gen_wlod_asm.py > /dev/null && make -s -C garlic_amd/csrc 2>&1 | grep -E "error"
  for v in 0 40000 54000; do
    r=$(GARLIC_WLOD_LDS_MIN=$v python3 tools/bench_variants.py --snps 2000000 --inds 1280 --modes wlod --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
    echo "PFW=$k LDS_MIN=$v | wlod 2M x 1280 W=100: $r"
  done
done
python3 -m pytest tests/test_gpu_variants.py -x -q -k wlod 2>&1 | tail -3
