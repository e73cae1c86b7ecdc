#!/bin/bash
python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_ld.py -x -q 2>&1 | tail -3
for W in 50 100 200 400; do
  r=$(python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes wlod --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
  echo "W=$W | wlod 2M x 1280: $r"
done
