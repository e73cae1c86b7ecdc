#!/bin/bash
timeout -k 10 300 python3 -m pytest tests/test_gpu_variants.py -x -q -k "wlod" 2>&1 | tail -5 || exit 1
for W in 100; do
  for e in "GARLIC_X=1" "GARLIC_WLOD_GL_NO_STRIP=1"; do
  r=$(env $e timeout -k 10 120 python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes wlodgl --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
  echo "$e W=$W | wlodgl 2M x 1280: $r"
  done
done
