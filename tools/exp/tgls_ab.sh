#!/bin/bash
# A/B of the TGLS chain kernels on one box: ring (persistent) vs the resident-items kernel
python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_tgls_continuous.py -m gpu -x -q 2>&1 | tail -3
for sz in "--snps 200000 --inds 1000" "--snps 2000000 --inds 1280" "--snps 10000000 --inds 1250"; do
  for w in 100 50 200; do
    echo "== $sz W=$w ring"; python3 tools/bench_variants.py $sz --winsize $w --modes tgls --steps 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['roofline']['frac'])"
    echo "== $sz W=$w old";  GARLIC_TGLS_NO_RING=1 python3 tools/bench_variants.py $sz --winsize $w --modes tgls --steps 5 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], d['roofline']['frac'])"
  done
done
