#!/bin/bash
timeout -k 10 600 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_tgls_continuous.py tests/test_gpu_host_tool.py -x -q 2>&1 | tail -3
for m in wlod wlodgl; do for W in 2 4 8 10 15 16; do
  r=$(timeout -k 10 200 python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes $m --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3), round(d['out_GBps']))")
  echo "$m W=$W | 2M x 1280: $r"
done; done
timeout -k 10 900 python3 tools/soak.py --trials 120 --seed 5 2>&1 | tail -2
