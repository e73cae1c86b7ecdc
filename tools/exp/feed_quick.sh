#!/bin/bash
# thinned-feed kernel: parity tests that touch it, then its time at C2 and C3 size (one window size)
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_variants.py tests/test_gpu_fullsize.py -x -q -m gpu -k "feed or thin" > gpurun_out/r3/t_feed.log 2>&1; tail -3 gpurun_out/r3/t_feed.log
for sz in "--snps 1000000 --inds 1000" "--snps 5000000 --inds 5000"; do
  for w in ${WS:-100}; do
    python tools/bench_variants.py $sz --winsize $w --modes feed --steps 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$sz W=$w', 'kernel ms', round(d['chain_kernel_ms'],3), 'call ms', round(d['call_ms'],2))"
  done
done
