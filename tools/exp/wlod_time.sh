#!/bin/bash
# wlod kernel times at a few shapes; args: env assignments for the run (e.g. GARLIC_WLOD_ONE_BLOCK=1)
for sz in "--snps 2000000 --inds 1280" "--snps 200000 --inds 1000" "--snps 10000000 --inds 1250"; do for w in 100 400; do
  if [ "$sz" = "--snps 10000000 --inds 1250" ] && [ $w = 400 ]; then continue; fi
  env "$@" python3 tools/bench_variants.py $sz --winsize $w --modes wlod --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['snps'], d['inds'], d['winsize'], round(d['kernel_ms'],3), round(d['roofline']['frac'],3))"
done; done
