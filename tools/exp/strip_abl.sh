#!/bin/bash
run() {
  r=$(env $2 timeout -k 10 120 python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize ${3:-100} --modes wlodgl --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
  echo "$1 $2 W=${3:-100} | wlodgl 2M x 1280: $r"
}
run "n5" GARLIC_X=1 50
run "n5" GARLIC_X=1 64
run "n5" GARLIC_X=1 80
