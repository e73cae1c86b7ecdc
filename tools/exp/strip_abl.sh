#!/bin/bash
run() {
  r=$(env $2 timeout -k 10 120 python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize ${3:-100} --modes wlodgl --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
  echo "$1 $2 W=${3:-100} | wlodgl 2M x 1280: $r"
}
export GARLIC_WLOD_STRIP_MIN_W=32
for W in 40 50 60 70 80; do
run "auto" GARLIC_X=1 $W
run "n7" GARLIC_WLOD_STRIP_WAVES=7 $W
done
