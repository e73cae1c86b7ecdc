#!/bin/bash
run() {
  r=$(env $2 timeout -k 10 120 python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize ${3:-100} --modes wlodgl --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
  echo "$1 $2 W=${3:-100} | wlodgl 2M x 1280: $r"
}
timeout -k 10 600 python3 -m pytest tests/test_gpu_wlod_strip.py -x -q 2>&1 | tail -2
for W in 100 130 160 200 241; do
run "strip" GARLIC_X=1 $W
run "tile " GARLIC_WLOD_GL_NO_STRIP=1 $W
done
