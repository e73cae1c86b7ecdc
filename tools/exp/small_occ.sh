#!/bin/bash
for n in 4 6 8; do
touch garlic_amd/csrc/garlic_hip.hip; timeout 900 make -s -C garlic_amd/csrc EXTRA=-DGARLIC_SMALL_WAVES=$n 2>&1 | grep error
for W in 4 10 15; do timeout -k 10 200 python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes wlod,wlodgl --steps 3 2>/dev/null | python3 -c "
import json,sys
print(\"waves=$n W=$W\", [ (json.loads(l)[\"mode\"], round(json.loads(l)[\"kernel_ms\"],2)) for l in sys.stdin])"; done; done
