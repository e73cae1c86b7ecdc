#!/bin/bash
# what the plain two-block wLOD kernel spends its time on (2M x 1280): generator ablations + no write-out
run() { python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $1 --modes wlod --steps 3 2>/dev/null | python3 -c "import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if d.get('mode')=='wlod': print(round(d['kernel_ms'],2), round(d['roofline']['frac'],3))"; }
for w in 100; do echo "baseline W=$w: $(run $w)"; done
for v in nowait nosload; do
  GARLIC_WLOD_ABLATE=$v python3 tools/gen_wlod_asm.py > /dev/null && make -s -C garlic_amd/csrc 2>&1 | grep -E "error"
  for w in 100; do echo "ABLATE=$v W=$w: $(run $w)"; done
done
python3 tools/gen_wlod_asm.py > /dev/null && make -s -C garlic_amd/csrc EXTRA=-DGARLIC_WLOD_ABL_NO_WRITE 2>&1 | grep -E "error"
for w in 100; do echo "no write-out W=$w: $(run $w)"; done
