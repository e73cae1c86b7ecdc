#!/bin/bash
# ablations of the GL-weighted wLOD loop (timing only; results are wrong under an ablation)
for v in "$@"; do
  env ${v//,/ } python3 tools/gen_wlod_asm.py > /dev/null && make -s -C garlic_amd/csrc 2>&1 | grep -E "error"
  r=$(python3 tools/bench_variants.py --snps 2000000 --inds 1280 --modes wlodgl --steps 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernel_ms'], round(d['roofline']['frac'],3))")
  echo "$v | wlodgl 2M x 1280 W=100: $r"
done
