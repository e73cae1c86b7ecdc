#!/bin/bash
timeout -k 10 900 python3 -m pytest tests/test_gpu_ld.py tests/test_gpu_host_tool.py -x -q 2>&1 | tail -3
for e in GARLIC_X=1 GARLIC_LD_HR2_PLAIN=1; do
  env $e timeout -k 10 300 python3 tools/bench_variants.py --snps 10000000 --inds 1250 --winsize 100 --modes ld --steps 3 2>/dev/null | python3 -c "
import json,sys
for ln in sys.stdin:
    d=json.loads(ln); print('$e', {k:(round(v,2) if isinstance(v,float) else v) for k,v in d.items() if k!='roofline'})"
done
