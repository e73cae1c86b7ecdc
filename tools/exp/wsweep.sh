#!/bin/bash
for W in 2 5 10 30 60 100 300; do
  timeout -k 10 300 python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes lod,tgls,feed,ld --steps 3 2>/dev/null | python3 -c "
import json,sys
out=[]
for ln in sys.stdin:
    d=json.loads(ln)
    out.append(d['mode']+':'+str(round(d.get('kernel_ms', d.get('call_ms',0)),2)))
print('W=$W', ' '.join(out))"
done
