#!/bin/bash
# usage: tools/exp/run_variants.sh "ENV1=.. ENV2=.." "ENV.." ...   -> regenerate chain asm with that env, rebuild, run variance.py
for v in "$@"; do
  env $v python3 tools/gen_chain_asm.py > /dev/null && make -s -C garlic_amd/csrc 2>&1 | grep -E "error" 
  python3 tools/exp/variance.py 2>/dev/null | python3 -c "
import sys,json
ks=[]
for l in sys.stdin:
    d=json.loads(l)
    if d['rep']==0: ks.append(round(d['k_mean'],3))
print('$v', ks)
"
done
