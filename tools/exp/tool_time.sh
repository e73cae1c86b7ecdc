#!/bin/bash
# wall time of the host tool's flows on a synthetic TPED (timing only)
set -e
D=/tmp/gt; mkdir -p $D
python3 - <<'PY'
import numpy as np
rng=np.random.default_rng(5)
n,nind=100000,200
pos=np.cumsum(rng.integers(500,3000,size=n))
with open('/tmp/gt/x.tped','w') as f:
    g=rng.integers(0,2,size=(n,2*nind))
    for l in range(n):
        f.write(f"1 rs{l} {pos[l]*1e-6:.6f} {pos[l]} "+" ".join("AG"[x] for x in g[l])+"\n")
with open('/tmp/gt/x.tfam','w') as f:
    for i in range(nind): f.write(f"pop1 ind{i} 0 0 0 0\n")
with open('/tmp/gt/x.map','w') as f:
    for l in range(0,n,50): f.write(f"1 rs{l} {pos[l]*1.1e-6:.8f} {pos[l]}\n")
PY
T=garlic_amd/host/garlic-lod
run() { local s=$(date +%s%N); "$@" > $D/out.log 2>&1 || { tail -3 $D/out.log; }; local e=$(date +%s%N); echo "$(( (e - s) / 1000000 )) ms : ${*:10}"; }
run $T --tped $D/x.tped --tfam $D/x.tfam --build hg19 --error 0.001 --winsize 10 --out $D/o1
run $T --tped $D/x.tped --tfam $D/x.tfam --build hg19 --error 0.001 --winsize 10 --raw-lod --out $D/o2
run $T --tped $D/x.tped --tfam $D/x.tfam --build hg19 --error 0.001 --winsize 10 --weighted --map $D/x.map --out $D/o3
run $T --tped $D/x.tped --tfam $D/x.tfam --build hg19 --error 0.001 --winsize-multi 10 20 30 40 50 --out $D/o4
run $T --tped $D/x.tped --tfam $D/x.tfam --build hg19 --error 0.001 --winsize-multi 10 20 30 40 50 --weighted --map $D/x.map --out $D/o5
run $T --tped $D/x.tped --tfam $D/x.tfam --build hg19 --error 0.001 --winsize 100 --weighted --map $D/x.map --out $D/o6
