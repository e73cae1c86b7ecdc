#!/bin/bash
# time every library in build/var/ on the variants bench (same box, same call)
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in build/var/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  r=$(python tools/bench_variants.py --modes ${MODES:-wlod} --snps ${SNPS:-200000} --winsize ${WIN:-100} 2>/dev/null | python -c "
import json,sys
print(' '.join('%s=%.3fms' % (d['mode'], d['kernel_ms']) for d in map(json.loads, sys.stdin.read().strip().splitlines())))")
  echo "$(basename $f) $r"
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
