#!/bin/bash
# as run_var.sh at the C3-sized workload (5M x 5k)
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in build/var/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  r=$(python bench.py --workload c3w100 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['kernel_ms'], d['roofline']['frac'])")
  echo "$(basename $f) c3w100 $r"
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
