#!/bin/bash
# like run_var.sh, but prints the whole step (ms_per_step) next to the kernel time
cp garlic_amd/libgarlic_hip.so /tmp/orig.so
for f in build/var/*.so; do
  cp $f garlic_amd/libgarlic_hip.so
  r=$(python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])")
  echo "$(basename $f) step/kernel ms: $r"
done
cp /tmp/orig.so garlic_amd/libgarlic_hip.so
