#!/bin/bash
# chain kernel time of bench.py for several persistent-workgroup counts (one box, one call)
for k in ${KS:-256 248 240 224 192 256}; do
  r=$(GARLIC_WORKERS=$k python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel_ms'])")
  echo "K=$k $r"
done
