#!/bin/bash
timeout -k 10 900 python3 -m pytest tests/test_gpu_ld.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ldprof5
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ldprof5 -o ld -- python3 tools/bench_variants.py --snps 10000000 --inds 1250 --winsize 100 --modes ld --steps 3 > gpurun_out/ldprof5.log 2>&1
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/ldprof5/**/*kernel_stats.csv', recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if 'garlic::ld' in r['Name'] or 'rocclr' in r['Name']:
        print(r['Name'][:50], r['Calls'], round(float(r['TotalDurationNs'])/1e6/4,2))
PY
grep call_ms gpurun_out/ldprof5.log | head -1 | cut -c1-110
