#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/tprof
GARLIC_TGLS_CONTINUOUS=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tprof -o t -- python3 tools/bench_variants.py --snps 10000000 --inds 1250 --winsize 100 --modes tgls --steps 3 > gpurun_out/tprof.log 2>&1
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/tprof/**/*kernel_stats.csv', recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if 'garlic::' in r['Name']:
        print('  ', r['Name'][:60], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2))
PY
tail -2 gpurun_out/tprof.log | cut -c1-300
