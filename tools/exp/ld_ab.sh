#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for W in 200 256 300; do
rm -rf gpurun_out/ldprof4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ldprof4 -o ld -- python3 tools/bench_variants.py --snps 2000000 --inds 1280 --winsize $W --modes ld --steps 3 > gpurun_out/ldprof4.log 2>&1
echo "W=$W"
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/ldprof4/**/*kernel_stats.csv', recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if 'garlic::ld' in r['Name'] or 'rocclr' in r['Name'] or 'skew' in r['Name'] or 'reciprocal' in r['Name']:
        print('  ', r['Name'][:50], r['Calls'], round(float(r['TotalDurationNs'])/1e6/4,2))
PY
done
