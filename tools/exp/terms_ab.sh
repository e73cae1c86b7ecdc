#!/bin/bash
timeout -k 10 900 python3 -m pytest tests/test_gpu_variants.py tests/test_gpu_tgls_continuous.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in GARLIC_X=1 GARLIC_GL_TERMS_GATHER=1; do
rm -rf gpurun_out/tprof
env $e rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tprof -o t -- python3 tools/bench_variants.py --snps 10000000 --inds 1250 --winsize 100 --modes tgls --steps 3 > gpurun_out/tprof.log 2>&1
echo $e
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/tprof/**/*kernel_stats.csv', recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if 'gl_terms' in r['Name'] or 'fillBuffer' in r['Name'] or 'ring' in r['Name']:
        print('  ', r['Name'][:50], r['Calls'], round(float(r['TotalDurationNs'])/1e6,2))
PY
done
