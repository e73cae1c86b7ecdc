#!/bin/bash
# rocprofv3 kernel-trace stats of the secondary kernels (wLOD, TGLS, LD weights, thinned feed); run on
# the GPU box through gpurun, then tools/summarize_variants.py condenses the table into profiles/.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -rf $OUT/r01_variants
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r01_variants -- python3 tools/bench_variants.py --modes ld,feed,lod,tgls,wlod,wlodgl --steps 5 > $OUT/r01_variants_bench.json 2> $OUT/r01_variants.err
find $OUT/r01_variants -name "*_kernel_stats.csv"
