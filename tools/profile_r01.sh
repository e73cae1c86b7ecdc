#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun); summaries are copied to profiles/.
set -e
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
ARGS="bench.py --steps 10 --warmup 2 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r01_trace -- python3 $ARGS > $OUT/r01_trace_bench.json 2> $OUT/r01_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/r01_pmc_fetch -- python3 $ARGS > $OUT/r01_pmc_fetch_bench.json 2> $OUT/r01_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/r01_pmc_write -- python3 $ARGS > $OUT/r01_pmc_write_bench.json 2> $OUT/r01_pmc_write.err
find $OUT/r01_trace $OUT/r01_pmc_fetch $OUT/r01_pmc_write -name "*.csv" | head -20
