#!/bin/bash
# Per-round profiling recipe (run on the GPU box through gpurun, ONE lease): the driver's bench command
# plainly, then the same headline under rocprofv3 --kernel-trace and the two PMC passes, so that the
# committed kernel average, the bench line's kernel_ms and the HBM traffic all come from one box.
#   usage: tools/profile_round.sh r02      -> gpurun_out/r02_*; then tools/summarize_profiles.py r02
set -e
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_plain.json 2> $OUT/${TAG}_bench_plain.err
echo "plain bench done"
ARGS="bench.py --steps 20 --warmup 5 --no-cpu --also none"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $ARGS > $OUT/${TAG}_trace_bench.json 2> $OUT/${TAG}_trace.err
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 $ARGS > $OUT/${TAG}_pmc_fetch_bench.json 2> $OUT/${TAG}_pmc_fetch.err
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python3 $ARGS > $OUT/${TAG}_pmc_write_bench.json 2> $OUT/${TAG}_pmc_write.err
echo "pmc write done"
python3 bench.py --steps 20 --warmup 5 --no-cpu --also none > $OUT/${TAG}_bench_plain2.json 2>> $OUT/${TAG}_bench_plain.err
# the raw traces exceed what gpurun brings back (64 MiB): condense here, keep the summaries
GARLIC_PROF_OUT=$OUT/profiles_${TAG} python3 tools/summarize_profiles.py $TAG
cp $OUT/${TAG}_bench_plain.json $OUT/profiles_${TAG}/${TAG}_bench_plain.json
cp $OUT/${TAG}_bench_plain2.json $OUT/profiles_${TAG}/${TAG}_bench_plain_second_run.json
rm -rf $OUT/${TAG}_trace $OUT/${TAG}_pmc_fetch $OUT/${TAG}_pmc_write
