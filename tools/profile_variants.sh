#!/bin/bash
# rocprofv3 summaries of the secondary kernels at the north star's shard shape (10M SNPs x 1250 individuals,
# W = 100): kernel-trace stats of every variant (LD weights, thinned feed, unweighted, TGLS, wLOD, GL-weighted
# wLOD) and the HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the TGLS chain.  Run on the GPU box
# through gpurun; tools/summarize_variants.py TAG condenses the output into profiles/.
#   usage: tools/profile_variants.sh r02 [snps inds]
set -e
TAG=${1:-r02}
SNPS=${2:-10000000}
INDS=${3:-1250}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_variants $OUT/${TAG}_tgls_fetch $OUT/${TAG}_tgls_write $OUT/${TAG}_wlodgl_fetch
ARGS="tools/bench_variants.py --snps $SNPS --inds $INDS --modes ld,feed,lod,tgls,wlod,wlodgl --steps 3"
python3 $ARGS > $OUT/${TAG}_variants_plain.json 2> $OUT/${TAG}_variants.err
echo "plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_variants -- python3 $ARGS > $OUT/${TAG}_variants_bench.json 2>> $OUT/${TAG}_variants.err
echo "trace done"
TG="tools/bench_variants.py --snps $SNPS --inds $INDS --modes tgls --steps 3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_tgls_fetch -- python3 $TG > $OUT/${TAG}_tgls_fetch.json 2>> $OUT/${TAG}_variants.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_tgls_write -- python3 $TG > $OUT/${TAG}_tgls_write.json 2>> $OUT/${TAG}_variants.err
WG="tools/bench_variants.py --snps $SNPS --inds $INDS --modes wlodgl --steps 3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_wlodgl_fetch -- python3 $WG > $OUT/${TAG}_wlodgl_fetch.json 2>> $OUT/${TAG}_variants.err
echo "pmc done"
GARLIC_PROF_OUT=$OUT/profiles_${TAG} python3 tools/summarize_variants.py $TAG
rm -rf $OUT/${TAG}_variants $OUT/${TAG}_tgls_fetch $OUT/${TAG}_tgls_write $OUT/${TAG}_wlodgl_fetch
