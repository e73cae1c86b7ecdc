#!/bin/bash
# rocprofv3 summaries of the thinned-feed kernel (lod_feed_kernel) at config 3's shape -- 5M SNPs x 5000 individuals,
# --winsize-multi 50 100 200 300: kernel-trace stats of the four sizes (single calls and one garlic_lod_feed_multi call),
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and the SQ instruction / wait counters at W = 100.
# Run on the GPU box through gpurun; tools/summarize_feed.py TAG condenses the output into profiles/.
#   usage: tools/profile_feed.sh r03
set -e
TAG=${1:-r03}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
rm -rf $OUT/${TAG}_feed_trace $OUT/${TAG}_feed_fetch $OUT/${TAG}_feed_write $OUT/${TAG}_feed_sq
python3 tools/exp/feed_multi_time.py > $OUT/${TAG}_feed_plain.json 2> $OUT/${TAG}_feed.err
echo "plain done"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_feed_trace -- python3 $R/tools/exp/feed_multi_time.py > $OUT/${TAG}_feed_traced.json 2>> $OUT/${TAG}_feed.err
echo "trace done"
ONE="$R/tools/bench_variants.py --snps 5000000 --inds 5000 --winsize 100 --modes feed --steps 3"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_feed_fetch -- python3 $ONE > $OUT/${TAG}_feed_fetch.json 2>> $OUT/${TAG}_feed.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_feed_write -- python3 $ONE > $OUT/${TAG}_feed_write.json 2>> $OUT/${TAG}_feed.err
echo "pmc traffic done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_feed_sq/a -- python3 $ONE > /dev/null 2>> $OUT/${TAG}_feed.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_VMEM --output-format csv -d $OUT/${TAG}_feed_sq/b -- python3 $ONE > /dev/null 2>> $OUT/${TAG}_feed.err
echo "sq counters done"
cd $R
# the raw traces are tens of MB (gpurun brings back 64 MiB at most): condense here, keep the summaries only
GARLIC_PROF_OUT=$OUT/profiles_${TAG} python3 tools/summarize_feed.py $TAG
rm -rf $OUT/${TAG}_feed_trace $OUT/${TAG}_feed_fetch $OUT/${TAG}_feed_write $OUT/${TAG}_feed_sq
