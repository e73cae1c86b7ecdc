#!/bin/bash
# round 4: the thinned feed / coverage-bit chains (tools/gen_feed_asm.py): parity tests that touch them, then their times:
# the final pass at 10M x 1250 (garlic_roh_segments, garlic_roh_coverage_fused) and the four feeds of C3 (5M x 5k)
O=gpurun_out/r4; mkdir -p $O
T=${TAG:-a}
timeout -k 10 900 python -m pytest tests/test_gpu_variants.py tests/test_gpu_fullsize.py tests/test_gpu_soak.py tests/test_gpu_shardshape.py -x -q -m gpu \
    -k "feed or thin or coverage or segments or soak or shard" > $O/t_feed_$T.log 2>&1; rc=$?; tail -5 $O/t_feed_$T.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/exp/roh_segments_time.py > $O/seg_time_$T.log 2>&1 || exit 1; cat $O/seg_time_$T.log
timeout -k 10 400 python tools/exp/r4_feed_time.py 2> $O/feed_c3_$T.err | tee $O/feed_c3_$T.log
