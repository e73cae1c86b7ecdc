#!/bin/bash
# LD weights: parity tests, then the warm call at the shard shape (10M SNPs x 1250 individuals, W = 100) fused and in two steps
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ld.py tests/test_gpu_switches.py -x -q -m gpu -k "ld or LD" > $O/t_ld_${TAG:-a}.log 2>&1; rc=$?; tail -4 $O/t_ld_${TAG:-a}.log
[ $rc -ne 0 ] && exit $rc
for env in "" "GARLIC_LD_UNFUSED=1"; do
  echo "== ${env:-fused}"
  env $env timeout -k 10 300 python tools/bench_variants.py --snps 10000000 --inds 1250 --winsize 100 --modes ld --steps 5 2>/dev/null | tail -1
done | tee $O/ld_time_${TAG:-a}.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/ld_trace -- python3 $OLDPWD/tools/bench_variants.py --snps 10000000 --inds 1250 --winsize 100 --modes ld --steps 5 > /dev/null 2>&1; cd $OLDPWD
python tools/exp/kstats.py $O/ld_trace | tee $O/ld_kernels_${TAG:-a}.txt; rm -rf $O/ld_trace
