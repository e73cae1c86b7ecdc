#!/bin/bash
# HBM traffic of every kernel of bench.py's shard leg (10M x 1250): FETCH_SIZE / WRITE_SIZE, separate passes
O=$GRAFT_REPO_ROOT/gpurun_out/r4; mkdir -p $O; cd /tmp; export TMPDIR=/tmp GARLIC_BENCH_NO_CLOCK=1
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 550 rocprofv3 --pmc $c --output-format csv -d /tmp/np_$c -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --also ns > $O/ns_pmc_$c.json 2> $O/ns_pmc_$c.err || { echo failed $c; tail -3 $O/ns_pmc_$c.err; }
  python3 $R/tools/exp/pmc_kernels.py /tmp/np_$c $c | tee $O/ns_pmc_$c.txt
done
