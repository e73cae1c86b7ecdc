#!/bin/bash
# rocprofv3 summaries of the final pass without a score matrix at the shard shape (10M SNPs x 1250 individuals, W = 100):
# garlic_roh_segments / garlic_roh_coverage_fused = lod_bits_kernel + the passes over its bits.  Kernel-trace stats, HBM
# traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and the SQ instruction counters of lod_bits_kernel.
#   usage (on the GPU box through gpurun): tools/profile_final_pass.sh r04   -> gpurun_out/profiles_r04/r04_final_pass_kernels.txt
TAG=${1:-r04}
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out
P=$OUT/profiles_$TAG
mkdir -p $P
F=$P/${TAG}_final_pass_kernels.txt
CMD="$R/tools/exp/roh_segments_time.py"
cd /tmp
rm -rf $OUT/fp_trace $OUT/fp_fetch $OUT/fp_write $OUT/fp_sq
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/fp_trace -- python3 $CMD > $OUT/fp_calls.txt 2> $OUT/fp.err || exit 1
echo "# rocprofv3 --kernel-trace --stats -- python3 tools/exp/roh_segments_time.py   (10M SNPs x 1250 individuals, W = 100; tools/profile_final_pass.sh $TAG)" > $F
echo "# per-kernel averages (tools/exp/kstats.py)" >> $F
python3 $R/tools/exp/kstats.py $OUT/fp_trace >> $F
echo "# wall clock of the calls in the same run:" >> $F
cat $OUT/fp_calls.txt >> $F
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fp_fetch -- python3 $CMD > /dev/null 2>> $OUT/fp.err || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/fp_write -- python3 $CMD > /dev/null 2>> $OUT/fp.err || exit 1
echo "" >> $F
echo "# HBM traffic of the same kernels: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/exp/pmc_kernels.py;" >> $F
echo "# KiB per launch as counted -- gfx950: FETCH_SIZE counts 128-B read requests as 64 B, so reads are TWICE the figure (MI355X_MICROARCH.md, HBM);" >> $F
echo "# algorithmic: bits 1.56 GB out, genotypes 3.1 GB in; counts 1.56 GB in / 25.0 GB out; mask 1.56 in / 1.56 out; segments 1.56 in" >> $F
python3 $R/tools/exp/pmc_kernels.py $OUT/fp_fetch FETCH_SIZE >> $F
python3 $R/tools/exp/pmc_kernels.py $OUT/fp_write WRITE_SIZE >> $F
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_SMEM --output-format csv -d $OUT/fp_sq -- python3 $CMD > /dev/null 2>> $OUT/fp.err || exit 1
echo "" >> $F
echo "# SQ instruction counters of lod_bits_kernel (mean per launch) and per window and wave: 10M x 1250 = 1.25e10 windows / 64 lanes" >> $F
python3 - $OUT/fp_sq >> $F <<'PY'
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "lod_bits_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
wave_windows = 10_000_000 * 1280 / 64.0      # 20 blocks of 64 lanes
tot = 0.0
for k in sorted(acc):
    m = sum(acc[k]) / len(acc[k])
    print(f"{k:24s} {m:16.0f}   per window and wave {m / wave_windows:7.3f}")
    if k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH"): tot += m
print(f"{'all of the above classes':24s} {tot:16.0f}   per window and wave {tot / wave_windows:7.3f}   (SQ_INSTS_SALU includes s_waitcnt / s_barrier / s_nop)")
PY
rm -rf $OUT/fp_trace $OUT/fp_fetch $OUT/fp_write $OUT/fp_sq
cat $F
