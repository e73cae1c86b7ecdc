#!/bin/bash
# SQ counters of the thinned-feed kernel (lod_feed_kernel): instruction mix and what its waves wait for
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
ARGS="tools/bench_variants.py --snps ${SNPS:-1000000} --inds ${INDS:-1000} --modes feed --steps 3"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/fd_pmc_a -- python3 $ARGS > $OUT/fd_pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVES SQ_INSTS_BRANCH SQ_INSTS_VMEM --output-format csv -d $OUT/fd_pmc_b -- python3 $ARGS > $OUT/fd_pmc_b.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH --output-format csv -d $OUT/fd_pmc_c -- python3 $ARGS > $OUT/fd_pmc_c.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
for d in ("fd_pmc_a","fd_pmc_b","fd_pmc_c"):
    fs=sorted(glob.glob(f"gpurun_out/{d}/**/*_counter_collection.csv", recursive=True))
    if not fs: print(d, "no output"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if "lod_feed_kernel" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc["ms"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    print(d, {k: round(sum(v)/len(v),3) if k=="ms" else "%.4g"%(sum(v)/len(v)) for k,v in acc.items()})
PY
rm -rf $OUT/fd_pmc_a $OUT/fd_pmc_b $OUT/fd_pmc_c
