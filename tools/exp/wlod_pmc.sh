#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
ARGS="tools/bench_variants.py --snps 2000000 --inds 1280 --modes wlod,wlodgl --steps 3"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/wl_pmc_a -- python3 $ARGS > $OUT/wl_pmc_a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_SMEM SQ_WAVES --output-format csv -d $OUT/wl_pmc_b -- python3 $ARGS > $OUT/wl_pmc_b.log 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/wl_pmc_c -- python3 $ARGS > $OUT/wl_pmc_c.log 2>&1
python3 - <<'PY'
import csv, glob, collections, os
OUT=os.environ.get("OUT", "gpurun_out")
for d in ("wl_pmc_a","wl_pmc_b","wl_pmc_c"):
    f=sorted(glob.glob(f"gpurun_out/{d}/**/*_counter_collection.csv", recursive=True))[-1]
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "wlod_tile" not in k: continue
        name="glring" if "glring" in k else "plain"
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[name]["ms"].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
    for name,c in acc.items():
        print(d, name, {k: round(sum(v)/len(v),3) if k=="ms" else "%.4g"%(sum(v)/len(v)) for k,v in c.items()})
PY
rm -rf $OUT/wl_pmc_a $OUT/wl_pmc_b $OUT/wl_pmc_c
