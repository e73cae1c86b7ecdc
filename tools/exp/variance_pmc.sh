#!/bin/bash
export TMPDIR=/tmp
OUT=$PWD/gpurun_out
rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE --output-format csv -d $OUT/var_pmc_a -- python3 tools/exp/variance_pmc.py > $OUT/var_pmc_a.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum --output-format csv -d $OUT/var_pmc_b -- python3 tools/exp/variance_pmc.py > $OUT/var_pmc_b.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_BUSY_sum TCC_CYCLE_sum --output-format csv -d $OUT/var_pmc_c -- python3 tools/exp/variance_pmc.py > $OUT/var_pmc_c.log 2>&1
python3 tools/exp/variance_pmc_join.py $OUT/var_pmc_a $OUT/var_pmc_b $OUT/var_pmc_c
