#!/bin/bash
# what bounds the thinned feed kernel, and what a kernel serving several window sizes from one genotype
# stream ("K2m") could save: the thinned chain with parts of its input side removed.
#   nochunk   no genotype-stream loads at all  = the most a second window size could save by sharing the stream
#   noexpand  PRE does not expand genotype pairs (W-dependent work: not shareable)
#   notab     no term-row loads
for v in "GARLIC_NSLOT=8" "GARLIC_ABLATE=nochunk" "GARLIC_ABLATE=noexpand" "GARLIC_ABLATE=notab" "GARLIC_ABLATE=nodma" "GARLIC_ABLATE=nodp"; do
  env $v python3 tools/gen_chain_asm.py > /dev/null && make -s -C garlic_amd/csrc 2>&1 | grep -E "error"
  for sz in "--snps 1000000 --inds 1000" "--snps 5000000 --inds 5000"; do
    r=$(python3 tools/bench_variants.py $sz --modes feed --steps 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['chain_kernel_ms'])")
    echo "$v | $sz | thinned feed kernel ms: $r"
  done
done
