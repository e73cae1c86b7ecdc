#!/bin/bash
# Rebuild the device ISA (build/asm) and print the chain kernel's resource usage and wait structure.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -s -C "$ROOT/garlic_amd/csrc" asm
cd "$ROOT/build/asm"
grep "error" resource_usage.txt && exit 1
K=${1:-_ZN6garlic16lod_chain_kernelILb1EEEvNS_9ChainArgsE}
grep -A9 "Function Name: $K" resource_usage.txt | grep -i "VGPRs:\|LDS\|Occ\|Scratch\|SGPRs:" | sed 's/.*remark: [^ ]* *//'
awk "/^$K:/,/s_endpgm/" garlic_hip-hip-amdgcn-amd-amdhsa-gfx950.s > chain1.s
wc -l chain1.s
grep -n "Loop Header\|s_waitcnt vmcnt" chain1.s | tail -14
