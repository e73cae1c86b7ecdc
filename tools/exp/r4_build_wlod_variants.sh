#!/bin/bash
# builds libgarlic_hip.so variants of the GL strip loop into build/abl/wlodgl_<name>.so: name:VAR=x,VAR=y are the
# environment of tools/gen_wlod_asm.py; SED_<n> style edits come as a sed script in $WLOD_SED (applied to wlod_strip_kernel.hpp)
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-inline-asm -pthread -shared"
for v in "$@"; do
  name=${v%%:*}; envs=""
  [[ $v == *:* ]] && envs=$(echo "${v#*:}" | tr ',' ' ')
  d=build/abl/src_$name; rm -rf $d; mkdir -p $d
  cp garlic_amd/csrc/*.hip garlic_amd/csrc/*.hpp garlic_amd/csrc/*.inc $d/
  sed -i "s#\"../../include/garlic_hip.h\"#\"$PWD/include/garlic_hip.h\"#" $d/garlic_hip.hip
  env $envs GARLIC_GEN_OUT=$d python3 tools/gen_wlod_asm.py > /dev/null
  sedvar=WLOD_SED_$name
  [[ -n "${!sedvar}" ]] && sed -i "${!sedvar}" $d/wlod_strip_kernel.hpp
  /opt/rocm/bin/hipcc $FLAGS -I include -I garlic_amd/csrc -Rpass-analysis=kernel-resource-usage -o build/abl/wlodgl_$name.so $d/garlic_hip.hip 2>&1 \
    | grep -A9 "Name: _ZN6garlic21wlod_strip_gl3_kernelE" | grep -E "error|VGPRs|Scratch" | tr '\n' ' '
  rm -rf $d
  echo built $name
done
