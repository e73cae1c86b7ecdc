#!/bin/bash
# builds libgarlic_hip.so variants of the feed / bits loop into build/abl/feed_<name>.so (timing experiments; results of an
# ablated loop are wrong).  Run here (hipcc cross-compiles), the .so files travel with gpurun.
set -e
cd "$(dirname "$0")/../.."
mkdir -p build/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-inline-asm -pthread -shared"
build() {   # name, env assignments...
  name=$1; shift
  d=build/abl/src_$name; rm -rf $d; mkdir -p $d
  cp garlic_amd/csrc/*.hip garlic_amd/csrc/*.hpp garlic_amd/csrc/*.inc $d/
  sed -i "s#\"../../include/garlic_hip.h\"#\"$PWD/include/garlic_hip.h\"#" $d/garlic_hip.hip
  env "$@" GARLIC_GEN_OUT=$d python3 tools/gen_feed_asm.py > /dev/null
  /opt/rocm/bin/hipcc $FLAGS -I include -I garlic_amd/csrc -Rpass-analysis=kernel-resource-usage -o build/abl/feed_$name.so $d/garlic_hip.hip 2>&1 | grep -A9 "Name: _ZN6garlic15lod_feed_kernelE" | grep -E "error|VGPRs|Scratch|Occupancy" | tr "\n" " " || true
  rm -rf $d
  echo built $name
}
# arguments: name or name:VAR=x,VAR=y (environment of tools/gen_feed_asm.py); a bare name other than "main" is an ablation
for v in "$@"; do
  name=${v%%:*}
  if [[ $v == *:* ]]; then envs=$(echo "${v#*:}" | tr ',' ' '); build $name $envs
  elif [[ $v == main ]]; then build main X=1
  else build $v GARLIC_FEED_ABLATE=$v; fi
done
