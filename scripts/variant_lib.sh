#!/bin/bash
# Build a variant of libmicronet_hip.so with one source recompiled under extra -D flags (A/B inside one gpurun call via MN_LIB_PATH).
# Usage: scripts/variant_lib.sh <tag> <source.hip> <flags...>   -> micronet_amd/lib/libmicronet_hip_<tag>.so
set -e
cd "$(dirname "$0")/.."
L=micronet_amd/lib
tag=$1; src=$2; shift 2
base=${src%.hip}
extra="-fno-slp-vectorize"; [ -n "$MN_VARIANT_SLP" ] && extra=""          # as micronet_amd/build.py FLAGS
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-gpu-rdc $extra "$@" -c micronet_amd/csrc/$src -o $L/${base}_$tag.o
objs=""
for f in $(python -c "from micronet_amd.build import SOURCES; print(' '.join(s[:-4] for s in SOURCES))"); do          # (every object of the library: micronet_amd/build.py)
  if [ $f = $base ]; then objs="$objs $L/${base}_$tag.o"; else objs="$objs $L/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmicronet_hip_$tag.so $objs
echo $L/libmicronet_hip_$tag.so
