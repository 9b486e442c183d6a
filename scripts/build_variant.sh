#!/bin/bash
# usage: scripts/build_variant.sh NAME FILE.hip [-DFLAGS...]  -> build/variants/libNAME.so (other objects from build/*.o)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -c -Iinclude -Iscorp_amd/csrc "$@" scorp_amd/csrc/$src -o build/variants/$name.o 2>/dev/null
objs=$(ls build/*.hip.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib$name.so $objs build/variants/$name.o
echo build/variants/lib$name.so
