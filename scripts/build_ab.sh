#!/bin/bash
# usage: scripts/build_ab.sh NAME "FLAGS" file1.hip [file2.hip ...]  -> build/variants/libNAME.so
# The listed kernel files are compiled with FLAGS (e.g. -DSCORP_BWD_WAVES=3), every other object comes from build/*.o
# (python -m scorp_amd.build first).  For same-box A/B runs through scripts/ab_variants.sh (SCORP_GS_LIB).
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; shift 2
mkdir -p build/variants
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -Iinclude -Iscorp_amd/csrc"
objs=""
skip=""
for f in "$@"; do
  $HIPCC -c $flags scorp_amd/csrc/$f -o build/variants/${name}_$f.o 2>build/variants/${name}_$f.err &
  objs="$objs build/variants/${name}_$f.o"
  skip="$skip|/$f.o"
done
wait
others=$(ls build/*.hip.o | grep -Ev "${skip#|}")
$HIPCC -shared -o build/variants/lib$name.so $others $objs
echo build/variants/lib$name.so
