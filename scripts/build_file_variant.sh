#!/bin/bash
# usage: scripts/build_file_variant.sh NAME SOURCE_FILE REPLACES.hip "FLAGS"  -> build/variants/libNAME.so
# SOURCE_FILE (anywhere, e.g. a parked experiment under scripts/dev/) is compiled in place of scorp_amd/csrc/REPLACES.hip;
# every other object comes from build/*.o (python -m scorp_amd.build first).
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; repl=$3; flags=$4
mkdir -p build/variants
cp "$src" build/variants/${name}_src.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -Iinclude -Iscorp_amd/csrc $flags -c build/variants/${name}_src.hip -o build/variants/${name}.o 2>build/variants/${name}.err
others=$(ls build/*.hip.o | grep -v "/$repl.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib$name.so $others build/variants/${name}.o
echo build/variants/lib$name.so
