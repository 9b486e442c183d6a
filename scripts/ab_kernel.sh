#!/bin/bash
# Runs ON the GPU box: rocprofv3 kernel-trace average of kernels matching PATTERN for each library given.
#   gpurun -- 'bash scripts/ab_kernel.sh TAG PATTERN default build/variants/libx.so ...'   (AB_BENCH_ARGS="--scene S6": extra bench.py arguments)
tag=$1; pat=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  ( if [ "$lib" != default ]; then export SCORP_GS_LIB=$PWD/$lib; fi
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 bench.py --steps 20 --warmup 3 --lead-in 5 --no-cpu-baseline --no-secondary --no-kernel-events $AB_BENCH_ARGS > $out/$name.log 2>&1 )
  f=$(find $out/$name -name '*kernel_stats.csv' | head -1)
  python3 - "$name" "$f" "$pat" <<'PY' | tee -a $out/ab_kernel.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[2])):
    if sys.argv[3] in r["Name"]:
        print(f"{sys.argv[1]:24s} {r['Name'][28:88]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:7.1f} min {float(r['MinNs'])/1e3:7.1f}")
PY
done
