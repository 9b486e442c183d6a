#!/bin/bash
# Runs ON the GPU box: same-box A/B of library variants (scripts/build_variant.sh) through bench.py's per-kernel event table.
#   gpurun -- 'bash scripts/ab_variants.sh TAG lib1.so lib2.so ...'   -> gpurun_out/TAG/ab.txt
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for lib in default "$@"; do
  if [ "$lib" = default ]; then unset SCORP_GS_LIB; else export SCORP_GS_LIB=$PWD/$lib; fi
  python3 bench.py --no-cpu-baseline --steps ${STEPS:-40} --warmup 10 ${BENCH_ARGS} > $out/tmp.json 2> $out/tmp.err || { echo "$lib FAILED" >> $out/ab.txt; tail -3 $out/tmp.err >> $out/ab.txt; continue; }
  python3 - "$lib" $out/tmp.json >> $out/ab.txt << 'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["kernels"]
print(f"{sys.argv[1]:44s} views/s {d['value']:8.1f}  " + "  ".join(f"{n.replace('blend_','b').replace('preprocess','pp')}={v['avg_us']:.1f}" for n, v in k.items()))
PY
done
cat $out/ab.txt
