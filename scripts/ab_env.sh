#!/bin/bash
# Runs ON the GPU box: same-box A/B over "LIB[,ENV=VAL...]" specs (LIB = default or a path):
#   gpurun -- 'bash scripts/ab_env.sh TAG default default,SCORP_FWD_GRID=6144 build/variants/libx.so,SCORP_FWD_GRID=6144'
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for spec in "$@"; do
  IFS=',' read -ra parts <<< "$spec"
  lib=${parts[0]}
  envs=("${parts[@]:1}")
  (
    if [ "$lib" != default ]; then export SCORP_GS_LIB=$PWD/$lib; fi
    for e in "${envs[@]}"; do export "$e"; done
    python3 bench.py --no-cpu-baseline --no-secondary --steps ${STEPS:-40} --warmup 10 ${BENCH_ARGS} > $out/tmp.json 2> $out/tmp.err || { echo "$spec FAILED" >> $out/ab.txt; tail -3 $out/tmp.err >> $out/ab.txt; exit 0; }
    python3 - "$spec" $out/tmp.json >> $out/ab.txt << 'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = d["kernels_us_GBs"]
print(f"{sys.argv[1]:56s} views/s {d['value']:8.1f}  " + "  ".join(f"{n.replace('blend_','b').replace('preprocess','pp').replace('ssim_l1_','loss_')}={v[0]:.1f}" for n, v in k.items()))
PY
  )
done
cat $out/ab.txt
