#!/bin/bash
# Runs ON the GPU box: the two SQ counter passes of collect_profiles.sh alone (S3, 3 views), summarised per launch.
#   gpurun -- 'bash scripts/sq_passes.sh TAG [lib.so]'  -> gpurun_out/TAG/pmc_sq.json
set -e -o pipefail
tag=${1:-sq}; out=gpurun_out/$tag; mkdir -p $out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
if [ -n "$2" ]; then export SCORP_GS_LIB=$PWD/$2; fi
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $out/sq1 -- python3 bench.py --steps 3 --warmup 1 --lead-in 2 --no-twins --cams 8 --no-cpu-baseline --no-kernel-events --no-secondary $SQ_BENCH_ARGS > $out/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $out/sq2 -- python3 bench.py --steps 3 --warmup 1 --lead-in 2 --no-twins --cams 8 --no-cpu-baseline --no-kernel-events --no-secondary $SQ_BENCH_ARGS > $out/sq2.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM --kernel-trace --output-format csv -d $out/sq3 -- python3 bench.py --steps 3 --warmup 1 --lead-in 2 --no-twins --cams 8 --no-cpu-baseline --no-kernel-events --no-secondary $SQ_BENCH_ARGS > $out/sq3.log 2>&1 || echo "sq3 pass failed (counter names?)"
f() { find $out/$1 -name '*counter_collection.csv' | head -1; }
python3 scripts/pmc_summary.py $(f sq1) $(f sq2) $(f sq3) > $out/pmc_sq.json
python3 - $out/pmc_sq.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
for n,v in d.items():
    if 'blend' in n: print(n, {k:round(x/1e6,2) for k,x in v.items()})
PY
