#!/bin/bash
# Runs ON the GPU box (gpurun -- 'bash scripts/collect_profiles.sh r01i'): bench lines, rocprofv3 kernel stats and the
# PMC passes (each counter set in its own run, never together with a trace domain other than --kernel-trace) for S3 and
# S6, all under gpurun_out/<tag>/.  Copy what is to be judged into profiles/ afterwards (see profiles/README.md).
set -e -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python3 bench.py > $out/bench_S3.json 2> $out/bench_S3.err
python3 bench.py --scene S6 --steps 50 > $out/bench_S6.json 2> $out/bench_S6.err
for sc in S3 S6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$sc -- python3 bench.py --scene $sc --steps 30 --warmup 5 --no-cpu-baseline > $out/stats_$sc.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${sc}_$c -- python3 bench.py --scene $sc --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $out/pmc_${sc}_$c.log 2>&1
  done
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $out/sq1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $out/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $out/sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events > $out/sq2.log 2>&1
echo done > $out/DONE
