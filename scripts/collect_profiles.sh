#!/bin/bash
# Runs ON the GPU box (gpurun -- 'bash scripts/collect_profiles.sh r02x'): bench lines, rocprofv3 kernel stats and the
# PMC passes (each counter set in its own run, never together with a trace domain other than --kernel-trace) for S3 and
# S6, all under gpurun_out/<tag>/, then profiles/traffic.json (stamped with the kernel-source hash) from the two TCC
# passes.  Copy what is to be judged into profiles/ afterwards (see profiles/README.md).
set -e -o pipefail
tag=${1:-prof}
out=gpurun_out/$tag
mkdir -p $out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
for sc in S3 S6; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$sc -- python3 bench.py --scene $sc --steps 30 --warmup 5 --no-twins --cams 8 --no-cpu-baseline --no-secondary > $out/stats_$sc.log 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_${sc}_$c -- python3 bench.py --scene $sc --steps 3 --warmup 1 --lead-in 2 --no-twins --cams 8 --no-cpu-baseline --no-kernel-events --no-secondary > $out/pmc_${sc}_$c.log 2>&1
  done
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $out/sq1 -- python3 bench.py --steps 3 --warmup 1 --lead-in 2 --no-twins --cams 8 --no-cpu-baseline --no-kernel-events --no-secondary > $out/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS --kernel-trace --output-format csv -d $out/sq2 -- python3 bench.py --steps 3 --warmup 1 --lead-in 2 --no-twins --cams 8 --no-cpu-baseline --no-kernel-events --no-secondary > $out/sq2.log 2>&1
# summaries (small files; the raw counter CSVs stay under gpurun_out/)
f() { find $out/$1 -name '*counter_collection.csv' | head -1; }
python3 scripts/make_traffic.py S3 "$(f pmc_S3_FETCH_SIZE)" "$(f pmc_S3_WRITE_SIZE)" S6 "$(f pmc_S6_FETCH_SIZE)" "$(f pmc_S6_WRITE_SIZE)" > $out/traffic_summary.json
cp profiles/traffic.json $out/traffic.json
# the bench lines last: roofline.traffic comes from the profiles/traffic.json just written (same kernel-source hash)
python3 bench.py > $out/bench_S3.json 2> $out/bench_S3.err
python3 bench.py --scene S6 --steps 50 --warmup 10 > $out/bench_S6.json 2> $out/bench_S6.err
python3 scripts/pmc_summary.py $(f sq1) $(f sq2) > $out/pmc_sq.json
for sc in S3 S6; do cp "$(find $out/stats_$sc -name '*kernel_stats.csv' | head -1)" $out/kernel_stats_$sc.csv; done
echo done > $out/DONE
