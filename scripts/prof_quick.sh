#!/bin/bash
# Runs ON the GPU box: kernel stats + the two SQ counter passes of the S3 step, summarised (a quick look at a kernel change).
#   gpurun -- 'bash scripts/prof_quick.sh TAG'  -> gpurun_out/TAG/{kernel_stats.csv,pmc_sq.json}
tag=${1:-pq}; out=gpurun_out/$tag; mkdir -p $out
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $out/stats.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $out/sq1 -- python3 bench.py --steps 3 --warmup 1 --lead-in 2 --no-twins --cams 8 --no-cpu-baseline --no-kernel-events --no-secondary > $out/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/sq2 -- python3 bench.py --steps 3 --warmup 1 --lead-in 2 --no-twins --cams 8 --no-cpu-baseline --no-kernel-events --no-secondary > $out/sq2.log 2>&1
f() { find $out/$1 -name '*counter_collection.csv' | head -1; }
python3 scripts/pmc_summary.py $(f sq1) $(f sq2) > $out/pmc_sq.json
cp "$(find $out/stats -name '*kernel_stats.csv' | head -1)" $out/kernel_stats.csv
python3 - $out <<'PY'
import csv, json, sys
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f}")
d = json.load(open(out + "/pmc_sq.json"))
for n, v in d.items():
    if "ssim" in n:
        print(n[:50], {k: round(x / 1e6, 2) for k, x in v.items()})
PY
