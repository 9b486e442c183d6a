#!/bin/bash
# usage: scripts/install_profiles.sh TAG   - copies what scripts/collect_profiles.sh TAG left under gpurun_out/TAG into
# profiles/TAG_* (counter files cut down to this library's kernels) and installs its traffic.json.
set -e
cd "$(dirname "$0")/.."
tag=$1; o=gpurun_out/$tag
cp $o/bench_S3.json profiles/${tag}_bench.json; cp $o/bench_S6.json profiles/${tag}_bench_S6.json
cp $o/kernel_stats_S3.csv profiles/${tag}_kernel_stats.csv; cp $o/kernel_stats_S6.csv profiles/${tag}_kernel_stats_S6.csv
cp $o/pmc_sq.json profiles/${tag}_pmc_sq.json; cp $o/traffic.json profiles/traffic.json
for sc in S3 S6; do for c in FETCH_SIZE WRITE_SIZE; do
  f=$(find $o/pmc_${sc}_$c -name '*counter_collection.csv' | head -1)
  python3 - "$f" profiles/${tag}_pmc_${sc}_$c.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "scorp::" in r["Kernel_Name"]]
w = csv.DictWriter(open(sys.argv[2], "w", newline=""), fieldnames=rows[0].keys())
w.writeheader(); w.writerows(keep)
PY
done; done
python3 - $tag <<'PY'
import json, csv, sys
tag = sys.argv[1]
d = json.loads(open(f"profiles/{tag}_bench.json").read().strip().splitlines()[-1])
print(len(json.dumps(d)), "bytes;", {k: d[k] for k in ("value", "value_exact_fp32", "value_deterministic_backward", "ms_per_step")})
print(d["roofline"]); print(d["cpu_baseline"]); print(d["kernels_us_GBs"]); print(d["views_per_s_two_in_flight"], d["forward_only_views_per_s_per_gpu"], d["config"]["pixel_splat_pairs_P"], d["config"]["pixel_splat_pairs_P_backward"])
print({k: (v.get("value"), v.get("error")) for k, v in d["secondary"].items()})
for r in list(csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")))[:13]:
    print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} min {float(r['MinNs'])/1e3:8.1f}")
t = json.load(open("profiles/traffic.json")); print(t["source_sha"], t["S3"])
PY
