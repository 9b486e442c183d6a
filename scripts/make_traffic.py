"""profiles/traffic.json from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately).

    python scripts/make_traffic.py S3 fetch_counter_collection.csv write_counter_collection.csv [S6 fetch.csv write.csv ...]

Per kernel and launch: HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 — FETCH_SIZE / WRITE_SIZE count KiB and
gfx950 under-reports FETCH_SIZE by 2x (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

SHORT = {  # kernel symbol -> the name bench.py uses
    "preprocess_kernel": "preprocess", "count_tiles_lds_kernel": "count_tiles", "scan_tiles_kernel": "scan_tiles",
    "scan_block_hist_kernel": "scan_block_hist", "scatter_pairs_lds_kernel": "scatter_pairs", "expand_cells_kernel": "expand_cells", "sort_tiles_reg_kernel": "sort_tiles", "sort_tiles_kernel": "sort_tiles_long",
    "blend_forward_wave_kernel": "blend_forward", "blend_backward_wave_kernel": "blend_backward",
    "preprocess_backward_kernel": "preprocess_backward", "ssim_l1_forward_kernel": "ssim_l1_forward", "ssim_l1_forward_strip_kernel": "ssim_l1_forward",
    "ssim_l1_backward_kernel": "ssim_l1_backward", "preprocess2d_kernel": "preprocess_2d",
    "blend2d_forward_wave_kernel": "blend_forward_2d", "blend2d_backward_wave_kernel": "blend_backward_2d",
    "preprocess2d_backward_kernel": "preprocess_backward_2d", "maps_forward_kernel": "surfel_maps_forward",
    "maps_backward_kernel": "surfel_maps_backward",
}


def per_launch(path, counter):
    tot, ids = defaultdict(float), defaultdict(set)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            kn = row["Kernel_Name"].replace("(anonymous namespace)::", "")
            m = re.match(r"(?:void )?([\w:]+)", kn)
            name = SHORT.get(m.group(1).split("::")[-1]) if m else None
            if name:
                tot[name] += float(row["Counter_Value"])
                ids[name].add(row["Dispatch_Id"])
    return {k: tot[k] / len(ids[k]) for k in tot}


root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = os.path.join(root, "profiles", "traffic.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
args = sys.argv[1:]
for scene, fpath, wpath in zip(args[0::3], args[1::3], args[2::3]):
    f, w = per_launch(fpath, "FETCH_SIZE"), per_launch(wpath, "WRITE_SIZE")
    out[scene] = {k: int((2 * f[k] + w.get(k, 0.0)) * 1024) for k in f}
    out["_detail_" + scene] = {k: {"FETCH_SIZE_KB": round(f[k], 1), "WRITE_SIZE_KB": round(w.get(k, 0.0), 1)} for k in f}
sys.path.insert(0, root)
from scorp_amd.build import source_sha  # noqa: E402
out["source_sha"] = source_sha()   # bench.py reports roofline.traffic only while the loaded library carries the same stamp
json.dump(out, open(out_path, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if not k.startswith("_")}, indent=1))
