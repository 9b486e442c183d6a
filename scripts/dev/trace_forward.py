"""Occupancy timeline of ONE blend-forward launch on S3 from the diagnostic build (scripts/build_ab.sh fwd_trace
"-DSCORP_FWD_TRACE=1" gs3d_forward.hip):  SCORP_GS_LIB=build/variants/libfwd_trace.so python scripts/dev/trace_forward.py"""
import ctypes, json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scorp_amd import _C
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.renderer import render
from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
dev = torch.device("cuda:0")
N, W, H, deg, seed, _ = SCENES["S3"]
m = GaussianModel.from_raw(make_gaussians(N, deg, seed), deg, device=dev); m.active_sh_degree = deg
cam = ring_cameras(280, W, H, seed, device=dev)[0]


class Pipe:
    convert_SHs_python = False; compute_cov3D_python = False; debug = False; fused_activations = True


bg = torch.zeros(3, device=dev)
for _ in range(3):
    render(cam, m, Pipe(), bg)          # grad-enabled inputs: the <true> kernel
torch.cuda.synchronize()
L = _C.lib()
nw = ((W + 15) // 16) * ((H + 15) // 16) * 4
nw = (nw + 31) // 32 * 32
buf = (ctypes.c_ulonglong * (3 * nw))()
L.scorp_debug_fwd_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert L.scorp_debug_fwd_trace(buf, nw) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 3)
ok = t[:, 1] > 0
t0, t1 = t[ok, 0].astype(np.int64), t[ok, 1].astype(np.int64)
base = t0.min()
s, e = (t0 - base) * 0.01, (t1 - base) * 0.01          # microseconds (100 MHz counter)
dur = e - s
total = e.max()
grid = np.linspace(0, total, 41)
occ = [(float(((s < b) & (e > a)).sum())) for a, b in zip(grid[:-1], grid[1:])]
hw = (t[ok, 2] >> np.uint64(32)).astype(np.uint32); xcc = (t[ok, 2] & np.uint64(0xFFFFFFFF)).astype(np.uint32) & 0xF
print(json.dumps({"waves": int(ok.sum()), "kernel_us": round(float(total), 1), "wave_us_mean": round(float(dur.mean()), 2),
                  "wave_us_p50_p90_p99_max": [round(float(np.percentile(dur, p)), 1) for p in (50, 90, 99, 100)],
                  "sum_wave_us_over_slots_6144": round(float(dur.sum() / 6144), 1),
                  "waves_resident_per_2.5pct_slice": [int(o) for o in occ],
                  "start_us_p50_p90_p99_max": [round(float(np.percentile(s, p)), 1) for p in (50, 90, 99, 100)],
                  "per_xcc_last_end_us": [round(float(e[xcc == k].max()), 1) if (xcc == k).any() else None for k in range(8)],
                  "per_xcc_waves": [int((xcc == k).sum()) for k in range(8)]}))
