"""Per-kernel event table of the stacked rotation sweep (S4): python scripts/dev/prof_sweep.py [n_hypotheses]"""
import copy, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scorp_amd import _C
from scorp_amd.align import SweepPlan, render_views, rotation_sweep
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.transforms import gaussians_rotate
dev = torch.device("cuda:0")
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rots = np.load(os.path.join(ROOT, "tests", "golden", "rotations_128.npz"))["rotations"]
raw = make_gaussians(100_000, 0, 4, extent=0.8, log_scale_mean=math.log(0.01))
raw["xyz"][:, 0] *= 1.6
obj = GaussianModel.from_raw(raw, 0, device=dev)
cams = ring_cameras(15, 800, 800, 4, radius=3.0, device=dev)
bg = torch.zeros(3, device=dev)
tgt = copy.copy(obj)
tgt._xyz, tgt._rotation, tgt._features_rest = obj._xyz.detach().clone(), obj._rotation.detach().clone(), obj._features_rest.detach().clone()
gaussians_rotate(tgt, torch.tensor(rots[77], dtype=torch.float32, device=dev), fix_center=True)
targets = render_views(tgt, cams, bg)
plan = SweepPlan(obj, cams, targets, bg)
if os.environ.get("SWEEP_BATCH"):
    plan.stacked.batch = int(os.environ["SWEEP_BATCH"])
rotation_sweep(obj, rots[:4], cams, targets, bg, plan=plan)
torch.cuda.synchronize()
dts = []
for _ in range(3):      # (the first timed sweep still pays allocations of the full-length sweep: the last one is reported)
    t0 = time.perf_counter()
    ids, fit, best = rotation_sweep(obj, rots[:nh], cams, targets, bg, plan=plan)
    torch.cuda.synchronize()
    dts.append(time.perf_counter() - t0)
print("sweeps (s):", [round(x, 4) for x in dts], file=sys.stderr)
dt = dts[-1]
_C.prof_enable(True)
rotation_sweep(obj, rots[:16], cams, targets, bg, plan=plan)
torch.cuda.synchronize()
k = _C.prof_collect()
_C.prof_enable(False)
print(json.dumps({"hypotheses_per_s": round(nh / dt, 1), "ms_per_hypothesis": round(dt / nh * 1e3, 4), "best": best,
                  "kernels_us": {n: round(ms / c * 1e3, 1) for n, (ms, c) in k.items() if c}}))
