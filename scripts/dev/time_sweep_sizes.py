"""Dev: the rotation sweep's time against the number of hypotheses on one GPU (what a rank of an N-GPU run pays at 128 / N)."""
import copy, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scorp_amd.align import SweepPlan, render_views, rotation_sweep
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.transforms import gaussians_rotate
dev = torch.device('cuda:0')
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rots = np.load(os.path.join(ROOT, "tests", "golden", "rotations_128.npz"))["rotations"]
raw = make_gaussians(100_000, 0, 4, extent=0.8, log_scale_mean=math.log(0.01)); raw["xyz"][:, 0] *= 1.6
obj = GaussianModel.from_raw(raw, 0, device=dev)
cams = ring_cameras(15, 800, 800, 4, radius=3.0, device=dev); bg = torch.zeros(3, device=dev)
tgt = copy.copy(obj)
tgt._xyz, tgt._rotation, tgt._features_rest = obj._xyz.detach().clone(), obj._rotation.detach().clone(), obj._features_rest.detach().clone()
gaussians_rotate(tgt, torch.tensor(rots[77], dtype=torch.float32, device=dev), fix_center=True)
targets = render_views(tgt, cams, bg)
plan = SweepPlan(obj, cams, targets, bg)
rotation_sweep(obj, rots[:32], cams, targets, bg, plan=plan)
for n in (128, 64, 32, 16, 8, 1):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rotation_sweep(obj, rots[:n], cams, targets, bg, plan=plan)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print(n, "hypotheses:", round(1e3 * best, 2), "ms =", round(1e6 * best / n, 1), "us each", flush=True)
