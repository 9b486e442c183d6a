"""Distribution of per-block work of one S3 view (dev diagnostic): reads the forward's block_hits and n_contrib from the
state blob and simulates greedy dispatch of the 30 016 one-wave workgroups onto the chip's wave slots."""
import ctypes, heapq, math, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from scorp_amd import _C, rasterizer3d as R
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.renderer import render
from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
dev = torch.device('cuda:0')
N, W, H, deg, seed, ncam = SCENES["S3"]
model = GaussianModel.from_raw(make_gaussians(N, deg, seed), deg, device=dev); model.active_sh_degree = deg
class Pipe: convert_SHs_python = False; compute_cov3D_python = False; debug = False; fused_activations = True
cam = ring_cameras(ncam, W, H, seed, device=dev)[0]
R.KEEP_LAST_FORWARD = True
render(cam, model, Pipe(), torch.zeros(3, device=dev))
st, n_, w_, h_ = R.LAST_FORWARD
tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
tiles = tiles_x * tiles_y
# locate block_hits inside the state: mirror StateLayout (common.hpp)
al = lambda v, a=256: (v + a - 1) // a * a
off = 0
off = al(off + 64); off = al(off + N * 48); off = al(off + N * 16); off = al(off + N * 8)
off = al(off + (tiles + 1) * 4); off = al(off + (tiles + 1) * 4); off = al(off + W * H * 4); off = al(off + W * H * 4)
bh = st[off:off + tiles * 16].view(torch.int32).cpu().numpy().astype(np.int64)
print("blocks", bh.size, "mean hits", bh.mean(), "max", bh.max(), "p50", np.percentile(bh, 50), "p90", np.percentile(bh, 90), "p99", np.percentile(bh, 99))
# launch order: blockIdx b -> xcd = b & 7, kk = b >> 3, tile = (kk >> 2) * 8 + xcd, quad = kk & 3
order = []
blocks = ((tiles + 7) // 8) * 8 * 4
for b in range(blocks):
    xcd, kk = b & 7, b >> 3
    t, q = (kk >> 2) * 8 + xcd, kk & 3
    order.append(bh[t * 4 + q] if t < tiles else 0)
order = np.array(order, np.float64) + 8.0   # + fixed per-wave overhead in "hit" units
for slots in (1024 * 3, 1024 * 4, 1024 * 5):
    for name, seq in (("launch order", order), ("longest first", np.sort(order)[::-1])):
        h = [0.0] * slots
        heapq.heapify(h)
        for wk in seq:
            heapq.heappush(h, heapq.heappop(h) + wk)
        print(f"  slots {slots}: {name:14s} makespan {max(h):9.1f}  ideal {order.sum() / slots:9.1f}  ratio {max(h) / (order.sum() / slots):.3f}")
