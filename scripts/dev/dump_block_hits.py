"""Writes the blend forward's per-block iteration counts of one S3 view (block_hits[tile * 4 + quad], int32) to a file:
the work distribution scripts/mb_wg_barrier.hip replays.  Usage: python scripts/dev/dump_block_hits.py out.bin [camera]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scorp_amd import rasterizer3d as R
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.renderer import render
from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
dev = torch.device('cuda:0')
N, W, H, deg, seed, ncam = SCENES["S3"]
model = GaussianModel.from_raw(make_gaussians(N, deg, seed), deg, device=dev); model.active_sh_degree = deg
class Pipe: convert_SHs_python = False; compute_cov3D_python = False; debug = False; fused_activations = True
cam = ring_cameras(ncam, W, H, seed, device=dev)[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
R.KEEP_LAST_FORWARD = True
render(cam, model, Pipe(), torch.zeros(3, device=dev))
st, n_, w_, h_ = R.LAST_FORWARD
tiles = ((W + 15) // 16) * ((H + 15) // 16)
al = lambda v, a=256: (v + a - 1) // a * a     # mirror of StateLayout (common.hpp)
off = 0
off = al(off + 64); off = al(off + N * 48); off = al(off + N * 16); off = al(off + N * 8)
off = al(off + (tiles + 1) * 4); off = al(off + (tiles + 1) * 4); off = al(off + W * H * 4); off = al(off + W * H * 4)
bh = st[off:off + tiles * 16].view(torch.int32).cpu().numpy()
bh.tofile(sys.argv[1])
t = bh.reshape(tiles, 4).astype(np.int64)
print(f"blocks {bh.size}, hits {int(t.sum())}, mean {t.mean():.1f}, 4*max/sum over tiles {4 * t.max(1).sum() / t.sum():.3f}")
