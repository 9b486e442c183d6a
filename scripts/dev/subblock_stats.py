"""Dev diagnostic: how many blend iterations a wave would run if each of its four 4x4 sub-blocks kept its own hit list
(iterations = longest of the four) instead of one list per 8x8 block.  Geometry from scorp_gs3d_debug_geom of one S3 view."""
import ctypes, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from scorp_amd import _C, rasterizer3d as R
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.renderer import render
from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
dev = torch.device('cuda:0')
scene = sys.argv[1] if len(sys.argv) > 1 else "S3"
N, W, H, deg, seed, ncam = SCENES[scene]
model = GaussianModel.from_raw(make_gaussians(N, deg, seed), deg, device=dev); model.active_sh_degree = deg
class Pipe: convert_SHs_python = False; compute_cov3D_python = False; debug = False; fused_activations = True
cam = ring_cameras(ncam, W, H, seed, device=dev)[0]
R.KEEP_LAST_FORWARD = True
render(cam, model, Pipe(), torch.zeros(3, device=dev))
st, n_, w_, h_ = R.LAST_FORWARD
L = _C.lib()
xy = np.zeros((N, 2), np.float32); conic = np.zeros((N, 4), np.float32); rect = np.zeros((N, 4), np.int32)
p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
_C.check(L.scorp_gs3d_debug_geom(st.data_ptr(), N, W, H, p(xy), None, p(conic), None, p(rect), R._stream()), "geom")
xy, conic, rect = (torch.from_numpy(a).to(dev) for a in (xy, conic, rect))
tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
w = (rect[:, 2] - rect[:, 0]).clamp(min=0).long(); h = (rect[:, 3] - rect[:, 1]).clamp(min=0).long()
cnt = w * h
sid = torch.repeat_interleave(torch.arange(N, device=dev), cnt)
first = torch.cumsum(cnt, 0) - cnt
k = torch.arange(sid.numel(), device=dev) - first[sid]
tx = rect[sid, 0].long() + k % w[sid]; ty = rect[sid, 1].long() + k // w[sid]
print("rect pairs", sid.numel())
A, B, C, o = conic[sid, 0], conic[sid, 1], conic[sid, 2], conic[sid, 3]
kcut = 1.01 * 2.0 * torch.log((255.0 * o).clamp(min=1.0)) + 0.02
cx, cy = xy[sid, 0], xy[sid, 1]

def box_min(bx0, bx1, by0, by1):
    x0, x1, y0, y1 = bx0 - cx, bx1 - cx, by0 - cy, by1 - cy
    inside = (x0 <= 0) & (x1 >= 0) & (y0 <= 0) & (y1 >= 0)
    best = torch.full_like(cx, 3.4e38)
    for xe in (x0, x1):
        dy = torch.minimum(torch.maximum(-B * xe / C, y0), y1)
        best = torch.minimum(best, A * xe * xe + 2 * B * xe * dy + C * dy * dy)
    for ye in (y0, y1):
        dx = torch.minimum(torch.maximum(-B * ye / A, x0), x1)
        best = torch.minimum(best, A * dx * dx + 2 * B * dx * ye + C * ye * ye)
    return torch.where(inside, torch.zeros_like(best), best)

tile_hit = box_min(tx * 16.0, tx * 16.0 + 15, ty * 16.0, ty * 16.0 + 15) <= kcut
print("tile pairs (D)", int(tile_hit.sum()))
nblk = tiles_x * tiles_y * 4
len8 = torch.zeros(nblk, device=dev)
len4 = torch.zeros(nblk * 4, device=dev)
len82 = torch.zeros(nblk * 2, device=dev)   # 8x4 half-blocks (two per 8x8)
tile = ty * tiles_x + tx
for q in range(4):
    bx = tx * 16.0 + (q & 1) * 8; by = ty * 16.0 + (q >> 1) * 8
    h8 = tile_hit & (box_min(bx, bx + 7, by, by + 7) <= kcut)
    len8.index_add_(0, tile * 4 + q, h8.float())
    for s in range(4):
        sx = bx + (s & 1) * 4; sy = by + (s >> 1) * 4
        h4 = h8 & (box_min(sx, sx + 3, sy, sy + 3) <= kcut)
        len4.index_add_(0, (tile * 4 + q) * 4 + s, h4.float())
    for s in range(2):
        sy = by + s * 4
        h2 = h8 & (box_min(bx, bx + 7, sy, sy + 3) <= kcut)
        len82.index_add_(0, (tile * 4 + q) * 2 + s, h2.float())
l4 = len4.view(nblk, 4); l2 = len82.view(nblk, 2)
print("block hits (sum len8)", int(len8.sum()), "mean", float(len8.mean()))
print("4x4: sum of sub-hits", int(l4.sum()), "sub-hits per block hit", float(l4.sum() / len8.sum()))
print("4x4: sum over blocks of max sub-list", int(l4.max(1).values.sum()), "ratio to block hits", float(l4.max(1).values.sum() / len8.sum()))
print("8x4: sub-hits per block hit", float(l2.sum() / len8.sum()), "max ratio", float(l2.max(1).values.sum() / len8.sum()))
# rounding to groups of 8 per wave
g8 = torch.ceil(len8 / 8).sum(); g4 = torch.ceil(l4.max(1).values / 8).sum()
print("groups of 8: now", int(g8), "4x4 lists", int(g4), "ratio", float(g4 / g8))
