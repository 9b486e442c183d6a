"""One 2-D fuzz case (SCORP_FUZZ_SEED / index): which surfel's gradient is furthest from the oracle's, and what the float64
oracle says about it.   python scripts/dev/diag2d_case.py SEED K"""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import fuzz_cases, image_weights
from tests.test_gs2d_gpu import hip_render2d
from tests.test_oracle2d_cpu import make_case2d
from oracle.gs_oracle import OracleRender2D
dev = torch.device('cuda:0')
seed, k = int(sys.argv[1]), int(sys.argv[2])
case = fuzz_cases("2d", 160, seed)[k]
print(case)
kw, _ = make_case2d(**case)
o = OracleRender2D(np.float32, **kw); o64 = OracleRender2D(np.float64, **kw)
(color, radii, allmap), t = hip_render2d(kw, dev)
rng = np.random.default_rng(case["seed"] + 4242)
H, W = kw["H"], kw["W"]
wc = rng.normal(0, 1, (3, H, W)).astype(np.float32); wm = rng.normal(0, 1, (7, H, W)).astype(np.float32)
((color * torch.tensor(wc, device=dev)).sum() + (allmap * torch.tensor(wm, device=dev)).sum()).backward()
g32 = o.backward(wc, wm); g64 = o64.backward(wc.astype(np.float64), wm.astype(np.float64))
for name in ("means3D", "scales", "rotations", "opacities"):
    got = t[name].grad.detach().cpu().numpy().reshape(g32[name].shape)
    e = np.abs(got - g32[name]).reshape(got.shape[0], -1).max(1)
    e64 = np.abs(g32[name] - g64[name]).reshape(got.shape[0], -1).max(1)
    h64 = np.abs(got - g64[name]).reshape(got.shape[0], -1).max(1)
    worst = np.argsort(-e)[:3]
    sc = np.abs(g32[name]).max()
    print(name, "scale", sc)
    for i in worst:
        print(f"   surfel {i}: |hip - f32| {e[i]/sc:.3e}  |f32 - f64| {e64[i]/sc:.3e}  |hip - f64| {h64[i]/sc:.3e}   radius {int(radii[i])} opacity {float(kw['opacities'][i]):.4f} scales {kw['scales'][i]}")

if len(sys.argv) > 3:
    gid = int(sys.argv[3])
    one = dict(kw)
    for key in ("means3D", "opacities", "shs", "scales", "rotations"):
        if one.get(key) is not None:
            one[key] = np.ascontiguousarray(kw[key][gid:gid + 1])
    o1 = OracleRender2D(np.float32, **one); o1_64 = OracleRender2D(np.float64, **one)
    (c1, r1, a1), t1 = hip_render2d(one, dev)
    ((c1 * torch.tensor(wc, device=dev)).sum() + (a1 * torch.tensor(wm, device=dev)).sum()).backward()
    h32 = o1.backward(wc, wm); h64 = o1_64.backward(wc.astype(np.float64), wm.astype(np.float64))
    print("ALONE: image max |hip - f32|", float(np.abs(c1.detach().cpu().numpy() - o1.color).max()), "allmap", float(np.abs(a1.detach().cpu().numpy() - o1.allmap).max()))
    print("view", kw["view"], "\nmeans3D", one["means3D"], "rot", one["rotations"], "scales", one["scales"])
    for name in ("means3D", "scales", "rotations", "opacities"):
        got = t1[name].grad.detach().cpu().numpy().reshape(h32[name].shape)
        print(f"  {name}: hip {got.ravel()}  f32 {h32[name].ravel()}  f64 {h64[name].ravel()}")
    am_h, am_o = a1.detach().cpu().numpy(), o1.allmap
    d = np.abs(am_h - am_o)
    for ch in range(7):
        print("  allmap ch", ch, "max diff", d[ch].max(), "at", np.unravel_index(d[ch].argmax(), d[ch].shape), "covered px", int((am_o[1] > 0).sum()))
    a_h, a_o, a_64 = am_h[1], o1.allmap[1], o1_64.allmap[1]
    ys, xs = np.nonzero((a_o > 0) | (a_h > 0) | (a_64 > 0))
    for y, x in zip(ys, xs):
        flag = "" if abs(a_h[y, x] - a_o[y, x]) < 1e-5 else "   <<<<"
        print(f"   px ({x},{y}) alpha hip {a_h[y,x]:.7f} f32 {a_o[y,x]:.7f} f64 {a_64[y,x]:.7f}{flag}")
