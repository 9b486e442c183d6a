"""One surfel of a fuzz case rendered alone (HIP, fp32 oracle, fp64 oracle): per-pixel alpha, flagged where the HIP path and
the oracle disagree, and pixels whose alpha sits within 2e-4 relative of the 1/255 threshold.  DIAG_PICKS = seed:N:case:id,..."""
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import fuzz_cases
from tests.test_gs2d_gpu import hip_render2d
from tests.test_oracle2d_cpu import make_case2d
from oracle.gs_oracle import OracleRender2D
dev = torch.device('cuda:0')
picks = [tuple(int(v) for v in p.split(":")) for p in os.environ.get("DIAG_PICKS", "20261004:32:2:1322,20261004:32:7:3836").split(",")]
for seed, nn, k, gid in picks:
    case = fuzz_cases("2d", nn, seed)[k]
    kw, _ = make_case2d(**case)
    one = dict(kw)
    for key in ("means3D", "opacities", "shs", "scales", "rotations"):
        if one.get(key) is not None:
            one[key] = np.ascontiguousarray(kw[key][gid:gid + 1])
    o = OracleRender2D(np.float32, **one); o64 = OracleRender2D(np.float64, **one)
    (color, radii, allmap), t = hip_render2d(one, dev)
    am = allmap.detach().cpu().numpy()
    a_h, a_o, a_64 = am[1], o.allmap[1], o64.allmap[1]
    ys, xs = np.nonzero((a_o > 0) | (a_h > 0) | (a_64 > 0))
    print("seed", seed, "case", k, "surfel", gid, "radius hip", int(radii[0]), "oracle", int(o.radii[0]), "opacity", float(one["opacities"][0]),
          "scales", one["scales"][0], "pixels", len(ys))
    edge = 0
    for y, x in zip(ys, xs):
        vals = (a_h[y, x], a_o[y, x], a_64[y, x])
        near = any(abs(v * 255.0 - 1.0) < 2e-4 for v in vals)
        differ = abs(a_h[y, x] - a_o[y, x]) >= 1e-5
        edge += near
        if near or differ:
            print(f"   px ({x},{y}) alpha hip {vals[0]:.8f} f32 {vals[1]:.8f} f64 {vals[2]:.8f}  x255: {vals[0]*255:.6f} {vals[1]*255:.6f} {vals[2]*255:.6f}"
                  f"{'   <<<< differ' if differ else ''}{'   [on the 1/255 edge]' if near else ''}")
    print("   pixels on the edge:", edge)
    rng = np.random.default_rng(case["seed"] + 99)
    c = color.detach().cpu().numpy()
    wc = rng.normal(0, 1, c.shape).astype(np.float32); wa = rng.normal(0, 1, am.shape).astype(np.float32); wa[5] *= 0.1
    ((color * torch.tensor(wc, device=dev)).sum() + (allmap * torch.tensor(wa, device=dev)).sum()).backward()
    g, g64 = o.backward(wc, wa), o64.backward(wc, wa)
    for nm in ("means3D", "means2D", "scales"):
        print("   alone: grad", nm, "hip", t[nm].grad.detach().cpu().numpy().reshape(-1)[:3], "f32", np.asarray(g[nm]).reshape(-1)[:3], "f64", np.asarray(g64[nm]).reshape(-1)[:3])
