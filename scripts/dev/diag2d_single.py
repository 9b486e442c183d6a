import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import fuzz_cases
from tests.test_gs2d_gpu import hip_render2d
from tests.test_oracle2d_cpu import make_case2d
from oracle.gs_oracle import OracleRender2D
dev = torch.device('cuda:0')
for k, gid in ((2, 1322), (7, 3836)):
    case = fuzz_cases("2d", 32, 20261004)[k]
    kw, _ = make_case2d(**case)
    one = dict(kw)
    for key in ("means3D", "opacities", "shs", "scales", "rotations"):
        if one.get(key) is not None:
            one[key] = np.ascontiguousarray(kw[key][gid:gid + 1])
    o = OracleRender2D(np.float32, **one); o64 = OracleRender2D(np.float64, **one)
    (color, radii, allmap), t = hip_render2d(one, dev)
    am = allmap.detach().cpu().numpy()
    a_h, a_o, a_64 = am[1], o.allmap[1], o64.allmap[1]
    ys, xs = np.nonzero((a_o > 0) | (a_h > 0))
    print("case", k, "gaussian", gid, "radius hip", int(radii[0]), "oracle", int(o.radii[0]), "opacity", float(one["opacities"][0]), "scales", one["scales"][0])
    for y, x in zip(ys, xs):
        flag = "" if abs(a_h[y, x] - a_o[y, x]) < 1e-5 else "   <<<<"
        print(f"   px ({x},{y}) alpha hip {a_h[y,x]:.7f} f32 {a_o[y,x]:.7f} f64 {a_64[y,x]:.7f}{flag}")
