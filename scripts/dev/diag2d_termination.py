"""A fuzz case in full: pixels inside one surfel's rectangle whose accumulated alpha (allmap channel 1 = 1 - T_final) parts
from the oracle's by more than rounding - the T < 1e-4 termination decided differently.  DIAG_PICKS = seed:N:case:id,..."""
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import fuzz_cases
from tests.test_gs2d_gpu import hip_render2d
from tests.test_oracle2d_cpu import make_case2d
from oracle.gs_oracle import OracleRender2D
dev = torch.device('cuda:0')
for seed, nn, k, gid in [tuple(int(v) for v in p.split(":")) for p in os.environ["DIAG_PICKS"].split(",")]:
    kw, _ = make_case2d(**fuzz_cases("2d", nn, seed)[k])
    o = OracleRender2D(np.float32, **kw); o64 = OracleRender2D(np.float64, **kw)
    (color, radii, allmap), t = hip_render2d(kw, dev, requires_grad=False)
    a_h, a_o, a_64 = allmap.detach().cpu().numpy()[1].astype(np.float64), o.allmap[1].astype(np.float64), o64.allmap[1]
    x0, y0, x1, y1 = o.geom()["rect"][gid]
    print("seed", seed, "case", k, "surfel", gid, "rect (tiles)", (x0, y0, x1, y1))
    for y in range(y0 * 16, min(y1 * 16, kw["H"])):
        for x in range(x0 * 16, min(x1 * 16, kw["W"])):
            th, to, t64 = 1 - a_h[y, x], 1 - a_o[y, x], 1 - a_64[y, x]
            if abs(th - to) > 2e-6 or abs(to - t64) > 2e-6:
                print(f"   px ({x},{y}) T_final hip {th:.4e} f32 {to:.4e} f64 {t64:.4e}")
