"""One surfel of a fuzz case alone with its transform GIVEN (transmat_precomp = the fp32 oracle's T): the float64 build of
the oracle is then the exact answer for the inputs every implementation sees, and the HIP path and the fp32 oracle can be
held against it on equal terms.  DIAG_PICKS = seed:N:case:id,..."""
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import fuzz_cases
from tests.test_gs2d_gpu import hip_render2d
from tests.test_oracle2d_cpu import make_case2d
from oracle.gs_oracle import OracleRender2D
dev = torch.device('cuda:0')
for seed, nn, k, gid in [tuple(int(v) for v in p.split(":")) for p in os.environ["DIAG_PICKS"].split(",")]:
    case = fuzz_cases("2d", nn, seed)[k]
    kw, _ = make_case2d(**case)
    if gid < 0:   # the surfel whose means3D gradient is furthest from the fp32 oracle's in the whole scene
        (c0, _, a0), t0 = hip_render2d(kw, dev)
        rng0 = np.random.default_rng(case["seed"] + 99)
        wc0 = rng0.normal(0, 1, tuple(c0.shape)).astype(np.float32); wa0 = rng0.normal(0, 1, tuple(a0.shape)).astype(np.float32); wa0[5] *= 0.1
        ((c0 * torch.tensor(wc0, device=dev)).sum() + (a0 * torch.tensor(wa0, device=dev)).sum()).backward()
        gs = OracleRender2D(np.float32, **kw).backward(wc0, wa0)
        gid = int(np.abs(t0["means3D"].grad.detach().cpu().numpy() - gs["means3D"]).max(1).argmax())
    one = dict(kw)
    for key in ("means3D", "opacities", "shs", "scales", "rotations"):
        if one.get(key) is not None:
            one[key] = np.ascontiguousarray(kw[key][gid:gid + 1])
    T32 = OracleRender2D(np.float32, **one).geom()["T"].astype(np.float32)
    given = dict(one); given["scales"] = None; given["rotations"] = None; given["transmat_precomp"] = T32
    o32, o64 = OracleRender2D(np.float32, **given), OracleRender2D(np.float64, **given)
    (color, radii, allmap), t = hip_render2d(given, dev)
    am = allmap.detach().cpu().numpy()
    e_h, e_o = np.abs(am[1] - o64.allmap[1]).max(), np.abs(o32.allmap[1] - o64.allmap[1]).max()
    print("seed", seed, "case", k, "surfel", gid, "radii", int(radii[0]), int(o32.radii[0]), "| alpha: max |hip - exact|", e_h, " max |f32 oracle - exact|", e_o)
    rng = np.random.default_rng(case["seed"] + 99)
    c = color.detach().cpu().numpy()
    wc = rng.normal(0, 1, c.shape).astype(np.float32); wa = rng.normal(0, 1, am.shape).astype(np.float32); wa[5] *= 0.1
    ((color * torch.tensor(wc, device=dev)).sum() + (allmap * torch.tensor(wa, device=dev)).sum()).backward()
    g32, g64 = o32.backward(wc, wa), o64.backward(wc, wa)
    for nm, key in (("transmat", "transmat_precomp"), ("opacities", "opacities"), ("means3D", "means3D")):
        got = t[key].grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
        r32, r64 = np.asarray(g32[nm], np.float64).reshape(-1), np.asarray(g64[nm], np.float64).reshape(-1)
        sc = max(np.abs(r64).max(), 1e-300)
        print(f"   grad {nm}: |hip - exact| / scale {np.abs(got - r64).max() / sc:.3e}   |f32 oracle - exact| / scale {np.abs(r32 - r64).max() / sc:.3e}")
