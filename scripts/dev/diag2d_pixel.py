"""One surfel of a fuzz case alone, one pixel: where does the HIP path's alpha part from the oracle's - in the transform T
(preprocess) or in the evaluation of the intersection?  DIAG_PICKS = seed:N:case:id:px:py,..."""
import sys, os, ctypes
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import fuzz_cases
from tests.test_oracle2d_cpu import make_case2d
from oracle.gs_oracle import OracleRender2D
from scorp_amd import _C, rasterizer3d as R
dev = torch.device('cuda:0')
L = _C.lib()


def alpha_ref(T, xy, op, px, py, dt):
    T = T.astype(dt); Tu, Tv, Tw = T[0:3], T[3:6], T[6:9]
    k = dt(px) * Tw - Tu; l = dt(py) * Tw - Tv
    p = np.array([k[1] * l[2] - k[2] * l[1], k[2] * l[0] - k[0] * l[2], k[0] * l[1] - k[1] * l[0]], dt)
    s = p[:2] / p[2]
    rho3d = s[0] * s[0] + s[1] * s[1]
    rho2d = dt(2.0) * ((dt(xy[0]) - dt(px)) ** 2 + (dt(xy[1]) - dt(py)) ** 2)
    return float(dt(op) * np.exp(dt(-0.5) * min(rho3d, rho2d))), float(rho3d), float(rho2d), p


def alpha_lin(T, xy, op, px, py):   # the HIP path's linear form, fp32, expanded about the centre of the pixel's 8x8 block
    f = np.float32
    T = T.astype(f); Tu, Tv, Tw = T[0:3], T[3:6], T[6:9]
    bxc, byc = f((px // 8) * 8 + 3.5), f((py // 8) * 8 + 3.5)
    fma = lambda a, b, c: f(np.float64(a) * np.float64(b) + np.float64(c))
    pa = [fma(Tv[(i + 1) % 3], Tw[(i + 2) % 3], -f(Tv[(i + 2) % 3] * Tw[(i + 1) % 3])) for i in range(3)]
    pb = [fma(Tw[(i + 1) % 3], Tu[(i + 2) % 3], -f(Tw[(i + 2) % 3] * Tu[(i + 1) % 3])) for i in range(3)]
    kc = [fma(bxc, Tw[i], -Tu[i]) for i in range(3)]; lc = [fma(byc, Tw[i], -Tv[i]) for i in range(3)]
    pc = [fma(kc[(i + 1) % 3], lc[(i + 2) % 3], -f(kc[(i + 2) % 3] * lc[(i + 1) % 3])) for i in range(3)]
    qx, qy = f(px) - bxc, f(py) - byc
    p = [fma(pa[i], qx, fma(pb[i], qy, pc[i])) for i in range(3)]
    # the same in float64 from the same T (what exact arithmetic would give for each piece)
    d = np.float64
    T6 = T.astype(d); Tu6, Tv6, Tw6 = T6[0:3], T6[3:6], T6[6:9]
    pa6, pb6 = np.cross(Tv6, Tw6), np.cross(Tw6, Tu6)
    pc6 = np.cross(d(bxc) * Tw6 - Tu6, d(byc) * Tw6 - Tv6)
    s0, s1 = f(p[0] / p[2]), f(p[1] / p[2])
    rho3d = fma(s0, s0, f(s1 * s1))
    return float(rho3d), np.array(pa), pa6, np.array(pb), pb6, np.array(pc), pc6, np.array(p)


picks = [tuple(int(v) for v in p.split(":")) for p in os.environ["DIAG_PICKS"].split(",")]
for seed, nn, k, gid, px, py in picks:
    kw, _ = make_case2d(**fuzz_cases("2d", nn, seed)[k])
    one = dict(kw)
    for key in ("means3D", "opacities", "shs", "scales", "rotations"):
        if one.get(key) is not None:
            one[key] = np.ascontiguousarray(kw[key][gid:gid + 1])
    o = OracleRender2D(np.float32, **one); o64 = OracleRender2D(np.float64, **one)
    T_ = lambda a: None if a is None else torch.tensor(a, device=dev)
    N, W, H = 1, one["W"], one["H"]
    s = R.GaussianRasterizationSettings(H, W, one["tanfovx"], one["tanfovy"], T_(one["bg"]), one.get("scale_modifier", 1.0), T_(one["view"]),
                                        T_(one["proj"]), one["sh_degree"], T_(one["campos"]), False, True)
    ten = {kk: T_(one.get(kk)) for kk in ("means3D", "shs", "opacities", "scales", "rotations")}
    keep = []
    args = R._inputs_struct(s, ten["means3D"], ten["shs"], None, ten["opacities"], ten["scales"], ten["rotations"], None, keep)
    sb = L.scorp_gs2d_state_bytes(N, W, H)
    state = torch.empty(sb, dtype=torch.uint8, device=dev); radii = torch.empty(N, dtype=torch.int32, device=dev)
    _C.check(L.scorp_gs2d_preprocess(ctypes.byref(args), R._ptr(radii), R._ptr(state), sb, R._stream()), "pre")
    n = ctypes.c_uint64(); _C.check(L.scorp_gs3d_num_pairs(R._ptr(state), R._stream(), ctypes.byref(n)), "num")
    pairs = torch.empty(L.scorp_gs3d_pairs_bytes(max(n.value, 1)), dtype=torch.uint8, device=dev)
    color = torch.empty(3, H, W, device=dev); allmap = torch.empty(7, H, W, device=dev)
    _C.check(L.scorp_gs2d_render(ctypes.byref(args), R._ptr(state), R._ptr(pairs), max(n.value, 1), R._ptr(color), R._ptr(allmap), R._stream()), "render")
    Tm = np.zeros((N, 9), np.float32); xy = np.zeros((N, 2), np.float32); depth = np.zeros(N, np.float32)
    no = np.zeros((N, 4), np.float32); rgb = np.zeros((N, 3), np.float32); rect = np.zeros((N, 4), np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _C.check(L.scorp_gs2d_debug_geom(state.data_ptr(), N, W, H, p(Tm), p(xy), p(depth), p(no), p(rgb), p(rect), R._stream()), "geom")
    g, g64 = o.geom(), o64.geom()
    op = float(no[0, 3])
    print(f"seed {seed} case {k} surfel {gid} pixel ({px},{py})  opacity {op}")
    if px < 0:   # scan the surfel's rectangle for pixels on the low-pass switch rho3d == rho2d (exact evaluation of the fp32 T)
        x0, y0, x1, y1 = rect[0]
        for yy in range(max(y0 * 16, 0), min(y1 * 16, H)):
            for xx in range(max(x0 * 16, 0), min(x1 * 16, W)):
                a = alpha_ref(Tm[0], g64["xy"][0], op, xx, yy, np.float64); b = alpha_ref(Tm[0], g["xy"][0], op, xx, yy, np.float32)
                if a[0] >= 1 / 255 and abs(a[1] - a[2]) < 2e-3 * max(a[2], 1e-9):
                    print(f"   tie at ({xx},{yy}): exact rho3d {a[1]:.7f} rho2d {a[2]:.7f} | reference fp32 order rho3d {b[1]:.7f} rho2d {b[2]:.7f} alpha {a[0]:.5f}")
        continue
    print("   centre hip", repr(xy[0]), "f32", repr(g["xy"][0]), "f64", repr(g64["xy"][0]))
    for nm, c in (("hip", xy[0]), ("f32", g["xy"][0])):
        r2 = np.float32(2.0) * (np.float32(c[0] - np.float32(px)) ** 2 + np.float32(c[1] - np.float32(py)) ** 2)
        print(f"   rho2d in fp32 from the {nm} centre: {float(r2):.9f}")
    print("   T hip   ", Tm[0]); print("   T f32   ", g["T"][0]); print("   T f64   ", g64["T"][0])
    print("   rel |T hip - T f64| / |T f64| per entry", np.abs(Tm[0] - g64["T"][0]) / np.maximum(np.abs(g64["T"][0]), 1e-30))
    print("   rel |T f32 - T f64| / |T f64| per entry", np.abs(g["T"][0] - g64["T"][0]) / np.maximum(np.abs(g64["T"][0]), 1e-30))
    print("   alpha rendered: hip", float(allmap[1, py, px]), "f32", float(o.allmap[1, py, px]), "f64", float(o64.allmap[1, py, px]))
    for nm, T in (("hip T", Tm[0]), ("f32 T", g["T"][0]), ("f64 T", g64["T"][0])):
        a64 = alpha_ref(T, g64["xy"][0], op, px, py, np.float64); a32 = alpha_ref(T, g["xy"][0], op, px, py, np.float32)
        print(f"   reference formula on {nm}: in f64 alpha {a64[0]:.8f} rho3d {a64[1]:.6f} rho2d {a64[2]:.6f} | in f32 alpha {a32[0]:.8f} rho3d {a32[1]:.6f}   p(f64) {a64[3]}")
    r, pa, pa6, pb, pb6, pc, pc6, pv = alpha_lin(Tm[0], xy[0], op, px, py)
    print("   linear form on hip T (fp32 emulation): rho3d", r, " p", pv)
    print("     pa", pa, "exact", pa6, "rel", np.abs(pa - pa6) / np.abs(pa6))
    print("     pb", pb, "exact", pb6, "rel", np.abs(pb - pb6) / np.abs(pb6))
    print("     pc", pc, "exact", pc6, "rel", np.abs(pc - pc6) / np.abs(pc6))
