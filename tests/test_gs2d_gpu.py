"""GPU parity tests of the 2DGS (surfel) path against the CPU oracle, through the diff_surfel_rasterization shim."""
import ctypes
import math

import os

import numpy as np
import pytest
import torch

from tests.test_oracle2d_cpu import make_case2d

pytestmark = pytest.mark.gpu
IMG_L1_TOL = 1e-4
GRAD_REL_TOL = 2e-3
GRAD_L1_TOL = 1e-4     # sum |delta| / sum |ref| per gradient tensor (north_star's 1e-4 L1)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from scorp_amd import _C
    _C.lib()
    return torch.device("cuda:0")


PRECISIONS = ["split", "exact_fp32"]   # both forms of the backward's pixel -> surfel reduction (scorp_gs2d_backward_ex)


def hip_render2d(kw, dev, requires_grad=True, precision=None):
    """`precision`: rasterizer3d.backward_precision mode recorded with this forward (None: the library's default, the split)."""
    from scorp_amd.refcall import render2d_reference_call
    if precision is None:
        return render2d_reference_call(kw, dev, requires_grad=requires_grad)
    from scorp_amd.rasterizer3d import backward_precision
    with backward_precision(precision):
        return render2d_reference_call(kw, dev, requires_grad=requires_grad)


def assert_radii_match(got, ref):
    """Bit-exact, except that a surfel grazing the camera plane has an extent of 1e4+ px whose ceil() may flip on the
    last ulp of a cancelling difference (cx^2 - sum f Tu^2); its tile rectangle is clamped to the image either way."""
    bad = got != ref
    assert ((got > 0) == (ref > 0)).all()
    if bad.any():
        assert bad.mean() < 1e-3 and (np.abs(got - ref)[bad] <= 1).all() and (ref[bad] > 4096).all()


CASES = {
    "sh3_bg": dict(N=3000, W=160, H=120, deg=3, seed=1, bg=(0.2, 0.5, 0.7)),
    "sh2_ragged": dict(N=4000, W=137, H=91, deg=2, seed=2, log_scale=math.log(0.06)),
    "sh0_maxdeg0": dict(N=2000, W=128, H=72, deg=0, seed=9, max_deg=0),
    "precomp_color": dict(N=3000, W=128, H=66, deg=0, seed=3, precomp_color=True),
    "inside_cloud": dict(N=5000, W=80, H=80, deg=3, seed=7, radius=1.5),
    "tiny_surfels": dict(N=20000, W=256, H=192, deg=1, seed=5, log_scale=math.log(0.006)),   # low-pass branch
    "scale_mod": dict(N=2000, W=96, H=96, deg=1, seed=8, scale_modifier=1.4),
    # large surfels near the far plane (depth ~70 of kFarZ = 100) under depth / alpha / normal upstream gradients: the products
    # depth x 1 / p.z x r^2 of the backward's fp16-split values are at their largest here (gs2d.hip, k2TargetExp)
    "far_depth": dict(N=1500, W=160, H=120, deg=1, seed=21, radius=70.0, log_scale=math.log(2.0)),
}


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("name", list(CASES))
def test_forward_backward_parity_2d(name, precision, dev):
    _parity_2d(CASES[name], dev, precision=precision)


def _parity_2d(case, dev, report=None, outlier_gaussians=0, kw=None, tie_outliers=0, f64_report=None, precision=None):
    """`case`: arguments of make_case2d - or, with `kw` given (a full-size scene), only its "seed" is used.
    `f64_report` (a dict): every gradient tensor is ALSO held against the float64 build of the oracle
    (tests.util.assert_no_further_from_f64) and (relL1(HIP, f64), relL1(oracle32, f64)) is left there per tensor."""
    from oracle.gs_oracle import OracleRender2D
    if kw is None:
        kw, _ = make_case2d(**case)
    o = OracleRender2D(np.float32, **kw)
    assert o.num_pairs > 0
    out, t = hip_render2d(kw, dev, precision=precision)
    color, radii, allmap = out
    assert_radii_match(radii.cpu().numpy(), o.radii)
    c, am = color.detach().cpu().numpy(), allmap.detach().cpu().numpy()
    assert np.abs(c - o.color).mean() < IMG_L1_TOL and np.abs(c - o.color).max() < 2e-2
    for ch in range(7):
        scale = max(np.abs(o.allmap[ch]).max(), 1.0)
        assert np.abs(am[ch] - o.allmap[ch]).mean() / scale < IMG_L1_TOL, f"allmap channel {ch}"
    rng = np.random.default_rng(case["seed"] + 99)
    wc = rng.normal(0, 1, c.shape).astype(np.float32)
    wa = rng.normal(0, 1, am.shape).astype(np.float32)
    wa[5] *= 0.1                                    # median depth: a discontinuous selection, keep its weight modest
    ((color * torch.tensor(wc, device=dev)).sum() + (allmap * torch.tensor(wa, device=dev)).sum()).backward()
    g = o.backward(wc, wa)

    from tests.util import assert_grad_close
    cache = {}

    def ref64(nm):
        if "g" not in cache:   # the oracle's own band (float64 build, two perturbed fp32 runs): only for a tensor that misses
            from tests.test_gs3d_gpu import perturbed
            cache["g"] = [g64() if f64_report is not None else OracleRender2D(np.float64, **kw).backward(wc, wa),
                          OracleRender2D(np.float32, **perturbed(kw, +1)).backward(wc, wa),
                          OracleRender2D(np.float32, **perturbed(kw, -1)).backward(wc, wa)]
        return [x[nm] for x in cache["g"]]

    def g64():
        if "g64" not in cache:
            cache["g64"] = OracleRender2D(np.float64, **kw).backward(wc, wa)
        return cache["g64"]

    def close(nm, got, key):
        got = got.detach().cpu().numpy().reshape(g[key].shape).copy()
        refs = [g[key]]
        if f64_report is not None:
            from tests.util import assert_no_further_from_f64
            f64_report[nm] = assert_no_further_from_f64(key, got, g[key], g64()[key])
        if outlier_gaussians:   # a NAMED exception (FUZZ_2D_EXCEPTIONS): the worst few surfels are checked loosely, apart
            err = np.abs(got - g[key]).reshape(got.shape[0], -1).max(1)
            rows = np.argsort(-err)[:outlier_gaussians]
            assert err[rows].max() <= 5e-2 * np.abs(g[key]).max(), f"grad {nm}: an excepted surfel is off by more than 5 %"
            got[rows] = g[key][rows]
        e = assert_grad_close(key, got, refs[0], ref64, GRAD_REL_TOL, GRAD_L1_TOL, outliers=tie_outliers, outlier_tol=5e-2)
        if report is not None:
            report[nm] = e
    close("means3D", t["means3D"].grad, "means3D")
    close("means2D", t["means2D"].grad, "means2D")
    close("opacities", t["opacities"].grad, "opacities")
    close("scales", t["scales"].grad, "scales")
    close("rotations", t["rotations"].grad, "rotations")
    if t["shs"] is not None:
        close("shs", t["shs"].grad, "shs")
    else:
        close("colors", t["colors_precomp"].grad, "colors_precomp")


from tests.util import fuzz_cases  # noqa: E402
_FUZZ_SEED = int(os.environ.get("SCORP_FUZZ_SEED", "20261004"))   # (a longer, differently seeded draw for one-off soak runs)
FUZZ_2D = fuzz_cases("2d", int(os.environ.get("SCORP_FUZZ_N", "32")), _FUZZ_SEED)


# No named exceptions any more.  Round 1's linear form of the ray-surfel intersection, expanded about the image
# origin, needed two here (surfel 1322 of case 2 seen almost edge-on, surfel 3836 of case 7 far out on its long axis:
# scripts/dev/diag2d_single.py) and missed the tolerance in 7 of 300 differently seeded draws; expanded about the block
# centre, with the gradients gathered about the surfel's own centre (gs2d.hip, surfel_lin), all 32 pass as they are.  Of
# 4 320 differently seeded draws (SCORP_FUZZ_N=160, 27 seeds) fourteen miss, by one surfel each; of the first six:
# in five it is the fp32 ORACLE that is off, the sixth is a tie: CONDITIONING_PICKS and the test below them.
FUZZ_2D_EXCEPTIONS = {}


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("k", range(len(FUZZ_2D)))
def test_fuzz_parity_2d(k, precision, dev):
    """Randomised surfel cases (seeded), forward + backward against the 2-D oracle, same assertions as above."""
    _parity_2d(FUZZ_2D[k], dev, outlier_gaussians=FUZZ_2D_EXCEPTIONS.get(k, 0) if _FUZZ_SEED == 20261004 else 0,
               precision=precision)


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("seed,k", [(63, 122), (64, 140)])
def test_faint_hits_keep_their_bits(seed, k, precision, dev):
    """Two fuzz scenes whose means2D gradient is carried by a few low-pass pixels of faint or deeply covered surfels
    (alpha ~ 0.01, T down to 1e-4): the largest row of the tensor was 1e-3 / 1e-2 off while the backward's fp16 pairs rode on
    one power of two per wave, under their absolute floor of 2^-24; with the hit's own power of two (gs2d.hip, `sg`) it is
    within 1e-5 (scripts/dev/diag2d_rows.py prints the rows).  The all-fp32 form has no such floor to begin with."""
    _parity_2d(fuzz_cases("2d", 160, seed)[k], dev, precision=precision)


def _given_T(kw, rows=None):
    """The scene with its transforms GIVEN (transmat_precomp = the fp32 oracle's T, the reference's `cov3D_precomp` argument of
    the surfel rasterizer) - every implementation then starts from bit-identical T and the float64 build of the oracle is the
    exact answer for them.  `rows`: keep only these surfels."""
    from oracle.gs_oracle import OracleRender2D
    q = dict(kw)
    if rows is not None:
        for key in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations"):
            if q.get(key) is not None:
                q[key] = np.ascontiguousarray(q[key][rows])
    T32 = OracleRender2D(np.float32, **q).geom()["T"].astype(np.float32)
    q["scales"] = None; q["rotations"] = None; q["transmat_precomp"] = T32
    return q


def _weights(seed, c, am):
    rng = np.random.default_rng(seed + 99)
    wc = rng.normal(0, 1, c.shape).astype(np.float32)
    wa = rng.normal(0, 1, am.shape).astype(np.float32)
    wa[5] *= 0.1
    return wc, wa


# The surfels on which the first fifteen of twenty-seven differently seeded draws of 160 cases (SCORP_FUZZ_SEED = 1, 2, 61 .. 64,
# 71, 72, 81 .. 86, 777: 2 400 cases; 101 .. 112 added eight more of the same kinds, DESIGN.md section 2: 14 of 4 320 in all) left the
# HIP path outside the tolerance against the fp32 oracle - one surfel each, 6 cases of 2 400.  (seed, case, surfel):
# 82 / 115 / 159: ill-conditioned intersections (the centre column of T cancels against x Tw to 1e-4 of its terms);
# 68 (and seed 64's case 158): a pixel whose alpha is 1.000015 / 255.  The sixth, (71, 57, 79), is a TIE of the low-pass
# switch at the surfel's centre pixel: rho3d = 0.2234743 exactly (for the fp32 T; the HIP form gives 0.22347434, the
# reference's fp32 order 0.2234680) against rho2d = 0.2234723 from the fp32 centre, 0.2234823 from the float64 one - one ulp
# of cx = 95.86 moves rho2d by 4e-5.  It is not in the list: with T given, "exact" computes the centre in float64 and takes
# the other branch, as legitimately as the HIP path takes this one (scripts/dev/diag2d_pixel.py prints the four numbers).
CONDITIONING_PICKS = [(62, 82, 250), (1, 115, 5145), (64, 159, 418), (63, 68, 66)]


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("pick", CONDITIONING_PICKS, ids=lambda p: "seed%d_case%d_surfel%d" % p)
def test_fuzz_outliers_are_the_fp32_oracles_rounding_not_the_hip_paths(pick, precision, dev):
    """Each of those surfels alone, its transform given: against the EXACT answer for the same fp32 T (the float64 build) the
    HIP path's alpha map and gradients are closer than the fp32 oracle's - the reference's own order of operations in fp32
    (k = x Tw - Tu, p = k x l) loses 1e-4 .. 3e-2 there, the block-centred linear form (gs2d.hip, surfel_lin) 1e-6 .. 1e-4."""
    from oracle.gs_oracle import OracleRender2D
    seed, k, gid = pick
    case = fuzz_cases("2d", 160, seed)[k]
    kw, _ = make_case2d(**case)
    given = _given_T(kw, rows=slice(gid, gid + 1))
    o32, o64 = OracleRender2D(np.float32, **given), OracleRender2D(np.float64, **given)
    (color, radii, allmap), t = hip_render2d(given, dev, precision=precision)
    assert int(radii[0]) == int(o32.radii[0]) > 0
    am = allmap.detach().cpu().numpy()
    e_hip, e_o32 = np.abs(am[1] - o64.allmap[1]).max(), np.abs(o32.allmap[1] - o64.allmap[1]).max()
    assert e_hip <= max(1e-5, e_o32), f"alpha: HIP {e_hip:.3e} from exact, fp32 oracle {e_o32:.3e}"
    wc, wa = _weights(case["seed"], color.detach().cpu().numpy(), am)
    ((color * torch.tensor(wc, device=dev)).sum() + (allmap * torch.tensor(wa, device=dev)).sum()).backward()
    g32, g64 = o32.backward(wc, wa), o64.backward(wc, wa)
    for nm, key in (("transmat", "transmat_precomp"), ("opacities", "opacities")):
        got = t[key].grad.detach().cpu().numpy().reshape(-1).astype(np.float64)
        r32, r64 = np.asarray(g32[nm], np.float64).reshape(-1), np.asarray(g64[nm], np.float64).reshape(-1)
        sc = np.abs(r64).max()
        e_hip, e_o32 = np.abs(got - r64).max() / sc, np.abs(r32 - r64).max() / sc
        assert e_hip <= max(2e-4, e_o32), f"grad {nm}: HIP {e_hip:.3e} from exact, fp32 oracle {e_o32:.3e}"


def given_T_check(kw, seed, dev, report=None, precision=None):
    """The scene with its transforms given: per gradient tensor the HIP path's relative L1 distance to the exact answer (the
    float64 build on the same fp32 T) is at most max(1e-4, 1.25 x the fp32 oracle's own) - tests.util.assert_no_further_from_f64;
    `report[name]` = (relL1(HIP, exact), relL1(oracle32, exact), max-norm(HIP, exact), max-norm(oracle32, exact))."""
    from oracle.gs_oracle import OracleRender2D
    from tests.util import assert_no_further_from_f64, grad_errors
    given = _given_T(kw)
    o32, o64 = OracleRender2D(np.float32, **given), OracleRender2D(np.float64, **given)
    (color, radii, allmap), t = hip_render2d(given, dev, precision=precision)
    assert_radii_match(radii.cpu().numpy(), o32.radii)
    c, am = color.detach().cpu().numpy(), allmap.detach().cpu().numpy()
    assert np.abs(c - o64.color).mean() <= max(IMG_L1_TOL, 1.25 * np.abs(o32.color - o64.color).mean())
    wc, wa = _weights(seed, c, am)
    ((color * torch.tensor(wc, device=dev)).sum() + (allmap * torch.tensor(wa, device=dev)).sum()).backward()
    g32, g64 = o32.backward(wc, wa), o64.backward(wc, wa)
    names = [("transmat", "transmat_precomp"), ("opacities", "opacities"), ("means3D", "means3D")]
    names.append(("shs", "shs") if t["shs"] is not None else ("colors_precomp", "colors_precomp"))
    for nm, key in names:
        got = t[key].grad.detach().cpu().numpy()
        e = assert_no_further_from_f64(nm, got, g32[nm], g64[nm])
        if report is not None:
            report[nm] = e + (grad_errors(got, g64[nm])[0], grad_errors(g32[nm], g64[nm])[0])


@pytest.mark.parametrize("precision", PRECISIONS)
@pytest.mark.parametrize("k", list(range(8)) + [26])
def test_given_transforms_hip_is_no_further_from_exact_than_the_fp32_oracle(k, precision, dev):
    """Whole fuzz scenes (2 and 26 are the two of the 32 whose gradients pass against the fp32 oracle only through its
    indecision band) with the transforms given."""
    case = FUZZ_2D[k]
    given_T_check(make_case2d(**case)[0], case["seed"], dev, precision=precision)


def _grads_2d(kw, dev, mode, weights, reps=1):
    """`reps` backward passes on ONE forward issued under backward_precision(mode); per pass {leaf: gradient}."""
    (color, _, allmap), t = hip_render2d(kw, dev, precision=mode)
    loss = (color * weights[0]).sum() + (allmap * weights[1]).sum()
    res = []
    for r in range(reps):
        for v in t.values():
            if v is not None:
                v.grad = None
        loss.backward(retain_graph=r + 1 < reps)
        res.append({k: v.grad.detach().clone() for k, v in t.items() if v is not None and v.grad is not None})
    return res


@pytest.mark.parametrize("name", ["sh3_bg", "tiny_surfels", "far_depth", "inside_cloud"])
def test_split_backward_equals_exact_fp32_backward_2d(name, dev):
    """The default backward (eight values per hit as two fp16 terms under the hit's own power of two, fp16 MFMAs) against
    the all-fp32 form (fp32 values, fp32 MFMAs, no scales) on the same scene: every gradient tensor within 2e-5 relative L1
    and 1e-4 of its maximum; upstream gradients scaled by 1e-6 and 1e+4 exercise the split form's per-wave scale."""
    case = dict(CASES[name])
    kw, _ = make_case2d(**case)
    wc, wa = _weights(case["seed"], np.zeros((3, kw["H"], kw["W"])), np.zeros((7, kw["H"], kw["W"])))
    for scale in (1.0, 1e-6, 1e4):
        W2 = [torch.tensor(w * np.float32(scale), device=dev) for w in (wc, wa)]
        (gs,), (ge,) = _grads_2d(kw, dev, "split", W2), _grads_2d(kw, dev, "exact_fp32", W2)
        for k, ref in ge.items():
            got, ref = gs[k].double(), ref.double()
            assert torch.isfinite(got).all()
            l1 = float((got - ref).abs().sum() / ref.abs().sum().clamp_min(1e-300))
            mx = float((got - ref).abs().max() / ref.abs().max().clamp_min(1e-300))
            assert l1 < 2e-5 and mx < 1e-4, f"{k} (upstream x{scale:g}): rel L1 {l1:.2e}, max {mx:.2e}"


@pytest.mark.parametrize("mode", ["deterministic", "exact_fp32_deterministic"])
@pytest.mark.parametrize("name", ["sh3_bg", "tiny_surfels", "precomp_color", "inside_cloud"])
def test_deterministic_backward_2d_is_bit_reproducible_and_equals_the_atomic_form(name, mode, dev):
    """SCORP_BACKWARD_DETERMINISTIC on the surfel rasterizer (plain per-(block, hit) rows of twenty sums + an ordered
    per-surfel sum, no float atomics): two backward passes on one forward - what utils/mask.py:52,65,89 does before voting
    on gradient signs (:124) - give the same bits, and every gradient tensor equals the atomic form's within 2e-5 relative
    L1.  Both reduction forms."""
    case = dict(CASES[name])
    kw, _ = make_case2d(**case)
    wc, wa = _weights(case["seed"], np.zeros((3, kw["H"], kw["W"])), np.zeros((7, kw["H"], kw["W"])))
    W2 = [torch.tensor(w, device=dev) for w in (wc, wa)]
    d1, d2 = _grads_2d(kw, dev, mode, W2, reps=2)
    for k in d1:
        assert torch.equal(d1[k], d2[k]), f"{k}: two deterministic backward passes differ"
    (d3,) = _grads_2d(kw, dev, mode, W2)     # ... and so does a second forward + backward
    for k in d1:
        assert torch.equal(d1[k], d3[k]), f"{k}: a second deterministic forward + backward differs"
    (a1,) = _grads_2d(kw, dev, mode.replace("_deterministic", "").replace("deterministic", "split"), W2)
    for k, ref in a1.items():
        got = d1[k].double()
        l1 = float((got - ref.double()).abs().sum() / ref.double().abs().sum().clamp_min(1e-300))
        assert l1 < 2e-5, f"{k}: deterministic vs atomic rel L1 {l1:.2e}"


def test_edge_on_surfel_sliver_is_not_culled(dev):
    """A large surfel seen nearly edge-on projects to a sliver that owns ONE pixel every few dozen rows, far from its
    low-pass blob.  Its footprint conic (Mxx Myy ~ Mxy^2) has no digits left in fp32: the exact block culling must then
    stand back (preprocess2d_kernel, `sound`), or those pixels - alpha 0.24 in this case, found by a fuzz soak with seed
    777 - are dropped.  The surfel alone, every pixel the oracle covers."""
    from oracle.gs_oracle import OracleRender2D
    kw, _ = make_case2d(**fuzz_cases("2d", 160, 777)[54])
    one = dict(kw)
    for key in ("means3D", "opacities", "shs", "scales", "rotations"):
        one[key] = np.ascontiguousarray(kw[key][689:690])
    o = OracleRender2D(np.float32, **one)
    (color, radii, allmap), _ = hip_render2d(one, dev, requires_grad=False)
    a_h, a_o = allmap.detach().cpu().numpy()[1], o.allmap[1]
    far = (a_o > 0.02)
    assert far.sum() >= 10 and (np.nonzero(far)[0].max() - np.nonzero(far)[0].min()) > 40    # blob + the isolated pixels
    assert np.abs(a_h - a_o).max() < 2e-3, f"alpha off by {np.abs(a_h - a_o).max():.3e}"
    assert np.abs(color.detach().cpu().numpy() - o.color).max() < 2e-3


def test_stage_parity_2d(dev):
    """Surfel transforms, centres, normals match the oracle to rounding; radii and rectangles match exactly; the
    per-tile sorted lists are the oracle's with provably non-contributing pairs removed (exact footprint cull)."""
    from oracle.gs_oracle import OracleRender2D
    from scorp_amd import _C, rasterizer3d as R
    kw, _ = make_case2d(**CASES["sh3_bg"])
    o = OracleRender2D(np.float32, **kw)
    L = _C.lib()
    T = lambda a: None if a is None else torch.tensor(a, device=dev)
    N, W, H = kw["means3D"].shape[0], kw["W"], kw["H"]
    s = R.GaussianRasterizationSettings(H, W, kw["tanfovx"], kw["tanfovy"], T(kw["bg"]), 1.0, T(kw["view"]), T(kw["proj"]),
                                        kw["sh_degree"], T(kw["campos"]), False, True)
    ten = {k: T(kw.get(k)) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
    keep = []
    args = R._inputs_struct(s, ten["means3D"], ten["shs"], None, ten["opacities"], ten["scales"], ten["rotations"], None, keep)
    sb = L.scorp_gs2d_state_bytes(N, W, H)
    state = torch.empty(sb, dtype=torch.uint8, device=dev)
    radii = torch.empty(N, dtype=torch.int32, device=dev)
    _C.check(L.scorp_gs2d_preprocess(ctypes.byref(args), R._ptr(radii), R._ptr(state), sb, R._stream()), "pre")
    n = ctypes.c_uint64()
    _C.check(L.scorp_gs3d_num_pairs(R._ptr(state), R._stream(), ctypes.byref(n)), "num")
    assert 0 < n.value <= o.num_pairs
    pairs = torch.empty(L.scorp_gs3d_pairs_bytes(max(n.value, 1)), dtype=torch.uint8, device=dev)
    color = torch.empty(3, H, W, device=dev); allmap = torch.empty(7, H, W, device=dev)
    _C.check(L.scorp_gs2d_render(ctypes.byref(args), R._ptr(state), R._ptr(pairs), max(n.value, 1), R._ptr(color), R._ptr(allmap), R._stream()), "render")
    Tm = np.zeros((N, 9), np.float32); xy = np.zeros((N, 2), np.float32); depth = np.zeros(N, np.float32)
    no = np.zeros((N, 4), np.float32); rgb = np.zeros((N, 3), np.float32); rect = np.zeros((N, 4), np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    _C.check(L.scorp_gs2d_debug_geom(state.data_ptr(), N, W, H, p(Tm), p(xy), p(depth), p(no), p(rgb), p(rect), R._stream()), "geom")
    g = o.geom()
    assert_radii_match(radii.cpu().numpy(), o.radii)
    np.testing.assert_array_equal(rect, g["rect"])
    np.testing.assert_array_equal(depth, g["depth"])
    vis = o.radii > 0                                 # the oracle keeps T of surfels it culls later; compare visible ones
    np.testing.assert_allclose(Tm[vis], g["T"][vis], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(xy, g["xy"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(no, g["nrm_o"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rgb, g["rgb"], rtol=1e-5, atol=2e-6)
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    ts = np.zeros(tiles + 1, np.uint32); pl = np.zeros(max(n.value, 1), np.uint32)
    _C.check(L.scorp_gs2d_debug_tiles(state.data_ptr(), pairs.data_ptr(), max(n.value, 1), N, W, H, p(ts), p(pl), R._stream()), "tiles")
    ots, opl = o.tiles()
    # Same order as the oracle's lists; every (tile, surfel) pair the HIP path dropped must have alpha < 1/255 on all
    # 256 pixels of the tile, evaluated here with the reference formula in float64 from the oracle's own transforms.
    tiles_x = (W + 15) // 16
    T64, xy64, op64 = g["T"].astype(np.float64), g["xy"].astype(np.float64), g["nrm_o"][:, 3].astype(np.float64)
    dropped = 0
    for t in range(tiles):
        mine = pl[ts[t]:ts[t + 1]].astype(np.int64)
        ref = opl[ots[t]:ots[t + 1]].astype(np.int64)
        keep_ = np.isin(ref, mine)
        np.testing.assert_array_equal(ref[keep_], mine)
        miss = ref[~keep_]
        dropped += miss.size
        if miss.size:
            px = ((t % tiles_x) * 16 + np.arange(16))[None, None, :] + np.zeros((1, 16, 1))
            py = ((t // tiles_x) * 16 + np.arange(16))[None, :, None] + np.zeros((1, 1, 16))
            Tu, Tv, Tw = (T64[miss, 3 * r:3 * r + 3][:, None, None, :] for r in range(3))
            k = px[..., None] * Tw - Tu
            l = py[..., None] * Tw - Tv
            pv = np.cross(k, l)
            with np.errstate(divide="ignore", invalid="ignore"):
                rho3d = (pv[..., 0] ** 2 + pv[..., 1] ** 2) / pv[..., 2] ** 2
            rho3d = np.where(np.isfinite(rho3d), rho3d, np.inf)
            rho2d = 2.0 * ((xy64[miss, 0][:, None, None] - px) ** 2 + (xy64[miss, 1][:, None, None] - py) ** 2)
            alpha = op64[miss][:, None, None] * np.exp(-0.5 * np.minimum(rho3d, rho2d))
            assert (alpha < 1.0 / 255.0).all(), f"tile {t}: dropped a contributing surfel"
    assert dropped == o.num_pairs - n.value and dropped > 0


def test_render2d_dict_fused_vs_reference_convention_and_training_step(dev):
    """scorp_amd.renderer2d.render: the reference's nine keys; raw-leaf (fused) path == activated path incl. gradients;
    precomputed-transform branch (compute_cov3D_python) == in-kernel transform; one regularised training step runs."""
    from scorp_amd.fused_loss import fused_l1_ssim_loss
    from scorp_amd.gaussian_model import OptimizationParams
    from scorp_amd.renderer2d import GaussianModel2D, render, surfel_regularizers
    from scorp_amd.synthetic import make_gaussians, ring_cameras

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False
        depth_ratio = 1.0
        fused_activations = False

    raw = make_gaussians(5000 + 19, 3, 41, log_scale_mean=math.log(0.05), scale_dims=2)
    cam = ring_cameras(3, 150, 110, 5, device=dev)[2]
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    wc = torch.randn(3, 110, 150, device=dev, generator=g)
    res = []
    for fused in (False, True):
        pc = GaussianModel2D.from_raw(raw, 3, device=dev)
        pc.active_sh_degree = 2
        Pipe.fused_activations = fused
        r = render(cam, pc, Pipe(), bg)
        nl, dl = surfel_regularizers(r, 0.05, 100.0)
        ((r["render"] * wc).sum() + r["render_alpha"].sum() + nl + dl).backward()
        res.append((r, pc))
    (r0, p0), (r1, p1) = res
    assert set(r0) == {"render", "viewspace_points", "visibility_filter", "radii", "render_alpha", "render_normal",
                       "render_dist", "render_depth", "surf_normal"}
    assert r0["render_normal"].shape == (3, 110, 150) and r0["surf_normal"].shape == (3, 110, 150)
    assert torch.equal(r0["radii"], r1["radii"])
    assert (r0["render"] - r1["render"]).abs().max() < 2e-5
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        a, b = getattr(p0, name).grad, getattr(p1, name).grad
        assert a.shape == b.shape and torch.isfinite(a).all()
        assert (a - b).abs().max() <= 3e-3 * a.abs().max() + 1e-12, name
    # precomputed transform branch
    Pipe.fused_activations = False
    Pipe.compute_cov3D_python = True
    with torch.no_grad():
        r2 = render(cam, p0, Pipe(), bg)
    assert (r2["render"] - r0["render"]).abs().mean() < 1e-5
    Pipe.compute_cov3D_python = False
    # one optimisation step end to end
    Pipe.fused_activations = True
    p1.training_setup(OptimizationParams())
    p1.optimizer.zero_grad(set_to_none=True)
    r = render(cam, p1, Pipe(), bg)
    nl, dl = surfel_regularizers(r, 0.05, 100.0)
    loss = fused_l1_ssim_loss(r["render"], torch.rand(3, 110, 150, device=dev), 0.2) + nl + dl
    loss.backward()
    p1.add_densification_stats(r["viewspace_points"], r["visibility_filter"])
    p1.optimizer.step()
    assert torch.isfinite(p1._xyz).all() and torch.isfinite(p1._scaling).all() and p1._scaling.shape[1] == 2


@pytest.mark.parametrize("depth_ratio", [1.0, 0.0, 0.3])
@pytest.mark.parametrize("hw", [(37, 53), (64, 128), (5, 3)])
def test_surfel_maps_match_torch_restatement(dev, hw, depth_ratio):
    """scorp_gs2d_maps_forward/backward (the fused per-pixel tail of the 2DGS render()) against the op-for-op PyTorch
    restatement of gs2dgs/gaussian_renderer/__init__.py:131-160 + point_utils.py (oracle/surfel_maps_ref.py), forward
    maps and the gradient with respect to allmap, including empty pixels (alpha = 0 -> 0/0) and odd sizes.
    Tolerances: 1e-5 relative to each map's largest magnitude (same arithmetic, different association)."""
    from oracle.surfel_maps_ref import camera_rays, surfel_maps_ref
    from scorp_amd.rasterizer2d import surfel_maps
    from scorp_amd.synthetic import ring_cameras
    H, W = hw
    cam = ring_cameras(3, W, H, 11, device=dev)[1]
    rays_d, rays_o = camera_rays(cam.world_view_transform, cam.full_proj_transform, W, H)
    g = torch.Generator(device=dev).manual_seed(H * 1000 + W)
    alpha = torch.rand(1, H, W, device=dev, generator=g) * 0.98 + 0.01
    depth = 3.0 + torch.rand(1, H, W, device=dev, generator=g)
    allmap = torch.cat([depth * alpha, alpha, torch.randn(3, H, W, device=dev, generator=g) * alpha,
                        depth + 0.05 * torch.randn(1, H, W, device=dev, generator=g),
                        torch.rand(1, H, W, device=dev, generator=g) * 0.01], 0)
    empty = torch.rand(H, W, device=dev, generator=g) < 0.15        # pixels nothing was blended into
    allmap[:, empty] = 0.0
    ws = [torch.randn(c, H, W, device=dev, generator=g) for c in (1, 3, 1, 1, 3)]

    def run(fn, am):
        am = am.clone().requires_grad_(True)
        outs = fn(am)
        sum((o * w).sum() for o, w in zip(outs, ws)).backward()
        return [o.detach() for o in outs], am.grad

    outs_ref, g_ref = run(lambda am: surfel_maps_ref(am, cam.world_view_transform, rays_d, rays_o, depth_ratio), allmap)
    outs, g_hip = run(lambda am: surfel_maps(am, cam.world_view_transform, rays_d, rays_o, depth_ratio), allmap)
    for name, a, b in zip(("render_alpha", "render_normal", "render_dist", "surf_depth", "surf_normal"), outs, outs_ref):
        assert a.shape == b.shape, name
        assert (a - b).abs().max() <= 1e-5 * b.abs().max() + 1e-7, name
    assert torch.isfinite(g_hip).all()
    ok = torch.isfinite(g_ref)                      # PyTorch's chain leaves 0/0 = NaN at empty pixels; ours writes 0
    assert bool(ok[:, ~empty].all())
    for c in range(7):
        sel = ok[c]
        scale = g_ref[c][sel].abs().max()
        assert (g_hip[c][sel] - g_ref[c][sel]).abs().max() <= 2e-5 * scale + 1e-7, f"g_allmap[{c}]"
        # at the 0/0 pixels ours keeps only the direct term (the render_alpha weight on channel 1, nothing elsewhere)
        assert torch.equal(g_hip[c][~sel], ws[0][0][~sel] if c == 1 else torch.zeros_like(g_hip[c][~sel]))
    # None upstream gradients map to NULL pointers
    am = allmap.clone().requires_grad_(True)
    o = surfel_maps(am, cam.world_view_transform, rays_d, rays_o, depth_ratio)
    (o[4] * ws[4]).sum().backward()
    am2 = allmap.clone().requires_grad_(True)
    o2 = surfel_maps_ref(am2, cam.world_view_transform, rays_d, rays_o, depth_ratio)
    (o2[4] * ws[4]).sum().backward()
    sel = torch.isfinite(am2.grad)
    assert (am.grad[sel] - am2.grad[sel]).abs().max() <= 2e-5 * am2.grad[sel].abs().max() + 1e-7


@pytest.mark.parametrize("depth_ratio", [1.0, 0.0])
def test_fused_surfel_regularizers_match_the_torch_formulation(dev, depth_ratio):
    """fused_surfel_regularizers (one kernel each way, straight from allmap) == surfel_regularizers on the maps that
    render() returns (train_2dgs.py:142-150): loss values and the gradients that reach every Gaussian parameter."""
    from scorp_amd.renderer2d import GaussianModel2D, fused_surfel_regularizers, render, surfel_regularizers
    from scorp_amd.synthetic import make_gaussians, ring_cameras

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False
        fused_activations = True
    Pipe.depth_ratio = depth_ratio
    raw = make_gaussians(4000 + 3, 2, 17, log_scale_mean=math.log(0.05), scale_dims=2)
    cam = ring_cameras(3, 141, 97, 5, device=dev)[1]
    bg = torch.tensor([0.0, 0.0, 0.0], device=dev)
    res = []
    for fused in (False, True):
        pc = GaussianModel2D.from_raw(raw, 2, device=dev)
        r = render(cam, pc, Pipe(), bg)
        assert set(r) == {"render", "viewspace_points", "visibility_filter", "radii", "render_alpha", "render_normal",
                          "render_dist", "render_depth", "surf_normal"}
        nl, dl = (fused_surfel_regularizers if fused else surfel_regularizers)(r, 0.05, 100.0)
        (2.0 * nl + 0.5 * dl).backward()
        res.append((float(nl.detach()), float(dl.detach()), pc))
    (nl0, dl0, p0), (nl1, dl1, p1) = res
    assert abs(nl0 - nl1) <= 1e-6 * max(1.0, abs(nl0)) and abs(dl0 - dl1) <= 1e-6 * max(1.0, abs(dl0))
    for name in ("_xyz", "_opacity", "_scaling", "_rotation"):
        a, b = getattr(p0, name).grad, getattr(p1, name).grad
        assert torch.isfinite(b).all()
        assert (a - b).abs().max() <= 2e-4 * a.abs().max() + 1e-12, name


def test_image_only_render_2d_is_bit_identical(dev):
    """Without anything to differentiate the front-end takes scorp_gs2d_render_image (no state for a backward pass):
    colour, radii and the seven allmap channels must equal those of the differentiable render bit for bit."""
    name = next(iter(CASES))
    kw, _ = make_case2d(**CASES[name])
    out_g, _ = hip_render2d(kw, dev)                          # inputs require grad -> scorp_gs2d_render
    with torch.no_grad():
        out_n, _ = hip_render2d(kw, dev, requires_grad=False)  # -> scorp_gs2d_render_image
    for a, b in zip(out_g, out_n):
        assert torch.equal(a.detach(), b)


def test_full_size_properties_2d(dev):
    """BASELINE config #5 at its full size (S6: 1 M surfels, 1600x1200, SH3), where the oracle is too slow to run whole:
    size-independent properties of the surfel render - determinism, alpha in [0, 1], colour linear in the background
    with slope T_final = 1 - alpha, the geometry maps independent of the background, the distortion map non-negative,
    per-tile lists sorted by (depth, index), and a backward pass whose gradients are finite and deterministic up to the
    order of the float atomics."""
    from scorp_amd import _C
    from scorp_amd.synthetic import activate, make_gaussians, ring_cameras
    from tests.test_oracle2d_cpu import make_case2d  # noqa: F401  (same conventions)
    N, W, H = 1_000_000, 1600, 1200
    act = activate(make_gaussians(N, 3, 6, scale_dims=2))
    cam = ring_cameras(280, W, H, 6)[23]
    base = dict(means3D=act["means3D"], opacities=act["opacities"], shs=act["shs"], sh_degree=3, scales=act["scales"],
                rotations=act["rotations"], W=W, H=H, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
                view=cam.world_view_transform.numpy(), proj=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy())
    with torch.no_grad():
        (c0, r0, m0), _ = hip_render2d(dict(base, bg=np.zeros(3, np.float32)), dev, requires_grad=False)
        (c1, r1, m1), _ = hip_render2d(dict(base, bg=np.array([1.0, 0.5, 0.25], np.float32)), dev, requires_grad=False)
        (c2, r2, m2), _ = hip_render2d(dict(base, bg=np.zeros(3, np.float32)), dev, requires_grad=False)
    assert torch.equal(c0, c2) and torch.equal(m0, m2) and torch.equal(r0, r2)      # deterministic
    assert torch.equal(m0, m1) and torch.equal(r0, r1)                              # geometry maps do not see the background
    alpha = m0[1]
    assert float(alpha.min()) >= 0.0 and float(alpha.max()) <= 1.0 + 1e-5
    assert float(c0.min()) >= 0.0
    T_from_bg = c1[0] - c0[0]
    assert float((T_from_bg - (1.0 - alpha)).abs().max()) < 5e-5
    assert float(((c1[1] - c0[1]) - 0.5 * T_from_bg).abs().max()) < 5e-6
    assert float(m0[6].min()) >= -1e-6                                              # depth distortion is a sum of squares
    vis = int((r0 > 0).sum())
    assert 0.3 * N < vis <= N
    # backward: finite, and two runs agree to float-atomics noise
    grads = []
    for _ in range(2):
        (c, r, m), t = hip_render2d(dict(base, bg=np.zeros(3, np.float32)), dev)
        (c.mean() + 0.1 * m[0].mean() + 0.05 * m[6].mean()).backward()
        grads.append({k: v.grad.detach().clone() for k, v in t.items() if v is not None and v.grad is not None})
    for k, g in grads[0].items():
        assert bool(torch.isfinite(g).all()), k
        assert float((g - grads[1][k]).abs().max()) <= 1e-3 * float(g.abs().max()) + 1e-12, k
