"""Lane occupancy of the blend forward's hits from the diagnostic build (scripts/build_ab.sh fwd_stats "-DSCORP_FWD_STATS=1"
gs3d_forward.hip):  SCORP_GS_LIB=build/variants/libfwd_stats.so python scripts/dev/stats_forward.py [S3 S2 ...]"""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scorp_amd import _C
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.renderer import render
from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
dev = torch.device("cuda:0")


class Pipe:
    convert_SHs_python = False; compute_cov3D_python = False; debug = False; fused_activations = True


L = _C.lib()
L.scorp_debug_fwd_stats.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 12)()
for name in (sys.argv[1:] or ["S3"]):
    N, W, H, deg, seed, _ = SCENES[name]
    m = GaussianModel.from_raw(make_gaussians(N, deg, seed), deg, device=dev); m.active_sh_degree = deg
    cams = ring_cameras(280, W, H, seed, device=dev)
    bg = torch.zeros(3, device=dev)
    render(cams[0], m, Pipe(), bg); torch.cuda.synchronize()
    assert L.scorp_debug_fwd_stats(buf, 1) == 0
    nv = 8
    for k in range(nv):
        render(cams[k * 35], m, Pipe(), bg)
    torch.cuda.synchronize()
    assert L.scorp_debug_fwd_stats(buf, 1) == 0
    hits, live, anyl, qsum, qmax, hsum, hmax, lmax, pairs, g8, g16, g64 = [v / nv for v in buf]
    print(json.dumps({"scene": name, "block_hits_per_view": round(hits), "live_lanes_per_hit": round(live / hits, 2),
                      "hits_with_a_live_lane": round(anyl / hits, 3),
                      "quadrant_hits_per_block_hit": round(qsum / hits, 3), "iterations_if_per_quadrant_lists": round(qmax / hits, 3),
                      "half_hits_per_block_hit": round(hsum / hits, 3), "iterations_if_per_half_lists": round(hmax / hits, 3),
                      "iterations_if_per_pixel_lists": round(lmax / hits, 3),
                      "quadrant_lists_balanced_ideal": round(qsum / hits / 4, 3),
                      "quadrant_lists_synchronised_every_8_16_64_kept_hits": [round(g8 / hits, 3), round(g16 / hits, 3), round(g64 / hits, 3)],
                      "iterations_if_adjacent_hits_with_disjoint_lanes_shared_one": round(1.0 - pairs / hits, 3)}))
