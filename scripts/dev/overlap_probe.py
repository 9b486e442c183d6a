"""How much of the loss kernels' time disappears when they run beside a blend kernel on a second stream (S3 sizes):
   render (no grad: preprocess .. blend forward) on stream A, photometric loss forward + backward on stream B."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.renderer import render
from scorp_amd.fused_loss import fused_l1_ssim_loss as photometric_loss
from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
dev = torch.device("cuda:0")


class Pipe:
    convert_SHs_python = False; compute_cov3D_python = False; debug = False; fused_activations = True


N, W, H, deg, seed, _ = SCENES["S3"]
m = GaussianModel.from_raw(make_gaussians(N, deg, seed), deg, device=dev); m.active_sh_degree = deg
cam = ring_cameras(280, W, H, seed, device=dev)[0]
bg = torch.zeros(3, device=dev)
img = torch.rand(3, H, W, device=dev, requires_grad=True)
gt = torch.rand(3, H, W, device=dev)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run_render():
    with torch.cuda.stream(sa), torch.no_grad():
        render(cam, m, Pipe(), bg)


def run_loss():
    with torch.cuda.stream(sb):
        l = photometric_loss(img, gt)
        l.backward()
        img.grad = None


def timed(fns, reps=30):
    for f in fns: f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for f in fns: f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for _ in range(3):
    a = timed([run_render]); b = timed([run_loss]); c = timed([run_render, run_loss])
    print(f"render alone {a:.0f} us, loss fwd+bwd alone {b:.0f} us, both on two streams {c:.0f} us (sum {a + b:.0f})")
