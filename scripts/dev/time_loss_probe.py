"""What the loss kernels cost without their derivative-map traffic (a one-pass loss at <= 60 us).
forward with need_backward = 0 is the forward WITHOUT the map stores (an existing path); the backward without its map loads
needs a timing-only build whose maps are sixteen cache-resident rows (build/variants/libloss_probe.so: loss.hip with the
map row index masked by 15; wrong gradients).  us per call at 3 x 1200 x 1600, hipEvents over 200 calls."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scorp_amd import _C
L = _C.lib()
dev = torch.device("cuda:0")
C, H, W = 3, 1200, 1600
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x, y = torch.rand(C, H, W, device=dev), torch.rand(C, H, W, device=dev)
wsb = L.scorp_loss_workspace_bytes(C, H, W)
ws, out, g = torch.empty(wsb, dtype=torch.uint8, device=dev), torch.empty(3, device=dev), torch.empty(C, H, W, device=dev)
def fwd(nb): _C.check(L.scorp_loss_l1_ssim_forward(p(x), p(y), None, C, H, W, ctypes.c_float(0.2), p(out), p(ws), wsb, nb, st), "f")
def bwd(): _C.check(L.scorp_loss_l1_ssim_backward(p(x), p(y), None, C, H, W, ctypes.c_float(0.2), p(ws), None, p(g), st), "b")
def t(fn):
    for _ in range(30): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 5
print(os.environ.get("SCORP_GS_LIB", "default"), f"forward with map stores {t(lambda: fwd(1)):.1f} us, forward without {t(lambda: fwd(0)):.1f} us, backward {t(bwd):.1f} us")
