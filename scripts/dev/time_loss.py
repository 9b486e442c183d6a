"""us per call of the fused loss forward / backward at 1600x1200 for C = 1, 2, 3 channels (hipEvents, 100 calls): how the
loss kernels scale with the number of waves (C = 1 is about one wave per SIMD for the strip kernels)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scorp_amd import _C
L = _C.lib()
dev = torch.device("cuda:0")
H, W = 1200, 1600
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for C in (1, 2, 3):
    x, y = torch.rand(C, H, W, device=dev), torch.rand(C, H, W, device=dev)
    wsb = L.scorp_loss_workspace_bytes(C, H, W)
    ws, out, g = torch.empty(wsb, dtype=torch.uint8, device=dev), torch.empty(3, device=dev), torch.empty(C, H, W, device=dev)
    def fwd(): _C.check(L.scorp_loss_l1_ssim_forward(p(x), p(y), None, C, H, W, ctypes.c_float(0.2), p(out), p(ws), wsb, 1, st), "f")
    def bwd(): _C.check(L.scorp_loss_l1_ssim_backward(p(x), p(y), None, C, H, W, ctypes.c_float(0.2), p(ws), None, p(g), st), "b")
    res = []
    for fn in (fwd, bwd):
        for _ in range(20): fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(100): fn()
        b.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 10)
    print(f"C={C}: forward(+finalize) {res[0]:6.1f} us   backward {res[1]:6.1f} us   loss {float(out[0]):.6f}")
