"""`FusedAdam` — torch.optim.Adam semantics (per-group lr, betas, eps, no weight decay / amsgrad), all parameter
groups updated by ONE HIP launch (scorp_adam_step).  State layout (`step`, `exp_avg`, `exp_avg_sq` per parameter) is
torch's, so `GaussianModel`'s optimizer surgery (densify / prune / reset, gaussian_model.py:412-601) and
`state_dict()` checkpoints (`capture`/`restore`, :92-124) work unchanged."""
import ctypes

import torch

from . import _C


class FusedAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False, fused=False)
        # Optional 1-element int32 device tensor: the NEXT step() updates nothing if it is non-zero when the kernel runs
        # (scorp_adam_step_guarded).  train() sets it to the overflow word of a view rendered with a reserved pair buffer;
        # step() consumes it.  The host-side step counter advances regardless (the host does not know yet); whoever reads
        # the overflow words later (train._drain_reservation) takes the skipped steps back with rollback_steps().
        self.skip_flag = None
        self._stepped = []

    def rollback_steps(self, n=1):
        """`n` of the steps taken so far were skipped on the device (guarded by a non-zero overflow word): take them out of
        every parameter's bias-correction counter, so that the next update is scaled as torch.optim.Adam would scale it."""
        if n <= 0:
            return
        for st in self._stepped:       # (the parameters the last step() advanced: frozen leaves keep their counters)
            st["step"] -= min(float(n), float(st["step"]))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = _C.lib()
        by_cfg, stepped = {}, []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam needs GPU parameters (scorp_amd has no CPU path)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                stepped.append(st)
                key = (group["betas"], group["eps"], int(st["step"]))
                by_cfg.setdefault(key, []).append((p, p.grad.contiguous(), st, float(group["lr"])))
        self._stepped = stepped
        stream = ctypes.c_void_p(_C.current_stream_ptr())
        skip, self.skip_flag = self.skip_flag, None
        skip_ptr = None if skip is None else ctypes.c_void_p(skip.data_ptr())
        for (betas, eps, step), items in by_cfg.items():
            for i in range(0, len(items), 8):
                chunk = items[i:i + 8]
                arr = (_C.ScorpAdamTensor * len(chunk))()
                for k, (p, g, st, lr) in enumerate(chunk):
                    assert p.is_contiguous() and p.dtype == torch.float32
                    arr[k].param, arr[k].grad = p.data_ptr(), g.data_ptr()
                    arr[k].exp_avg, arr[k].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    arr[k].numel, arr[k].lr = p.numel(), lr
                _C.check(L.scorp_adam_step_guarded(arr, len(chunk), betas[0], betas[1], eps, step, skip_ptr, stream),
                         "scorp_adam_step_guarded")
        return loss
