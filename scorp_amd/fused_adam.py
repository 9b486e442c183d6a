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
        # Device word counting the steps the DEVICE skipped (scorp_adam_step_guarded_ex / the fused step of
        # scorp_gs3d_train_view increment it when their guard word is set).  It counts THIS optimizer's steps only, and in a
        # data-parallel run the guard is the all-reduced overflow word, so every replica reads the same count at the same
        # drain: take_skipped() is what rollback_steps() is fed from (train._drain_reservation).
        self._skipped = None

    def skipped_counter(self, device):
        if self._skipped is None or self._skipped.device != device:
            self._skipped = torch.zeros(1, dtype=torch.int32, device=device)
        return self._skipped

    def take_skipped(self):
        """Number of steps the device skipped since the last call (one device-to-host read; call it where the loop
        synchronises anyway) - and the counter starts again from zero."""
        if self._skipped is None:
            return 0
        n = int(self._skipped.item())
        if n:
            self._skipped.zero_()
        return n

    def fused_view_pack(self, leaves, stats=None):
        """The optimizer step of THIS iteration as a `_C.ScorpFusedAdam` for scorp_gs3d_train_view (train_view.train_view(...,
        optimizer=...)): leaves in the order xyz, features_dc, features_rest, opacity, scaling, rotation.  Advances the step
        counters as step() does.  Returns (struct, keep-alive list), or None if this step cannot be fused (a leaf has a
        pending gradient to accumulate into, step counters that disagree, a leaf this optimizer does not hold)."""
        by_param = {id(p): g for g in self.param_groups for p in g["params"]}
        groups, states = [], []
        for p in leaves:
            if not p.requires_grad:
                groups.append(None); states.append(None)
                continue
            g = by_param.get(id(p))
            if g is None or p.grad is not None or not p.is_cuda or not p.is_contiguous() or p.dtype != torch.float32:
                return None
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            groups.append(g); states.append(st)
        live = [(g, st) for g, st in zip(groups, states) if g is not None]
        if not live or len({(g["betas"], g["eps"], int(st["step"])) for g, st in live}) != 1:
            return None
        fa = _C.ScorpFusedAdam()
        for k, (g, st) in enumerate(zip(groups, states)):
            if g is None:
                continue
            fa.exp_avg[k], fa.exp_avg_sq[k], fa.lr[k] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), float(g["lr"])
        g0, st0 = live[0]
        fa.beta1, fa.beta2, fa.eps = g0["betas"][0], g0["betas"][1], g0["eps"]
        for _, st in live:
            st["step"] += 1
        self._stepped = [st for _, st in live]
        fa.step = int(st0["step"])
        fa.skipped_counter = self.skipped_counter(leaves[0].device).data_ptr()
        if stats is not None:
            fa.max_radii2D, fa.xyz_gradient_accum, fa.denom = (t.data_ptr() for t in stats)
        return fa, [st for _, st in live]

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
        # (counted once per step, by the step's first launch)
        count_ptr = None if skip is None or not stepped else ctypes.c_void_p(self.skipped_counter(skip.device).data_ptr())
        for (betas, eps, step), items in by_cfg.items():
            for i in range(0, len(items), 8):
                chunk = items[i:i + 8]
                arr = (_C.ScorpAdamTensor * len(chunk))()
                for k, (p, g, st, lr) in enumerate(chunk):
                    assert p.is_contiguous() and p.dtype == torch.float32
                    arr[k].param, arr[k].grad = p.data_ptr(), g.data_ptr()
                    arr[k].exp_avg, arr[k].exp_avg_sq = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                    arr[k].numel, arr[k].lr = p.numel(), lr
                _C.check(L.scorp_adam_step_guarded_ex(arr, len(chunk), betas[0], betas[1], eps, step, skip_ptr, count_ptr, stream),
                         "scorp_adam_step_guarded")
                count_ptr = None
        return loss
