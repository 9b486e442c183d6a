"""One-process-per-GPU sharding helpers.  The path shards over independent units — views, objects,
pose hypotheses — so the only collectives are one broadcast of the Gaussians at start and one gather of small results
at the end; nothing on the per-view critical path.  Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in CPU tests.
"""
import os

import torch
import torch.distributed as dist

# A process group of ONE rank short-circuits every helper below (nothing to exchange).  With this switch (or
# SCORP_SINGLE_RANK_COLLECTIVES=1 in the environment) an initialised group of one rank issues its collectives anyway:
# trivial exchanges, but the real code path - init_process_group(device_id=...), broadcast, all_gather_into_tensor,
# reduce_scatter_tensor on device tensors over RCCL - which is how tests/test_rccl_gpu.py executes the "nccl" branches
# on the one GPU a test box has.
SINGLE_RANK_COLLECTIVES = os.environ.get("SCORP_SINGLE_RANK_COLLECTIVES", "0") == "1"


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def collective():
    """True where the helpers have to talk: more than one rank, or one rank with SINGLE_RANK_COLLECTIVES."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or SINGLE_RANK_COLLECTIVES


def all_ok(local_ok, device="cpu"):
    """Collective agreement that every rank got through its local part: all-reduce(MIN) of one int.  Call it between the
    local part of a unit of work and its exchange step, so that a rank that failed does not leave the others waiting in
    a collective it never enters (it raises on every rank instead)."""
    if not collective():
        return bool(local_ok)
    if dist.get_backend() == "nccl" and torch.device(device).type != "cuda":
        device = torch.device("cuda", torch.cuda.current_device())
    t = torch.tensor([1 if local_ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def shard_indices(n, rank=None, world_size=None):
    """Unit i -> rank i mod world (round-robin keeps neighbouring, similarly expensive units on different ranks)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    return list(range(rank, n, world_size))


def broadcast_tensors(tensors, src=0):
    """In-place broadcast of a dict of float32 tensors (the scene: N x 236 B = 236 MB at 1 M Gaussians, SH3) as ONE flat
    buffer: one collective instead of one per tensor (xGMI is point-to-point - few, large messages).  Every
    rank passes tensors of the same shapes; the source's values arrive in place."""
    r, w = world()
    if collective() and tensors:
        keys = sorted(tensors)
        flat = torch.cat([tensors[k].reshape(-1) for k in keys]) if r == src else \
            torch.empty(sum(tensors[k].numel() for k in keys), dtype=tensors[keys[0]].dtype, device=tensors[keys[0]].device)
        dist.broadcast(flat, src=src)
        if r != src:
            off = 0
            for k in keys:
                n = tensors[k].numel()
                tensors[k].copy_(flat[off:off + n].view_as(tensors[k]))
                off += n
    return tensors


def gather_results(ids, values, n_total=None, failed=None):
    """All ranks receive every (id, value-row) pair, ordered by id.  `ids`: int64[m]; `values`: float32[m, c].

    `failed` (with `n_total`; a bool, this rank's local part raised): one extra row per rank carries the flag through the SAME
    all-gather - every rank sends it, also one whose share of the units is empty - and the call returns a third value, a
    0-d device tensor that is non-zero if ANY rank failed (read it together with the results: no extra collective, no extra
    host synchronisation).  A failed rank's value rows are whatever it passed (zeros): a score that is legitimately NaN is
    not mistaken for a failure, and a failure cannot hide behind an empty shard.

    With `n_total` (the units were dealt by `shard_indices(n_total)`, as every caller here does) the exchange is ONE
    fixed-size all-gather and no host synchronisation: every rank's share is at most ceil(n_total / world) rows, so the
    padded size is known without asking, ids ride in the same buffer as a float column (exact below 2^24) and the padding
    is removed by construction (rank r holds ids r, r + world, ...).  Without it (ragged, unknown shares) the counts are
    exchanged first."""
    _, w = world()
    dev = values.device
    ids = torch.as_tensor(ids, dtype=torch.int64, device=dev)
    if collective() and n_total is not None and n_total < (1 << 24):
        m, c = -(-int(n_total) // w), values.shape[1]
        mr = m + (1 if failed is not None else 0)          # (+ the status row)
        buf = torch.full((mr, c + 1), -1.0, dtype=torch.float32, device=dev)
        buf[: ids.numel(), 0] = ids.to(torch.float32)
        buf[: ids.numel(), 1:] = values.to(torch.float32)
        if failed is not None:
            buf[m, 0] = 1.0 if failed else 0.0
        out = torch.empty((w * mr, c + 1), dtype=torch.float32, device=dev)
        dist.all_gather_into_tensor(out, buf)
        # rank r's row k is unit r + k * world, so unit u sits in row (u mod world) * m + u div world: the valid rows and
        # their order follow from n_total alone - index arithmetic, no boolean mask (out[mask] is a host synchronisation)
        unit = torch.arange(int(n_total), device=dev)
        rows = out[(unit % w) * mr + unit // w]
        if failed is not None:
            return rows[:, 0].to(torch.int64), rows[:, 1:].to(values.dtype), out.view(w, mr, c + 1)[:, m, 0].amax()
        return rows[:, 0].to(torch.int64), rows[:, 1:].to(values.dtype)
    if collective():
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(w)]
        dist.all_gather(counts, torch.tensor([ids.numel()], dtype=torch.int64, device=dev))
        m = int(max(c.item() for c in counts))
        pad_i = torch.full((m,), -1, dtype=torch.int64, device=dev)
        pad_v = torch.zeros((m, values.shape[1]), dtype=values.dtype, device=dev)
        pad_i[: ids.numel()] = ids
        pad_v[: ids.numel()] = values
        gi = [torch.empty_like(pad_i) for _ in range(w)]
        gv = [torch.empty_like(pad_v) for _ in range(w)]
        dist.all_gather(gi, pad_i)
        dist.all_gather(gv, pad_v)
        ids, values = torch.cat(gi), torch.cat(gv)
        keep = ids >= 0
        ids, values = ids[keep], values[keep]
    order = torch.argsort(ids)
    if failed is not None:      # (no group, or the ragged exchange: the flag is this rank's own / reduced apart)
        f = torch.tensor(1.0 if failed else 0.0, device=dev)
        if collective():
            dist.all_reduce(f, op=dist.ReduceOp.MAX)
        return ids[order], values[order], f
    return ids[order], values[order]


def sweep(n_units, score_fn, device="cpu"):
    """Evaluate score_fn(i) -> float tensor[c] for the units of this rank, gather, and return (ids, scores, argmax id).
    This is the shape of the 128-rotation alignment sweep: arg-max of an independent per-hypothesis fitness
    (align_3dgs_clpe_9dof.py:95-111)."""
    mine = shard_indices(n_units)
    vals = [score_fn(i).reshape(-1).float() for i in mine]
    c = vals[0].numel() if vals else 1
    v = torch.stack(vals) if vals else torch.zeros((0, c), dtype=torch.float32, device=device)
    ids, scores = gather_results(mine, v.to(device), n_total=n_units)
    best = int(ids[torch.argmax(scores[:, 0])]) if ids.numel() else -1
    return ids, scores, best


class GradArena:
    """The gradients of a model's leaves as views of ONE persistent flat buffer, widest leaf first (round 6).

    `average_gradients` packs the leaves' .grad into 64 MiB buckets (torch.cat: a pass over 248 MB at 1 M Gaussians),
    all-reduces each bucket while the host waits, and copies the result back (a third pass).  With the arena the one-call view
    (train_view(grad_out=arena.views)) writes its gradients where the collective reads them and the averaged values are read by
    the optimizer where the collective left them: no packing, no copy-back, no allocation per iteration.  The buffer is
    exchanged as TWO collectives issued back to back without a host wait in between - `_features_rest` (180 of the 248
    bytes per Gaussian) and everything else - so that the second one's launch and the first one's wire time overlap; the
    wait happens once, behind both.  On RCCL each part is a reduce-scatter + all-gather pair (every rank reduces 1 / G of
    the floats, all seven xGMI links of a GPU carry payload both ways); on gloo (the CPU tests) an all-reduce.
    `views[k]` is None for a leaf that does not require a gradient."""

    def __init__(self, params):
        self.params = list(params)
        live = [p for p in self.params if p.requires_grad]
        dev, dt = live[0].device, live[0].dtype
        order = sorted(range(len(live)), key=lambda i: -live[i].numel())      # the widest leaf (features_rest) first
        _, w = world()
        pad = lambda n: -(-n // (4 * max(w, 1))) * (4 * max(w, 1))            # every part divisible by the world size, 16-byte pieces
        sizes = [pad(live[i].numel()) for i in order]
        self.flat = torch.zeros(sum(sizes), dtype=dt, device=dev)
        self.key = tuple((p.shape, p.requires_grad) for p in self.params)
        offs, off = {}, 0
        for i, sz in zip(order, sizes):
            offs[id(live[i])] = off
            off += sz
        self.views = [self.flat[offs[id(p)]:offs[id(p)] + p.numel()].view_as(p) if p.requires_grad else None for p in self.params]
        head = sizes[0]
        self.parts = [self.flat[:head], self.flat[head:]] if len(sizes) > 1 else [self.flat]

    def matches(self, params):
        return tuple((p.shape, p.requires_grad) for p in params) == self.key and all(
            v is None or v.device == p.device for v, p in zip(self.views, params))

    def attach(self):
        """Point every leaf's .grad at its view (for gradients that were produced elsewhere: they are copied in)."""
        for p, v in zip(self.params, self.views):
            if v is None:
                continue
            if p.grad is None:
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v

    def average(self):
        """In place: every float of the arena becomes its mean over the ranks."""
        _, w = world()
        if not collective():
            return
        works = []
        nccl = dist.get_backend() == "nccl"
        for part in self.parts:
            if part.numel() == 0:
                continue
            if nccl:
                shard = torch.empty(part.numel() // w, dtype=part.dtype, device=part.device)
                dist.reduce_scatter_tensor(shard, part, op=dist.ReduceOp.SUM)
                shard /= w
                works.append(dist.all_gather_into_tensor(part, shard, async_op=True))
            else:
                works.append((dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True), part))
        for wk in works:
            if isinstance(wk, tuple):
                wk[0].wait()
                wk[1].div_(w)
            else:
                wk.wait()


def average_gradients(params, bucket_bytes=64 << 20):
    """Data-parallel training of ONE scene: every rank renders a different view, then the
    per-Gaussian gradients are averaged before the optimizer step.  Gradients are packed into a few large flat
    buckets (default 64 MiB; at 1 M Gaussians, SH3 the 248 MB of gradients make 4 collectives) — xGMI is
    point-to-point, so a ring all-reduce is per-link bound (~153 GB/s) and wants few, large messages.
    Parameters whose .grad is None on this rank (nothing visible) contribute zeros."""
    _, w = world()
    if not collective():
        return
    params = [p for p in params if p.requires_grad]
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= w
        off = 0
        for p in bucket:
            n = p.numel()
            g = flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n
        bucket, size = [], 0

    for p in params:
        bucket.append(p)
        size += p.numel() * p.element_size()
        if size >= bucket_bytes:
            flush()
    flush()


def average_gradients_sparse(params, visible, bucket_bytes=64 << 20, dense_above=0.6):
    """The same average as `average_gradients`, moving only the rows that can be non-zero: a Gaussian's gradient row is
    zero on a rank where it was not rendered (radii == 0), so rows outside the UNION of the ranks' visibility masks are
    zero everywhere and need no traffic (train_3dgs.py:56-193 with one view per rank).

      1. union of the masks: one all-reduce(MAX) of N bytes (1 MB at 1 M Gaussians);
      2. the union's rows of every per-Gaussian gradient, packed row-major into one flat buffer, are summed with
         reduce-scatter + all-gather over RCCL (every rank reduces 1/G of the rows: all seven xGMI links of a GPU carry
         payload both ways) - or one all-reduce where the backend has no reduce-scatter (gloo, the CPU tests);
      3. the averaged rows are scattered back; every other row of .grad is zero.

    `params`: per-Gaussian parameter tensors ([N, ...]); `visible`: bool[N] of this rank's view.  Returns the number
    of rows in the union.  In a large scene seen from inside a view shows a fraction of the Gaussians and the 248 MB dense
    all-reduce shrinks by that fraction; where the union covers more than `dense_above` of the Gaussians the rows are not
    packed at all and the dense bucketed average runs (decided from the reduced mask, so every rank takes the same branch)."""
    r, w = world()
    if not collective():
        return int(visible.sum())
    params = [p for p in params if p.requires_grad]
    dev = params[0].device
    union = visible.to(torch.uint8).clone()
    dist.all_reduce(union, op=dist.ReduceOp.MAX)
    n = int(union.sum())                       # (the same number on every rank: the branch below is taken together)
    if n == 0:
        for p in params:
            p.grad = torch.zeros_like(p)
        return 0
    if n > dense_above * union.numel():
        # The union is (nearly) everything - an object seen whole, the synthetic S3: the packed form would move the dense
        # payload AND pay a row gather before and a row scatter after the exchange (2 x 248 MB of indexed copies at 1 M
        # Gaussians; the round-3 rehearsal measured 46 it/s sparse against 52 dense for exactly this reason).
        average_gradients(params, bucket_bytes)
        return n
    idx = torch.nonzero(union, as_tuple=False).squeeze(-1)
    rows = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(p.shape[0], -1)[idx] for p in params]
    widths = [x.shape[1] for x in rows]
    flat = torch.cat(rows, dim=1).reshape(-1)                     # [n, sum(widths)] row-major
    pad = (-flat.numel()) % w
    if pad:
        flat = torch.cat([flat, torch.zeros(pad, dtype=flat.dtype, device=dev)])
    backend = dist.get_backend()
    if backend == "nccl":
        part = torch.empty(flat.numel() // w, dtype=flat.dtype, device=dev)
        dist.reduce_scatter_tensor(part, flat, op=dist.ReduceOp.SUM)
        part /= w
        dist.all_gather_into_tensor(flat, part)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        flat /= w
    got = flat[: n * sum(widths)].view(n, sum(widths))
    off = 0
    for p, wd in zip(params, widths):
        g = torch.zeros((p.shape[0], wd), dtype=p.dtype, device=dev)
        g[idx] = got[:, off:off + wd]
        p.grad = g.view_as(p)
        off += wd
    return n


def gather_rows(rows_by_unit, n_units, widths, c=None, device=None):
    """Every rank ends up with the per-unit row blocks of ALL units in ONE fixed-size all-gather.  `rows_by_unit`:
    {unit id: float32 tensor [widths[unit], c]} for the units of this rank (dealt by `shard_indices(n_units)`);
    `widths`: the row count of every unit, known to every rank (object sizes); `c` / `device`: row width and device,
    needed from a rank that holds no unit (more ranks than objects).  Returns a list of n_units tensors."""
    r, w = world()
    if c is None:
        c = next(iter(rows_by_unit.values())).shape[1]
    dev = device if device is not None else next(iter(rows_by_unit.values())).device
    if not collective():
        return [rows_by_unit[j] for j in range(n_units)]
    shares = [sum(widths[j] for j in range(k, n_units, w)) for k in range(w)]    # rows every rank contributes
    m = max(shares) if shares else 0
    buf = torch.zeros((m, c), dtype=torch.float32, device=dev)
    off = 0
    for j in range(r, n_units, w):
        buf[off:off + widths[j]] = rows_by_unit[j].to(torch.float32)
        off += widths[j]
    out = torch.empty((w * m, c), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(out, buf)
    res = [None] * n_units
    for k in range(w):
        off = k * m
        for j in range(k, n_units, w):
            res[j] = out[off:off + widths[j]]
            off += widths[j]
    return res
