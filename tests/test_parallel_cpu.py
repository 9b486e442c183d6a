"""world_size-2 gloo tests (CPU) of the N>1 path: round-robin sharding, scene broadcast, result gather, arg-max."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scorp_amd import parallel as P
    try:
        # shard: disjoint cover, view i -> rank i mod world
        mine = P.shard_indices(11)
        assert mine == list(range(rank, 11, world))
        # scene broadcast: only rank 0 holds the data
        t = {"xyz": torch.arange(12.0).reshape(4, 3) if rank == 0 else torch.zeros(4, 3),
             "opacity": torch.ones(4, 1) * 3 if rank == 0 else torch.zeros(4, 1)}
        P.broadcast_tensors(t, src=0)
        assert torch.equal(t["xyz"], torch.arange(12.0).reshape(4, 3)) and float(t["opacity"].sum()) == 12.0
        # sweep: fitness peaks at unit 7; uneven shard sizes (11 units over 2 ranks)
        ids, scores, best = P.sweep(11, lambda i: torch.tensor([-(i - 7.0) ** 2, float(i)]))
        assert ids.tolist() == list(range(11)) and best == 7
        assert scores[:, 1].tolist() == [float(i) for i in range(11)]
        # a rank with nothing to do (more ranks than units)
        ids, scores, best = P.sweep(1, lambda i: torch.tensor([1.0]))
        assert ids.tolist() == [0] and best == 0
        # data-parallel gradient averaging in flat buckets; a rank with a missing gradient contributes zeros
        pa = torch.nn.Parameter(torch.zeros(5, 3))
        pb = torch.nn.Parameter(torch.zeros(7))
        pa.grad = torch.full((5, 3), float(rank + 1))
        pb.grad = torch.arange(7.0) * (rank + 1) if rank == 0 else None
        P.average_gradients([pa, pb], bucket_bytes=32)
        assert torch.allclose(pa.grad, torch.full((5, 3), 1.5))
        assert torch.allclose(pb.grad, torch.arange(7.0) * 0.5)
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_broadcast_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_single_process_paths():
    from scorp_amd import parallel as P
    assert P.world() == (0, 1)
    assert P.shard_indices(5) == [0, 1, 2, 3, 4]
    assert P.shard_indices(10, rank=3, world_size=4) == [3, 7]
    ids, scores, best = P.sweep(4, lambda i: torch.tensor([float(-abs(i - 2))]))
    assert best == 2 and ids.tolist() == [0, 1, 2, 3]


def test_sh_rotation_blocks_are_consistent():
    """c' = D c rotates the function: f'(d) = f(R^-1 d); D of a product is the product; D is orthogonal."""
    from scorp_amd.sh import eval_sh
    from scorp_amd.transforms import matrix_to_quat, quat_multiply, sh_rotation_blocks
    rots = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotations_128.npz"))["rotations"]
    assert rots.shape == (128, 3, 3)
    R1, R2 = torch.tensor(rots[5]), torch.tensor(rots[77])
    B1, B2, B12 = sh_rotation_blocks(R1), sh_rotation_blocks(R2), sh_rotation_blocks(R1 @ R2)
    g = torch.Generator().manual_seed(1)
    c = torch.randn(1, 3, 16, generator=g, dtype=torch.float64)
    d = torch.randn(50, 3, generator=g, dtype=torch.float64)
    d = d / d.norm(dim=1, keepdim=True)
    c_rot = c.clone()
    for l, D in enumerate(B1, start=1):
        sl = slice(l * l, (l + 1) ** 2)
        assert torch.allclose(D @ D.T, torch.eye(2 * l + 1), atol=1e-5)
        assert torch.allclose(B12[l - 1], B1[l - 1] @ B2[l - 1], atol=1e-5)
        c_rot[..., sl] = torch.einsum("ij,ncj->nci", D.double(), c[..., sl])
    f_rot = eval_sh(3, c_rot.expand(50, 3, 16), d)
    f_ref = eval_sh(3, c.expand(50, 3, 16), d @ R1.double())       # f(R1^-1 d)
    assert torch.allclose(f_rot, f_ref, atol=1e-5)
    q = matrix_to_quat(R1)
    assert abs(float(q.norm()) - 1) < 1e-6
    qq = quat_multiply(q, torch.tensor([1.0, 0, 0, 0]))
    assert torch.allclose(qq, q)
