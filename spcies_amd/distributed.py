"""Multi-GPU: the batch of independent instances shards trivially.

One process per GPU (``torch.distributed``, backend ``nccl`` = RCCL over xGMI).  The only exchange
is a ONE-TIME broadcast of the factorised problem blob from rank 0; there is no collective inside
the iteration and results stay sharded.  ``shard_range`` / ``shard_inputs`` are backend-agnostic so
the partition logic is tested with ``gloo`` on CPU (tests/test_distributed_gloo.py).
"""
from __future__ import annotations

import numpy as np


def shard_range(total, world, rank):
    """Contiguous split of ``total`` instances: rank r gets [lo, hi); sizes differ by at most one."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_inputs(cfg, B_per_rank, rank):
    """Weak-scaling inputs: rank r draws its own B instances from a generator seeded with ``(cfg.seed, r)``, so a
    rank's shard does not depend on the world size and two ranks never share a stream (checked by the gloo test)."""
    from . import benchmarks
    rng = np.random.default_rng([cfg.seed, rank])
    sys = cfg.sys
    x0 = rng.uniform(-0.1, 0.1, size=(B_per_rank, sys.n))
    ur = 0.5 + 0.1 * rng.uniform(-1.0, 1.0, size=(B_per_rank, sys.m))
    xr = np.linalg.solve(sys.A - np.eye(sys.n), -(sys.B @ ur.T)).T.copy()
    return x0, xr, ur


def broadcast_blob(blob, device=None, src=0):
    """Broadcast the problem blob (bytes on ``src``, ``None`` elsewhere) to every rank.

    Two small collectives at handle-creation time: the length (int64) and the payload (uint8),
    on ``device`` when given (RCCL) or on CPU tensors (gloo).  Without an initialised process group
    ``blob`` is returned as it is; with one the collectives RUN, also at world size 1 (the same code
    path as at 8 ranks: ``bench.py --force-dist`` and ``tests/test_rccl_world1.py`` execute it on the
    one GPU a builder's box has).
    """
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return blob
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(blob) if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if dist.get_rank() == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return bytes(buf.cpu().numpy().tobytes())


def gather_results(u_local, total=None):
    """Optional: all-gather the sharded ``u`` (results normally stay sharded).  Tensors in, tensor out."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return u_local
    sizes = [torch.zeros(1, dtype=torch.int64, device=u_local.device) for _ in range(dist.get_world_size())]
    dist.all_gather(sizes, torch.tensor([u_local.shape[0]], dtype=torch.int64, device=u_local.device))
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((mx,) + tuple(u_local.shape[1:]), dtype=u_local.dtype, device=u_local.device)
    pad[: u_local.shape[0]] = u_local
    outs = [torch.empty_like(pad) for _ in sizes]
    dist.all_gather(outs, pad)
    return torch.cat([o[: int(s.item())] for o, s in zip(outs, sizes)], dim=0)
