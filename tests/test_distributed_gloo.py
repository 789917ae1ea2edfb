"""CPU, world_size 2, gloo: the N>1 plumbing - blob broadcast, shard partition, result gather.
(The solve itself needs a GPU; here each rank runs the ORACLE on its shard as the stand-in compute so
the partition + gather logic is checked end to end against a single-process run.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    from spcies_amd import benchmarks, blob as blobmod, distributed as spdist
    cfg = benchmarks.config("C1")
    blob = blobmod.pack(benchmarks.ingredients(cfg)) if rank == 0 else None
    blob = spdist.broadcast_blob(blob)
    v = blobmod.unpack(blob)  # every rank rebuilds the controller from the broadcast bytes only
    x0, xr, ur = benchmarks.sample_batch(cfg, total)
    lo, hi = spdist.shard_range(total, world, rank)
    u, k, e, *_ = oracle.admm_banded_batch(v, x0[lo:hi], xr[lo:hi], ur[lo:hi], want_sol=False)
    ug = spdist.gather_results(torch.from_numpy(u))
    if rank == 0:
        np.save(os.path.join(out_dir, "u.npy"), ug.numpy())
        np.save(os.path.join(out_dir, "blob_len.npy"), np.array([len(blob)]))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from spcies_amd.distributed import shard_range
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, world, r) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_weak_scaling_shards_are_world_size_independent():
    from spcies_amd import benchmarks
    from spcies_amd.distributed import shard_inputs
    cfg = benchmarks.config("C2")
    a = shard_inputs(cfg, 32, 1)
    b = shard_inputs(cfg, 32, 1)
    c = shard_inputs(cfg, 32, 0)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert not np.array_equal(a[0], c[0])


def test_two_rank_gloo_matches_single_process(tmp_path):
    total = 11  # ragged: 6 + 5
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from spcies_amd import benchmarks
    cfg = benchmarks.config("C1")
    v = benchmarks.ingredients(cfg)
    x0, xr, ur = benchmarks.sample_batch(cfg, total)
    u, *_ = oracle.admm_banded_batch(v, x0, xr, ur, want_sol=False)
    assert np.array_equal(np.load(tmp_path / "u.npy"), u)
