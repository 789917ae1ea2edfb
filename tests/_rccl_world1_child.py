"""Child of tests/test_rccl_world1.py: ONE rank of the torch.distributed leg, run in a fresh process.

usage: python tests/_rccl_world1_child.py <backend: nccl|gloo> <out_dir> <total> [rank world port]

nccl (= RCCL, needs a GPU): init_process_group(world_size=1, device_id=cuda:0), the blob broadcast on DEVICE tensors, the all_reduce
behind bench.py's `rccl_ranks_seen`, barrier, the all_gather of `gather_results`, destroy_process_group; then the solver is built
from the BROADCAST bytes and solves its shard.  gloo: the same calls on CPU tensors, the blob round trip only (no solver: CPU run).
"""
import os
import sys
from datetime import timedelta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    backend, out_dir, total = sys.argv[1], sys.argv[2], int(sys.argv[3])
    rank, world, port = (int(a) for a in sys.argv[4:7]) if len(sys.argv) > 4 else (0, 1, 0)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before torch touches the GPU
    import numpy as np
    import torch
    import torch.distributed as dist

    from spcies_amd import benchmarks, blob as blobmod, distributed as spdist
    if port:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = None
    if backend == "nccl":
        if not torch.cuda.is_available():
            raise SystemExit("needs a HIP device")
        torch.cuda.set_device(rank % torch.cuda.device_count())
        dev = torch.device("cuda", rank % torch.cuda.device_count())
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timedelta(seconds=120))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
    cfg = benchmarks.config("C2")
    blob0 = blobmod.pack(benchmarks.ingredients(cfg)) if rank == 0 else None
    blob = spdist.broadcast_blob(blob0, dev)
    ones = torch.ones(1, dtype=torch.int32, device=dev if dev is not None else "cpu")
    dist.all_reduce(ones)
    dist.barrier()
    x0, xr, ur = benchmarks.sample_batch(cfg, total)
    lo, hi = spdist.shard_range(total, world, rank)
    report = {"ranks_seen": int(ones.item()), "blob_len": len(blob), "blob_equal": (blob == blob0) if rank == 0 else None,
              "backend": dist.get_backend(), "world": dist.get_world_size()}
    if backend == "nccl":
        from spcies_amd.solver import HipSolver
        solver = HipSolver(blob, device=dev.index)  # from the broadcast bytes only
        tx0, txr, tur = (torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(dev) for a in (x0, xr, ur))
        tu = torch.empty((hi - lo, cfg.sys.m), dtype=torch.float64, device=dev)
        tk = torch.empty(hi - lo, dtype=torch.int32, device=dev)
        te = torch.empty(hi - lo, dtype=torch.int32, device=dev)
        solver.solve_device(tx0, txr, tur, tu, tk, te, stream=torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        ug = spdist.gather_results(tu)       # all_gather on device tensors (ragged shards padded)
        kg = spdist.gather_results(tk)
        report["variant"] = solver.variant
        solver.close()
        if rank == 0:
            np.save(os.path.join(out_dir, "u.npy"), ug.cpu().numpy())
            np.save(os.path.join(out_dir, "k.npy"), kg.cpu().numpy())
    else:
        t = spdist.gather_results(torch.from_numpy(x0[lo:hi].copy()))
        if rank == 0:
            np.save(os.path.join(out_dir, "x0.npy"), t.numpy())
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        import json
        with open(os.path.join(out_dir, "report.json"), "w") as f:
            json.dump(report, f)
    print("rank", rank, "ok", report, flush=True)


if __name__ == "__main__":
    main()
