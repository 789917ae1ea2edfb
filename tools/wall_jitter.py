"""Where does wall time go beyond kernel time?  (diagnostic; run on the GPU box)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from spcies_amd import benchmarks, blob as blobmod, distributed as spdist
from spcies_amd.solver import HipSolver

cfg = benchmarks.config("C2"); dev = torch.device("cuda", 0)
solver = HipSolver(blobmod.pack(benchmarks.ingredients(cfg)), device=0); solver.set_variant("mfma4")
B = 65536
x0, xr, ur = spdist.shard_inputs(cfg, B, 0)
tx0, txr, tur = (torch.from_numpy(a).to(dev) for a in (x0, xr, ur))
tu = torch.empty((B, cfg.sys.m), dtype=torch.float64, device=dev)
tk = torch.empty(B, dtype=torch.int32, device=dev); te = torch.empty(B, dtype=torch.int32, device=dev)
solver.reserve(B)
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(4): solver.solve_device(tx0, txr, tur, tu, tk, te, stream=st)
torch.cuda.synchronize()
for K in (1, 5, 10, 20, 50, 20, 10):
    for rep in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); e0.record()
        ts = []
        for _ in range(K):
            solver.solve_device(tx0, txr, tur, tu, tk, te, stream=st); ts.append(time.perf_counter())
        e1.record(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"K={K:3d} wall {1e3*(t2-t0):8.2f} ms  launch-loop {1e3*(t1-t0):7.2f}  events {e0.elapsed_time(e1):8.2f}  per-step wall {1e3*(t2-t0)/K:6.2f}")
