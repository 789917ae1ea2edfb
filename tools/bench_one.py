#!/usr/bin/env python3
"""Kernel time of one configuration / variant on the GPU box:  python tools/bench_one.py C5_HMPC_SADMM fused [B] [reps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver

name, variant = sys.argv[1], sys.argv[2]
cfg = benchmarks.config(name)
B = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
v = benchmarks.ingredients(cfg)
s = HipSolver(v)
if variant != "auto":
    s.set_variant(variant)
dev = torch.device("cuda", 0)
x0, xr, ur = benchmarks.sample_batch(cfg, B)
t = lambda a: torch.from_numpy(a).to(dev)
tx0, txr, tur = t(x0), t(xr), t(ur)
tu = torch.empty((B, cfg.sys.m), dtype=torch.float64, device=dev)
tk = torch.empty(B, dtype=torch.int32, device=dev)
te = torch.empty(B, dtype=torch.int32, device=dev)
extra = None
if cfg.formulation == "ellipMPC" and getattr(cfg, "submethod", "") == "soc":
    extra = torch.full((1,), float(cfg.param.r), dtype=torch.float64, device=dev)
s.reserve(B)
st = torch.cuda.current_stream(dev).cuda_stream
run = lambda: s.solve_device_ex(tx0, txr, tur, tu, tk, te, extra=extra, stream=st)
run()
torch.cuda.synchronize()
times = []
for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
ms = float(np.median(times))
print(json.dumps(dict(config=name, variant=s.variant, B=B, kernel_ms=round(ms, 3), solves_per_s=round(B / ms * 1e3),
                      all_ms=[round(x, 2) for x in times], k_unique=np.unique(tk.cpu().numpy()).tolist()[:4])), flush=True)
s.close()
