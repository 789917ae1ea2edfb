"""Kernel experiments on the headline MFMA4 kernel without rebuilding the library: each entry re-specialises the kernel at
run time (hiprtc) with extra compiler options (SPCIES_MFMA4_RTC_FLAGS) and times C2.  usage (GPU box): python tools/sweep_mfma4_flags.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver

FLAGS = sys.argv[1:] or ["", "-DSPCIES_MFMA4_IL=1", "-DSPCIES_MFMA4_IL=2", "-DSPCIES_MFMA4_IL=3", "-DSPCIES_MFMA4_IL=4",
                         "-DSPCIES_MFMA4_IL=2 -DSPCIES_MFMA4_PF=8", "-DSPCIES_NO_SEG_BARRIER=1", "-DSPCIES_NO_SEG_BARRIER=1 -DSPCIES_MFMA4_IL=2"]
cfg = benchmarks.config("C2")
v = benchmarks.ingredients(cfg)
B = 65536
x0, xr, ur = benchmarks.sample_batch(cfg, B)
dev = torch.device("cuda", 0)
tx0, txr, tur = (torch.from_numpy(a).to(dev) for a in (x0, xr, ur))
tu = torch.empty((B, cfg.sys.m), dtype=torch.float64, device=dev); tk = torch.empty(B, dtype=torch.int32, device=dev); te = torch.empty(B, dtype=torch.int32, device=dev)
ref = None
for fl in FLAGS:
    if fl:
        os.environ["SPCIES_MFMA4_RTC_FLAGS"] = fl + " "
    else:
        os.environ.pop("SPCIES_MFMA4_RTC_FLAGS", None)
    try:
        s = HipSolver(v)
        s.set_variant("mfma4")
        s.reserve(B)
        st = torch.cuda.current_stream(dev).cuda_stream
        s.time_device(tx0, txr, tur, tu, tk, te, stream=st, reps=2)
        ms = min(s.time_device(tx0, txr, tur, tu, tk, te, stream=st, reps=5) for _ in range(3))
        u = tu.cpu().numpy().copy()
        if ref is None:
            ref = u
        print(json.dumps(dict(flags=fl or "(built-in)", kernel_ms=round(ms, 3), solves_per_s=round(B / ms * 1e3), du_vs_first=float(np.abs(u - ref).max()))), flush=True)
        s.close()
    except Exception as ex:  # a flag set that does not compile is a data point, not a failure of the sweep
        print(json.dumps(dict(flags=fl, error=str(ex)[:300])), flush=True)
