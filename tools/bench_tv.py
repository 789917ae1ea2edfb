"""Time-varying laxMPC-ADMM at the C2 shape (one model per instance): device buffers, kernel time by HIP events around the whole
solve (update phase + inverses + iteration).  usage: python tools/bench_tv.py [B] [variant] [k_max|0] [config: C2_lax | C2_lax_FISTA | ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
variant = sys.argv[2] if len(sys.argv) > 2 else "auto"
cfg = benchmarks.config(sys.argv[4] if len(sys.argv) > 4 else "C2_lax")
vt = benchmarks.ingredients(cfg, time_varying=True)
torch.cuda.init()  # torch's bundled ROCm user space first, the library's second (tests/conftest.py: the other order loses torch its GPU)
s = HipSolver(vt)
if variant != "auto":
    s.set_variant(variant)
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:
    s.set_exit(k_max=int(sys.argv[3]))
sysm, prm = cfg.sys, cfg.param
LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)]); UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
rng = np.random.default_rng(5)
j = lambda a, sc: np.asarray(a, float)[None] * (1.0 + sc * (2 * rng.random((B,) + np.shape(a)) - 1))
model, stride = s._pack_model((j(sysm.A, 0.02), j(sysm.B, 0.02), j(np.diag(prm.Q), 0.02), j(np.diag(prm.R), 0.02), j(LB, 0.05), j(UB, 0.05)), B)
x0, xr, ur = benchmarks.sample_batch(cfg, B)
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(a).to(dev)
tx0, txr, tur, tm = t(x0), t(xr), t(ur), t(model)
tu = torch.empty((B, cfg.sys.m), dtype=torch.float64, device=dev); tk = torch.empty(B, dtype=torch.int32, device=dev); te = torch.empty(B, dtype=torch.int32, device=dev)
s.reserve(B)
st = torch.cuda.current_stream(dev).cuda_stream
run = lambda: s.solve_device_ex(tx0, txr, tur, tu, tk, te, extra=tm, extra_stride=stride, stream=st)
run(); torch.cuda.synchronize()
times = []
for _ in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1))
ms = min(times)
print(json.dumps(dict(config=cfg.name + " time-varying, one model per instance", B=B, variant=s.variant, ms=round(ms, 2), solves_per_s=round(B / ms * 1e3),
                      k_unique=np.unique(tk.cpu().numpy()).tolist()[:3])))
