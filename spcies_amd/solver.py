"""Generated-solver object of the HIP platform.

The reference's generated MATLAB function is called ``[u, k, e_flag, sol] = <name>(x0, xr, ur)``
(``formulations/+laxMPC/struct_laxMPC_ADMM_C_Matlab.c:8-166``).  :class:`HipSolver` keeps that call
and extends it to a batch: ``x0`` may be ``(n,)`` or ``(B, n)``; ``xr``/``ur`` one shared reference or
one per instance.  Argument checks raise with the reference's message ids
(``Spcies:<formulation>:nrhs:<arg>``, ``struct_laxMPC_ADMM_C_Matlab.c:34-55``).
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _lib, blob as _blob


class SpciesArgError(ValueError):
    """Mirror of ``mexErrMsgIdAndTxt("Spcies:...")`` argument errors; ``.identifier`` holds the id."""

    def __init__(self, identifier, msg):
        super().__init__(f"{identifier}: {msg}")
        self.identifier = identifier


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class HipSolver:
    """Handle on one controller (one problem blob) living on one GPU."""

    def __init__(self, vars_or_blob, device=0, name=None, debug=True):
        self.blob = vars_or_blob if isinstance(vars_or_blob, (bytes, bytearray)) else _blob.pack(vars_or_blob)
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.spcies_hip_create(self.blob, len(self.blob), int(device), C.byref(h)))
        self._h = h
        self._lib = lib
        info = _lib.Info()
        _lib.check(lib.spcies_hip_get_info(h, C.byref(info)))
        self.n, self.m, self.N, self.dim = info.n, info.m, info.N, info.dim
        self.dim_lambda = info.dim_lambda
        self.method = {v: k for k, v in _blob.METHOD.items()}[info.method]
        nf = C.c_int(0)
        dims = (C.c_int * 8)()
        names = (C.c_char_p * 8)()
        _lib.check(lib.spcies_hip_get_sol_layout(h, C.byref(nf), dims, names))
        # record fields of the generated solver, in the reference's order (e.g. z, v, lambda)
        self.sol_fields = [(names[i].decode(), int(dims[i])) for i in range(nf.value)]
        self.device = info.device
        self.formulation = {v: k for k, v in _blob.FORMULATION.items()}[info.formulation]
        self.submethod = int(info.submethod)  # 1 = soc, 2 = split (include/spcies_hip.h)
        self.name = name or self.formulation
        self.time_varying = bool(int.from_bytes(self.blob[28:32], "little") & 4)  # header flags bit2
        self.debug = bool(debug)  # reference option `debug`: copy z, v, lambda out (Spcies_options.m:121)

    # -- lifecycle
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.spcies_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- knobs
    @property
    def variant(self):
        info = _lib.Info()
        _lib.check(self._lib.spcies_hip_get_info(self._h, C.byref(info)))
        return {v: k for k, v in _lib.VARIANTS.items()}[info.variant]

    def set_variant(self, name):
        _lib.check(self._lib.spcies_hip_set_variant(self._h, _lib.VARIANTS[name]))

    @property
    def notes(self):
        """Which faster (run-time specialised) variants AUTO could not use for this controller, and why ('' if none)."""
        p = C.c_char_p()
        _lib.check(self._lib.spcies_hip_get_notes(self._h, C.byref(p)))
        return (p.value or b"").decode()

    def set_exit(self, k_max=0, tol=-1.0):
        _lib.check(self._lib.spcies_hip_set_exit(self._h, int(k_max), float(tol)))

    def reserve(self, B):
        _lib.check(self._lib.spcies_hip_reserve(self._h, C.c_long(int(B))))

    # -- the generated-solver call
    def _check_args(self, x0, xr, ur):
        f = self.formulation
        x0 = np.asarray(x0, dtype=np.float64)
        single = x0.ndim == 1
        x0 = np.ascontiguousarray(np.atleast_2d(x0))
        if x0.ndim != 2 or x0.shape[1] != self.n:
            raise SpciesArgError(f"Spcies:{f}:nrhs:x0", f"x0 must be of dimension {self.n}")
        B = x0.shape[0]
        xr = np.ascontiguousarray(np.asarray(xr, dtype=np.float64))
        ur = np.ascontiguousarray(np.asarray(ur, dtype=np.float64))
        if xr.ndim == 2 and xr.shape[0] == 1 and ur.ndim <= 2:
            xr = xr.reshape(-1)
        if ur.ndim == 2 and ur.shape[0] == 1 and xr.ndim == 1:
            ur = ur.reshape(-1)
        per = xr.ndim == 2
        if (xr.shape != ((B, self.n) if per else (self.n,))):
            raise SpciesArgError(f"Spcies:{f}:nrhs:xr", f"xr must be of dimension {self.n}")
        if (ur.shape != ((B, self.m) if per else (self.m,))):
            raise SpciesArgError(f"Spcies:{f}:nrhs:ur", f"ur must be of dimension {self.m}")
        return x0, xr, ur, B, per, single

    def _pack_model(self, extra_in, B):
        """Time-varying solvers: ``solver(x0, xr, ur, A, B, Q, R, LB, UB)`` as the 9-input mex gateway
        (``struct_laxMPC_ADMM_C_Matlab.c:29-31, 57-103``).  Each of the six may carry a leading batch axis
        (one model per instance).  Packed per instance as the C-ABI's ``extra``: A, B column-major, Q, R, LB, UB."""
        f, n, m = self.formulation, self.n, self.m
        if len(extra_in) != 6:
            raise SpciesArgError(f"Spcies:{f}:nrhs:number", "Nine inputs are required")
        names, shapes = ("A", "B", "Q", "R", "LB", "UB"), ((n, n), (n, m), (n,), (m,), (n + m,), (n + m,))
        arrs, per = [], False
        for name, shp, a in zip(names, shapes, extra_in):
            a = np.asarray(a, dtype=np.float64)
            if a.size == int(np.prod(shp)):
                a = a.reshape(shp)
            elif a.size == B * int(np.prod(shp)) and a.shape[0] == B:
                a = a.reshape((B,) + shp)
                per = True
            else:
                raise SpciesArgError(f"Spcies:{f}:nrhs:{name}", f"{name} must be of dimension {' by '.join(map(str, shp))}")
            arrs.append(a)
        rows = B if per else 1
        cols = []
        for shp, a in zip(shapes, arrs):
            a = np.broadcast_to(a, (rows,) + shp) if a.ndim == len(shp) else a
            if len(shp) == 2:
                a = np.transpose(a, (0, 2, 1))  # column-major, as MATLAB hands it to the mex
            cols.append(np.reshape(a, (rows, -1)))
        model = np.ascontiguousarray(np.hstack(cols))
        return model, (model.shape[1] if per else 0)

    def __call__(self, x0, xr, ur, *extra_in, want_sol=None):
        """``u, k, e_flag, sol = solver(x0, xr, ur)``; host (numpy) buffers in and out.  The ellipMPC soc
        solver takes the ellipsoid radius as a 4th input, ``solver(x0, xr, ur, r)``
        (``struct_ellipMPC_ADMM_soc_C_Matlab.c:24``), scalar or one value per instance."""
        x0, xr, ur, B, per, single = self._check_args(x0, xr, ur)
        extra, extra_stride = None, 0
        if self.formulation == "ellipMPC" and self.submethod == 1:
            if len(extra_in) != 1:
                raise SpciesArgError("Spcies:ellipMPC:nrhs:number", "Four inputs are required")
            extra = np.ascontiguousarray(np.atleast_1d(np.asarray(extra_in[0], dtype=np.float64)).ravel())
            if extra.size not in (1, B):
                raise SpciesArgError("Spcies:ellipMPC:nrhs:r", "r must be a scalar (or one value per instance)")
            extra_stride = 1 if (extra.size == B and B > 1) else 0
        elif self.time_varying:
            extra, extra_stride = self._pack_model(extra_in, B)
        elif extra_in:
            raise SpciesArgError(f"Spcies:{self.formulation}:nrhs:number", "Three inputs are required")
        want_sol = self.debug if want_sol is None else want_sol
        u = np.zeros((B, self.m))
        k = np.zeros(B, dtype=np.int32)
        e = np.zeros(B, dtype=np.int32)
        dp = C.POINTER(C.c_double)
        arrays = [np.zeros((B, d)) for _, d in self.sol_fields] if want_sol else None
        ptrs = (dp * len(self.sol_fields))(*[_dp(a) for a in arrays]) if want_sol else None
        t = _lib.Timing()
        self._solve_host(x0, xr, ur, per, extra, extra_stride, B, u, k, e, ptrs, t)
        fields = {name: (arrays[i] if want_sol else None) for i, (name, _) in enumerate(self.sol_fields)}
        if single:
            fields = {kf: (a[0] if a is not None else None) for kf, a in fields.items()}
        sol = SimpleNamespace(update_time=t.update_time, solve_time=t.solve_time, polish_time=t.polish_time,
                              run_time=t.run_time, **fields)
        sol.lam = fields.get("lambda")  # `lambda` is a Python keyword: sol.lam is the same array
        if "v" not in fields:
            sol.v = None
        if single:
            return u[0], int(k[0]), int(e[0]), sol
        return u, k, e, sol

    def _solve_host(self, x0, xr, ur, per, extra, extra_stride, B, u, k, e, ptrs, t):
        _lib.check(self._lib.spcies_hip_solve_batch_ex(
            self._h, _dp(x0), _dp(xr), _dp(ur), C.c_int(int(per)), _dp(extra) if extra is not None else None,
            C.c_int(int(extra_stride)), C.c_long(B),
            _dp(u), _ip(k), _ip(e), ptrs, len(self.sol_fields), C.byref(t)))

    def closed_loop(self, AB, x0, xr, ur, steps):
        """Closed-loop simulation of B plants on the device (``examples/cl_in_C/main_cl_in_C.c:98-117``): at every
        sample time solve, then ``x+ = A x + B u`` with the plant ``AB = [A B]``.  Returns
        ``x_traj (steps+1, B, n), u_traj (steps, B, m), k_traj, e_traj (steps, B), timing``."""
        x0, xr, ur, B, per, single = self._check_args(x0, xr, ur)
        AB = np.ascontiguousarray(np.asarray(AB, dtype=np.float64))
        if AB.shape != (self.n, self.n + self.m):
            raise SpciesArgError(f"Spcies:{self.formulation}:closed_loop:AB", f"AB must be {self.n} by {self.n + self.m}")
        steps = int(steps)
        xt = np.zeros((steps + 1, B, self.n)); ut = np.zeros((steps, B, self.m))
        kt = np.zeros((steps, B), dtype=np.int32); et = np.zeros((steps, B), dtype=np.int32)
        t = _lib.Timing()
        _lib.check(self._lib.spcies_hip_closed_loop(self._h, _dp(AB), _dp(x0), _dp(xr), _dp(ur), C.c_int(int(per)), C.c_long(B), C.c_int(steps), _dp(xt),
                                                    _dp(ut), _ip(kt), _ip(et), C.byref(t)))
        return xt, ut, kt, et, t

    def solve_device(self, x0, xr, ur, u, k, e_flag, z=None, v=None, lam=None, stream=0):
        """Device-resident call: arguments are objects with ``data_ptr()`` (torch tensors on this GPU)
        or raw integer device addresses; asynchronous on ``stream`` (a raw ``hipStream_t`` value)."""
        ptr = lambda a: None if a is None else C.c_void_p(a if isinstance(a, int) else a.data_ptr())
        B = x0.shape[0]
        per = 1 if xr.dim() == 2 else 0
        _lib.check(self._lib.spcies_hip_solve_batch_device(self._h, ptr(x0), ptr(xr), ptr(ur), C.c_int(per), C.c_long(B), ptr(u), ptr(k),
                                                          ptr(e_flag), ptr(z), ptr(v), ptr(lam), C.c_void_p(stream)))

    def solve_device_ex(self, x0, xr, ur, u, k, e_flag, extra=None, extra_stride=0, fields=None, stream=0):
        """Device-resident call through ``spcies_hip_solve_batch_device_ex``: any solver, its own record
        (``fields``: one tensor or ``None`` per entry of ``sol_fields``) and its extra input (``extra``)."""
        ptr = lambda a: None if a is None else C.c_void_p(a if isinstance(a, int) else a.data_ptr())
        nf = len(self.sol_fields)
        fp = None
        if fields is not None:
            if len(fields) != nf:
                raise SpciesArgError(f"Spcies:{self.formulation}:sol:fields", f"this solver's record has {nf} fields")
            fp = (C.c_void_p * nf)(*[ptr(f) for f in fields])
        per = 1 if xr.dim() == 2 else 0
        _lib.check(self._lib.spcies_hip_solve_batch_device_ex(self._h, ptr(x0), ptr(xr), ptr(ur), C.c_int(per), ptr(extra),
                                                             C.c_int(int(extra_stride)), C.c_long(x0.shape[0]), ptr(u), ptr(k),
                                                             ptr(e_flag), fp, nf, C.c_void_p(stream)))

    def k_histogram(self, k, e_flag, n_bins=10, stream=0):
        """Batch statistics of a device solve (the batch counterpart of the MATLAB solvers' ``genHist`` record):
        ``k``, ``e_flag`` are the int32 DEVICE arrays a device solve wrote.  Returns ``SimpleNamespace(hist, converged,
        k_max_reached, other, mean_k)`` - ``hist[b]`` counts instances with ``b k_max / n_bins < k <= (b + 1) k_max / n_bins``."""
        ptr = lambda a: C.c_void_p(a if isinstance(a, int) else a.data_ptr())
        B = int(k.shape[0])
        hist = (C.c_long * int(n_bins))()
        counts = (C.c_long * 4)()
        _lib.check(self._lib.spcies_hip_k_histogram_device(self._h, ptr(k), ptr(e_flag), C.c_long(B), int(n_bins), hist, counts,
                                                          C.c_void_p(stream)))
        return SimpleNamespace(hist=np.array(hist[:], dtype=np.int64), converged=int(counts[0]), k_max_reached=int(counts[1]),
                               other=int(counts[2]), mean_k=(counts[3] / B if B else 0.0))

    def residual_trace(self, x0, xr, ur, K):
        """Residual history of the lax / equ MPC ADMM solvers (the dense MATLAB solvers' ``hRp`` / ``hRd`` record,
        ``spcies_laxMPC_ADMM_solver.m:253-261, 311-319``): returns ``SimpleNamespace(r_p, r_d, k)`` with ``r_p[i, j-1] = ||z - v||_inf`` and
        ``r_d[i, j-1] = ||v - v_prev||_inf`` of iteration ``j`` of instance ``i`` (zero behind the iteration it leaves at, ``k[i]``)."""
        x0 = np.ascontiguousarray(np.atleast_2d(np.asarray(x0, dtype=np.float64)))
        B = x0.shape[0]
        xr = np.ascontiguousarray(np.asarray(xr, dtype=np.float64))
        ur = np.ascontiguousarray(np.asarray(ur, dtype=np.float64))
        per = 1 if xr.ndim == 2 else 0
        rp, rd, kx = np.zeros((B, int(K))), np.zeros((B, int(K))), np.zeros(B, dtype=np.int32)
        _lib.check(self._lib.spcies_hip_residual_trace(self._h, _dp(x0), _dp(xr), _dp(ur), C.c_int(per), C.c_long(B), int(K), _dp(rp), _dp(rd), _ip(kx)))
        return SimpleNamespace(r_p=rp, r_d=rd, k=kx)

    def time_device(self, x0, xr, ur, u, k, e_flag, stream=0, reps=1):
        """Mean ms per launch over ``reps`` back-to-back solves, hipEvents on ``stream``."""
        ptr = lambda a: C.c_void_p(a.data_ptr())
        ms = C.c_double(0)
        per = 1 if xr.dim() == 2 else 0
        _lib.check(self._lib.spcies_hip_time_device(self._h, ptr(x0), ptr(xr), ptr(ur), C.c_int(per), C.c_long(x0.shape[0]), ptr(u),
                                                   ptr(k), ptr(e_flag), C.c_void_p(stream), int(reps), C.byref(ms)))
        return ms.value


class MultiHipSolver(HipSolver):
    """One controller on several GPUs of one process (``spcies_hip_create_multi``): the host batch is split into
    contiguous shards, one host thread and one single-device handle per entry of ``devices`` (``None`` = every
    visible device; a device may be listed more than once).  Same call as :class:`HipSolver`; device-resident
    entry points are per device: ``self.single(i)``."""

    def __init__(self, vars_or_blob, devices=None, name=None, debug=True):
        self.blob = vars_or_blob if isinstance(vars_or_blob, (bytes, bytearray)) else _blob.pack(vars_or_blob)
        lib = _lib.load()
        mh = C.c_void_p()
        ids = None if devices is None else (C.c_int * len(devices))(*[int(d) for d in devices])
        _lib.check(lib.spcies_hip_create_multi(self.blob, len(self.blob), ids, 0 if devices is None else len(devices), C.byref(mh)))
        self._mh = mh
        self._lib = lib
        nd = C.c_int(0)
        _lib.check(lib.spcies_hip_multi_count(mh, C.byref(nd)))
        self.n_dev = nd.value
        self._h = self.single(0)  # info, record layout: the same on every device
        info = _lib.Info()
        _lib.check(lib.spcies_hip_get_info(self._h, C.byref(info)))
        self.n, self.m, self.N, self.dim, self.dim_lambda = info.n, info.m, info.N, info.dim, info.dim_lambda
        self.method = {v: k for k, v in _blob.METHOD.items()}[info.method]
        nf, dims, names = C.c_int(0), (C.c_int * 8)(), (C.c_char_p * 8)()
        _lib.check(lib.spcies_hip_get_sol_layout(self._h, C.byref(nf), dims, names))
        self.sol_fields = [(names[i].decode(), int(dims[i])) for i in range(nf.value)]
        self.device = info.device
        self.formulation = {v: k for k, v in _blob.FORMULATION.items()}[info.formulation]
        self.submethod = int(info.submethod)
        self.name = name or self.formulation
        self.time_varying = bool(int.from_bytes(self.blob[28:32], "little") & 4)
        self.debug = bool(debug)

    def single(self, i):
        h = C.c_void_p()
        _lib.check(self._lib.spcies_hip_multi_get(self._mh, int(i), C.byref(h)))
        return h

    def close(self):
        if getattr(self, "_mh", None) is not None and self._mh:
            self._lib.spcies_hip_multi_destroy(self._mh)
            self._mh = None
            self._h = None

    def set_variant(self, name):
        _lib.check(self._lib.spcies_hip_multi_set_variant(self._mh, _lib.VARIANTS[name]))

    def set_exit(self, k_max=0, tol=-1.0):
        _lib.check(self._lib.spcies_hip_multi_set_exit(self._mh, int(k_max), float(tol)))

    def reserve(self, B):
        for i in range(self.n_dev):
            _lib.check(self._lib.spcies_hip_reserve(self.single(i), C.c_long(-(-int(B) // self.n_dev))))

    def _solve_host(self, x0, xr, ur, per, extra, extra_stride, B, u, k, e, ptrs, t):
        width = int(extra.size // B) if (extra is not None and extra_stride) else 0
        _lib.check(self._lib.spcies_hip_multi_solve_batch_ex(
            self._mh, _dp(x0), _dp(xr), _dp(ur), C.c_int(int(per)), _dp(extra) if extra is not None else None,
            C.c_int(int(extra_stride)), C.c_long(width), C.c_long(B), _dp(u), _ip(k), _ip(e), ptrs, len(self.sol_fields), C.byref(t)))

    def closed_loop(self, *a, **kw):
        raise SpciesArgError(f"Spcies:{self.formulation}:closed_loop:multi", "closed_loop runs on one device: use HipSolver")
