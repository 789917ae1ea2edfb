"""Generate ``tests/golden/`` fixtures (run in THIS container; needs /root/reference).  TEST INFRASTRUCTURE.

1. ``reference_z_opt.json`` - the hard-coded optimum vectors the reference's own tests hold
   (``tests/test_<form>_<method>.m``, variable ``z_opt``) and the discretised 3-mass ``[A B]`` matrix
   embedded in ``examples/cl_in_C/main_cl_in_C.c:96``.  Pure data, extracted verbatim.
2. ``template_<cfg>.npz`` - inputs and outputs of the reference's C solver template instantiated by
   ``oracle/ref_template.py`` (see its header for exactly what is the reference's and what is ours) on
   seeded instances, plus the C oracle's outputs on the same inputs.
"""
from __future__ import annotations

import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def extract_z_opt():
    out = {}
    for fn in sorted(os.listdir(os.path.join(REF, "tests"))):
        if not fn.startswith("test_"):
            continue
        s = open(os.path.join(REF, "tests", fn)).read()
        mm = re.search(r"z_opt\s*=\s*\[([^\]]*)\]", s)
        if mm:
            out[fn[:-2]] = [float(x) for x in mm.group(1).replace(";", " ").split()]
    s = open(os.path.join(REF, "examples", "cl_in_C", "main_cl_in_C.c")).read()
    mm = re.search(r"double AB\[6\]\[8\]\s*=\s*\{(.*?)\};", s, re.S)
    rows = re.findall(r"\{([^{}]*)\}", mm.group(1))
    out["main_cl_in_C_AB"] = [[float(x) for x in r.split(",")] for r in rows]
    return out


def main():
    from oracle import oracle, ref_template
    from spcies_amd import benchmarks

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "reference_z_opt.json"), "w") as f:
        json.dump(extract_z_opt(), f, indent=0)
    for name, B, overrides in (("C1_lax", 16, {}), ("C1_lax_denseT", 16, {}), ("C1_equ", 16, {}),
                               ("C2_lax", 32, {}), ("C2_lax", 32, dict(tol=1e-6, k_max=3000)),
                               ("C2_equ", 32, dict(tol=1e-6, k_max=3000)),
                               ("C1_lax_FISTA", 16, {}), ("C1_equ_FISTA", 16, dict(k_max=500)),
                               ("C2_lax_FISTA", 32, {}), ("C2_lax_FISTA", 32, dict(tol=1e-6, k_max=2000)),
                               ("C2_equ_FISTA", 32, {}), ("C1_MPCT", 16, {}), ("C4", 8, {}), ("C1_MPCT_nd", 16, {}), ("C1_MPCT_nd0", 8, {}), ("C4_nd", 6, {}),
                               ("C1_ellip", 12, {}), ("C2_ellip", 8, {}), ("C1_ellip_vec", 12, {}), ("C2_ellip_vec", 8, {}), ("C1_ellip_inc", 12, {}),
                               ("C1_lax_gen", 12, {}), ("C1_equ_gen", 12, dict(k_max=3000)), ("C2_lax_gen", 8, {}),
                               ("C1_soc", 12, {}), ("C5_soc", 8, {}), ("C1_soc_inc", 12, {}),
                               ("C1_HMPC", 6, {}), ("C1_HMPC_SADMM", 6, {}), ("C1_HMPC_soc", 4, {}),
                               ("C1_HMPC_SADMM_soc", 4, {}), ("C5_HMPC_SADMM", 4, {}),
                               ("C1_HMPCcc", 6, {}), ("C1_HMPCcc_SADMM", 6, {}), ("C1_HMPCcc_soc", 4, {}),
                               ("C1_HMPC_nosplit", 6, {}), ("C1_HMPC_SADMM_nosplit", 6, {}), ("C1_HMPC_soc_nosplit", 4, {}),
                               ("C1_HMPC_SADMM_soc_nosplit", 4, {}), ("C5_HMPC_SADMM_nosplit", 4, {}),
                               ("C1_HMPCcc_nosplit", 6, {}), ("C1_HMPCcc_SADMM_nosplit", 6, {}), ("C1_HMPCcc_soc_nosplit", 4, {}),
                               ("C1_MPCT_cs", 8, {}), ("C1_MPCT_cs_vec", 6, {}), ("C2_cs", 6, {}), ("C4_cs", 4, {})):
        if len(sys.argv) > 1 and name not in sys.argv[1:]:  # `python -m oracle.make_golden C1_ellip ...`: only these
            continue
        cfg = benchmarks.config(name)
        v = benchmarks.ingredients(cfg, **overrides)
        x0, xr, ur = benchmarks.sample_batch(cfg, B)
        if name.startswith("C1"):  # first instance = the reference tester's own status
            st = benchmarks.tester_status(cfg.sys)
            x0[0], xr[0], ur[0] = st.x, st.xr, st.ur
        tag = name + ("_conv" if overrides else "")
        so = ref_template.build_admm(v, "golden_" + tag)
        if v["formulation"] == "HMPC" and v.get("submethod") != "split":
            T = ref_template.run_hmpc_nosplit(so, v, x0, xr, ur)
            O = oracle.hmpc_dense_batch(v, x0, xr, ur)
            print(tag, "template-vs-oracle(full doubles): z %.2e s %.2e dk %d" % (
                np.abs(T[3] - O[3]).max(), np.abs(T[4] - O[4]).max(), np.abs(T[1] - O[1]).max()))
            np.savez_compressed(os.path.join(OUT, f"template_{tag}.npz"), x0=x0, xr=xr, ur=ur, u=T[0], k=T[1],
                                e_flag=T[2], z=T[3], s=T[4], lam=T[5], solver_overrides=json.dumps(overrides))
            continue
        if v.get("submethod") == "split":
            T = ref_template.run_hmpc(so, v, x0, xr, ur)
            O = oracle.admm_hmpc_batch(v, x0, xr, ur)
            print(tag, "template-vs-oracle(full doubles): z %.2e s %.2e dk %d" % (
                np.abs(T[3] - O[3]).max(), np.abs(T[4] - O[4]).max(), np.abs(T[1] - O[1]).max()))
            np.savez_compressed(os.path.join(OUT, f"template_{tag}.npz"), x0=x0, xr=xr, ur=ur, u=T[0], k=T[1],
                                e_flag=T[2], z=T[3], s=T[4], z_hat=T[5], s_hat=T[6], lam=T[7], mu=T[8],
                                solver_overrides=json.dumps(overrides))
            continue
        if v.get("submethod") == "soc":
            r = cfg.param.r + 0.01 * np.arange(B)
            if name == "C1_soc":
                r[0] = cfg.param.r
            T = ref_template.run_soc(so, v, x0, xr, ur, r)
            O = oracle.admm_soc_batch(v, x0, xr, ur, r)
            print(tag, "template-vs-oracle(full doubles): z %.2e s %.2e dk %d" % (
                np.abs(T[3] - O[3]).max(), np.abs(T[4] - O[4]).max(), np.abs(T[1] - O[1]).max()))
            np.savez_compressed(os.path.join(OUT, f"template_{tag}.npz"), x0=x0, xr=xr, ur=ur, r=r, u=T[0], k=T[1],
                                e_flag=T[2], z=T[3], s=T[4], z_hat=T[5], s_hat=T[6], lam=T[7], mu=T[8],
                                solver_overrides=json.dumps(overrides))
            continue
        if v["method"] == "EADMM":
            ut, kt, et, z1t, z2t, z3t, lt = ref_template.run_admm(so, v, x0, xr, ur)
            O = oracle.eadmm_mpct_batch(v, x0, xr, ur)
            print(tag, "template-vs-oracle(full doubles): z1 %.2e z3 %.2e dk %d" % (
                np.abs(z1t - O[3]).max(), np.abs(z3t - O[5]).max(), np.abs(kt - O[1]).max()))
            np.savez_compressed(os.path.join(OUT, f"template_{tag}.npz"), x0=x0, xr=xr, ur=ur, u=ut, k=kt, e_flag=et,
                                z1=z1t, z2=z2t, z3=z3t, lam=lt, solver_overrides=json.dumps(overrides))
            continue
        ut, kt, et, zt, vt, lt = ref_template.run_admm(so, v, x0, xr, ur)
        if v.get("submethod") == "cs":
            uo, ko, eo, zo, vo, lo = oracle.mpct_cs_batch(v, x0, xr, ur)
        elif v["method"] == "FISTA":
            uo, ko, eo, zo, lo = oracle.fista_banded_batch(v, x0, xr, ur)
            vt = np.zeros((0,))
        else:
            uo, ko, eo, zo, vo, lo = oracle.admm_banded_batch(v, x0, xr, ur)
        print(tag, "template-vs-oracle(full doubles): z %.2e lam %.2e  dk %d" % (
            np.abs(zt - zo).max(), np.abs(lt - lo).max(), np.abs(kt - ko).max()))
        np.savez_compressed(os.path.join(OUT, f"template_{tag}.npz"), x0=x0, xr=xr, ur=ur, u=ut, k=kt, e_flag=et,
                            z=zt, v=vt, lam=lt, solver_overrides=json.dumps(overrides))


if __name__ == "__main__":
    main()
