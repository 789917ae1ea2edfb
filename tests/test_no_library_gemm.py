"""AUTO never hands a product to a library (round-4 review, "rocBLAS on AUTO fallbacks"): when the hand-written FUSED kernel of an HMPC
controller is unavailable (hiprtc off and the shape not built in), AUTO resolves to the library's own TILE / STREAM kernels, the process
never maps librocblas, and the results are the oracle's.  The rocBLAS variant GEMM stays reachable by name only (cross-check)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_auto_never_names_gemm_in_the_source():
    """Static half (CPU): `resolve_variant` has no path to SPCIES_VARIANT_GEMM."""
    src = open(os.path.join(ROOT, "spcies_amd", "csrc", "spcies_hip.hip")).read()
    body = src[src.index("static int resolve_variant(const Solver &s) {"):]
    body = body[:body.index("\n}\n")]
    assert "if (s.variant != SPCIES_VARIANT_AUTO) return s.variant;" in body
    code = "\n".join(line.split("//")[0] for line in body.splitlines())
    assert "GEMM" not in code, code


@pytest.mark.gpu
def test_auto_without_fused_runs_the_librarys_own_kernels():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_no_gemm_child.py")], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["rocblas_mapped"] is False
    assert out["C1_HMPC_SADMM"]["variant"] in ("tile", "stream") and out["C1_HMPC_nosplit"]["variant"] == "stream"
    assert out["C1_HMPCcc_nosplit"]["variant"] == "stream"
    for name in ("C1_HMPC_SADMM", "C1_HMPC_nosplit", "C1_HMPCcc_nosplit"):
        c = out[name]
        assert "FUSED" in c["notes"], c  # spcies_hip_get_notes says what AUTO gave up
        assert c["k_equal"] and c["du"] <= 1e-10 and c["dz"] <= 1e-10, (name, c)
