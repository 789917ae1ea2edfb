"""CPU: hiprtc runs in a process of its own (spcies_amd/spcies_rtc_helper, csrc/rtc_helper.cpp; round 5).

Spcies hands its generated C to `mex` - a compiler in its own process; the HIP platform's run-time specialised kernels used to be compiled
by hiprtc INSIDE the caller, where a compiler crash (ROCm 7.2 has one: DESIGN.md 4.2b''') would take a MATLAB session or a Python process
down.  The library now starts one helper per process, sends it compile requests over a pipe, and treats its death as a failed build.
hiprtc needs no GPU, so all of this runs here: a kernel compiles in the helper; a compiler that dies mid-request fails that build with the
signal in the message while this process lives; the next request gets a fresh helper; SPCIES_HIP_RTC_ISOLATE=0 switches the helper off."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "spcies_amd", "spcies_rtc_helper")
KERNEL = 'extern "C" __global__ void selftest_kernel(double *x) { x[threadIdx.x] = 2.0 * x[threadIdx.x] + %d.0; }\n'

_CHILD = r"""
import ctypes as C, json, os, sys
sys.path.insert(0, sys.argv[1])
from spcies_amd import _lib
lib = _lib.load()
out = []
for src in sys.argv[2:]:
    iso, n = C.c_int(-1), C.c_ulong(0)
    rc = lib.spcies_hip_rtc_compile_selftest(src.encode(), C.byref(iso), C.byref(n))
    out.append({"rc": rc, "isolated": iso.value, "bytes": n.value, "err": lib.spcies_hip_last_error().decode() if rc else ""})
print(json.dumps(out))
"""


def _run(sources, env=None):
    import json
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, "-c", _CHILD, ROOT] + sources, capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr[-2000:]  # the CALLER survives whatever the compiler did
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/lib/libhiprtc.so") and os.path.exists(HELPER)), reason="needs the ROCm installation's hiprtc and the built helper")
def test_compiler_runs_in_its_own_process_and_its_death_is_a_failed_build():
    good1, good2 = KERNEL % 1, KERNEL % 2
    crash = "//SPCIES_RTC_HELPER_SELFTEST_ABORT\n" + KERNEL % 3
    bad = 'extern "C" __global__ void selftest_kernel(double *x) { this is not C++ }\n'
    out = _run([good1, crash, good2, bad, good1])
    assert out[0] == {"rc": 0, "isolated": 1, "bytes": out[0]["bytes"], "err": ""} and out[0]["bytes"] > 1000
    assert out[1]["rc"] != 0 and out[1]["isolated"] == 1 and "died with signal 6" in out[1]["err"]  # abort() in the helper, not here
    assert out[2]["rc"] == 0 and out[2]["bytes"] > 1000                                              # a fresh helper served the next request
    assert out[3]["rc"] != 0 and "hiprtcCompileProgram failed" in out[3]["err"] and "error" in out[3]["err"]  # an ordinary compile error: the log comes back
    assert out[4]["rc"] == 0 and out[4]["bytes"] == out[0]["bytes"]                                  # ... and the same helper goes on


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/libhiprtc.so"), reason="needs the ROCm installation's hiprtc")
def test_isolation_can_be_switched_off():
    out = _run([KERNEL % 1], env={"SPCIES_HIP_RTC_ISOLATE": "0"})
    assert out[0]["rc"] != 0 and out[0]["isolated"] == 0 and "SPCIES_HIP_RTC_ISOLATE" in out[0]["err"]
