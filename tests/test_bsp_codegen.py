"""CPU: the BSP variant's code generator (spcies_amd/csrc/soc_bsp.hpp).  `spcies_hip_create` generates the controller's block
program while it parses the blob - before it needs a device - so the generated HIP source can be captured here
(SPCIES_BSP_DUMP), compiled for gfx950 with the installed hiprtc (which runs without a GPU) and inspected: it must compile, stay
under the spill bound finish_soc accepts (640 B of scratch per lane; the plain program without the split tail and with the 12-deep
ring keeps everything in registers) and issue exactly one MFMA per table block on either path through the iteration."""
import ctypes as C
import os
import re
import subprocess

import pytest

from spcies_amd import _lib, benchmarks, blob

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
HIPRTC = "/opt/rocm/lib/libhiprtc.so"

_COMPILE = r"""
import ctypes as C, sys
rt = C.CDLL(sys.argv[1])
src = open(sys.argv[2], "rb").read()
prog = C.c_void_p()
assert rt.hiprtcCreateProgram(C.byref(prog), src, b"prog.hip", 0, None, None) == 0
opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-fno-honor-nans"] + [o.encode() for o in sys.argv[4:]]
arr = (C.c_char_p * len(opts))(*opts)
rc = rt.hiprtcCompileProgram(prog, len(opts), arr)
n = C.c_size_t()
rt.hiprtcGetProgramLogSize(prog, C.byref(n))
log = C.create_string_buffer(n.value + 1)
rt.hiprtcGetProgramLog(prog, log)
if rc != 0:
    sys.exit(log.value.decode()[:2000])
rt.hiprtcGetCodeSize(prog, C.byref(n))
code = C.create_string_buffer(n.value)
rt.hiprtcGetCode(prog, code)
open(sys.argv[3], "wb").write(code.raw)
"""


def _compile(src_path, co_path, extra=()):
    """hiprtc in a process of its own: this one may have loaded another ROCm user-space (PyTorch bundles an older comgr, which
    the loader would then hand to the installed hiprtc as well - mfma4_rtc.hpp opens a private link namespace for that case)."""
    import sys
    r = subprocess.run([sys.executable, "-c", _COMPILE, HIPRTC, str(src_path), str(co_path), *extra], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    notes = subprocess.run([READELF, "--notes", str(co_path)], capture_output=True, text=True).stdout
    kernels = dict(re.findall(r"\.name:\s+(soc_bsp_kernel\w*)\s+\.private_segment_fixed_size:\s+(\d+)", notes))
    lds = [int(x) for x in re.findall(r"\.group_segment_fixed_size:\s+(\d+)", notes)]
    return kernels, lds


def _generate(cfg_name, path, monkeypatch):
    monkeypatch.setenv("SPCIES_BSP_DUMP", str(path))
    lib = _lib.load()
    b = blob.pack(benchmarks.ingredients(benchmarks.config(cfg_name)))
    h = C.c_void_p()
    rc = lib.spcies_hip_create(b, len(b), 0, C.byref(h))
    if rc == 0:  # a GPU is present: the solver exists, the dump was written all the same
        lib.spcies_hip_destroy(h)
    return open(path).read()


@pytest.mark.parametrize("cfg_name,n_blocks", [("C1_soc", 261), ("C5_soc", 1015)])
def test_bsp_program_is_generated_compiles_and_keeps_state_in_registers(cfg_name, n_blocks, tmp_path, monkeypatch):
    """The round-2 form (SPCIES_BSP_SCHED=0): the generator's own order, left to LLVM's scheduler."""
    if not (os.path.exists(HIPRTC) and os.path.exists(READELF)):
        pytest.skip("needs the ROCm installation's hiprtc and llvm-readelf")
    monkeypatch.setenv("SPCIES_BSP_SCHED", "0")
    src = _generate(cfg_name, tmp_path / "prog.hip", monkeypatch)
    assert "soc_bsp_kernel" in src
    # the iteration: a head, then ONE wave-uniform branch `if (all_hit) { tail without residual checks } else { tail }` (soc_bsp.hpp);
    # every table block is consumed by exactly one MFMA on either path, and refilled into the ring once
    body = src[src.index("while (true)"):]
    i_if, i_else = body.index("if (all_hit) {"), body.index("} else {", body.index("if (all_hit) {"))
    head, light, full = body[:i_if], body[i_if:i_else], body[i_else:]
    n_mf = lambda t: len(re.findall(r"\bMF\(", t))
    assert n_mf(head) + n_mf(light) == n_blocks and n_mf(light) == n_mf(full)
    assert "ZUPD_L(" in light and "ZUPD(" not in light.replace("ZUPD_L(", "") and "ZUPD(" in full and "ZUPD_L(" not in full
    assert len(re.findall(r"= BLK\(blk\d, \d+\);", head + full)) >= n_blocks
    kernels, lds = _compile(tmp_path / "prog.hip", tmp_path / "prog.co")
    assert set(kernels) == {"soc_bsp_kernel", "soc_bsp_kernel_sol"}
    assert lds and max(lds) <= 160 * 1024
    assert int(kernels["soc_bsp_kernel"]) <= 640  # the default 20-deep ring: within the bound finish_soc accepts (0 at C1, 504 B at C5)


def test_bsp_plain_program_has_no_scratch_at_c5(tmp_path, monkeypatch):
    if not (os.path.exists(HIPRTC) and os.path.exists(READELF)):
        pytest.skip("needs the ROCm installation's hiprtc and llvm-readelf")
    monkeypatch.setenv("SPCIES_BSP_SCHED", "0")
    monkeypatch.setenv("SPCIES_BSP_PF", "12")     # the round-1 program: one copy of the tail, 12-deep ring
    monkeypatch.setenv("SPCIES_BSP_NOSPLIT", "1")
    src = _generate("C5_soc", tmp_path / "prog.hip", monkeypatch)
    assert "if (all_hit) {" not in src and len(re.findall(r"\bMF\(", src)) == 1015 + 1  # + the macro definition
    kernels, _ = _compile(tmp_path / "prog.hip", tmp_path / "prog.co")
    assert int(kernels["soc_bsp_kernel"]) == 0


@pytest.mark.parametrize("cfg_name,n_blocks", [("C1_soc", 261), ("C5_soc", 1015)])
def test_scheduled_bsp_program(cfg_name, n_blocks, tmp_path, monkeypatch):
    """The default form (round 3): the iteration as micro-operations ordered by bsp_sched.hpp, block pairs through ds_read_b128, the
    distinct bound patterns in registers.  Every block is consumed by exactly one product on either path through the iteration, every
    pair of the table is read into the ring exactly once per iteration, each value is defined before it is used (the program is
    straight-line C: the compiler checks that), and the ITERATION touches no scratch memory."""
    if not (os.path.exists(HIPRTC) and os.path.exists(READELF)):
        pytest.skip("needs the ROCm installation's hiprtc and llvm-readelf")
    src = _generate(cfg_name, tmp_path / "prog.hip", monkeypatch)
    body = src[src.index("while (true)"):]
    i_if, i_else = body.index("if (all_hit) {"), body.index("} else {", body.index("if (all_hit) {"))
    i_end = body.index("res |= (rd_ > tol_d)")
    head, light, full = body[:i_if], body[i_if:i_else], body[i_else:i_end]
    n_mf = lambda t: len(re.findall(r"__builtin_amdgcn_mfma_f64_4x4x4f64\(", t))
    assert n_mf(head) + n_mf(light) == n_blocks and n_mf(light) == n_mf(full)
    n_pairs = (n_blocks + 1) // 2
    refills = lambda t: re.findall(r"a\d+ = PBLK\(blk(\d), (\d+)\);", t)
    tail_refills = refills(body[i_end:body.index("exit test per instance")])
    for path in (light, full):
        seen = sorted(int(a) * 256 + int(b) for a, b in refills(head) + refills(path) + tail_refills)
        assert seen == list(range(n_pairs)), "every pair of the table enters the ring once per iteration"
    assert "rd_ = fmax(rd_" in full and "rd_ = fmax(rd_" not in light  # the checks live in one branch only
    assert "lbv[" in src and "LBR(" not in head + light + full  # bounds: register patterns, no LDS read in the iteration
    kernels, lds = _compile(tmp_path / "prog.hip", tmp_path / "prog.co", ["-mllvm", "-enable-misched=false"])
    assert set(kernels) == {"soc_bsp_kernel", "soc_bsp_kernel_sol"}
    assert lds and max(lds) <= 160 * 1024
    assert int(kernels["soc_bsp_kernel"]) <= 640


def test_scheduler_respects_dependences_and_latencies():
    """bsp_sched.hpp on a toy program (host-only test binary): a chain D -> U -> D with independent fillers; the order must be a
    topological order of the read / write dependences, and the fillers must sit between the links of the chain."""
    import subprocess
    import sys
    import tempfile
    src = r'''
#include "bsp_sched.hpp"
using namespace spcies::bsp::sched;
int main() {
    Program P;
    double b[16] = {0};
    P.stmt(K_VALU, "x0", "", {}, {"x0"}, 1);
    P.mfma("y0", true, b, "x0");       // D0
    P.mfma("r1", true, b, "y0");       // U10 (critical)
    P.mfma("y1", true, b, "r1");       // D1
    P.mfma("f0", true, b, "x0");       // fillers: independent of the chain
    P.mfma("f1", true, b, "x0");
    P.mfma("f2", true, b, "x0");
    P.mfma("f3", true, b, "x0");
    P.stmt(K_VALU, "w", "", {"y1", "f0", "f1", "f2", "f3"}, {"w"}, 1);
    std::vector<int> ord = schedule(P, 64);
    std::vector<int> pos(ord.size());
    for (size_t i = 0; i < ord.size(); i++) pos[ord[i]] = (int)i;
    // dependences
    if (!(pos[0] < pos[1] && pos[1] < pos[2] && pos[2] < pos[3] && pos[3] < pos[8])) return 1;
    for (int f = 4; f < 8; f++) if (!(pos[0] < pos[f] && pos[f] < pos[8])) return 2;
    // the fillers hide the chain's latency: D0, U10, D1 are never adjacent
    if (pos[2] == pos[1] + 1 || pos[3] == pos[2] + 1) return 3;
    // model time: 7 products on the pipe (28 quads) + the two vector statements, no stall beyond the chain's ends
    int last = 0;
    for (const Op &o : P.ops) last = std::max(last, o.issue + o.cost);
    printf("%d\\n", last);
    return last <= 40 ? 0 : 4;
}
'''
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spcies_amd", "csrc")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.run(["g++", "-std=c++17", "-O1", "-I", inc, "-o", os.path.join(d, "t"), os.path.join(d, "t.cpp")], check=True)
        r = subprocess.run([os.path.join(d, "t")], capture_output=True, text=True)
        assert r.returncode == 0, (r.returncode, r.stdout)


def test_admm_r_kernels_compile_with_the_flags_the_library_uses(tmp_path):
    """admm_r_kernel.inc (lax / equ ADMM past MFMA4's register file) is specialised with hiprtc INSIDE the caller's process, so a
    compiler crash there takes the caller down: ROCm 7.2's "Rewrite AGPR-Copy-MFMA" pass does crash on these kernels under
    -amdgpu-mfma-vgpr-form when the allocator spills (admm_r.hip says why the switch is not used).  The instantiations that showed it -
    the record kernel at n = 12, N = 30 and the equMPC solve kernel at n = 20, N = 20 - compile here, in a process of their own, with the
    library's options, and stay within the CU's LDS."""
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(here, "spcies_amd", "csrc", "admm_r_kernel.inc")).read()
    sig = "(Args, const double *, const double *, const double *, const double *, double *, int *, int *, double *, double *, double *, double *, double *)"
    src += "\nnamespace spcies { namespace ar {\n"
    # <N, KX, KS, TERMINAL, WANT_SOL, NW, NLDS, UNIT, GEN>: the two round-4 instantiations in plain coordinates, and the round-5 forms (unit-box
    # coordinates; vector rho / stage-wise bounds with the row constants in the chunk stream)
    for targs in ("30, 3, 4, false, true, 4, 69, false, false", "20, 5, 6, false, false, 4, 55, false, false",
                  "30, 3, 4, true, true, 4, 69, true, false", "20, 5, 6, true, false, 4, 53, true, true"):
        src += f"template __global__ void admm_r_kernel<{targs}>{sig};\n"
    src += "}}\n"
    p = tmp_path / "admm_r_check.hip"
    p.write_text(src)
    import sys
    co = tmp_path / "admm_r_check.co"
    r = subprocess.run([sys.executable, "-c", _COMPILE, HIPRTC, str(p), str(co), "-DSPCIES_AR_PD=3", "-DSPCIES_AR_KREG=0", "-mllvm",
                        "-pragma-unroll-threshold=1000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    notes = subprocess.run([READELF, "--notes", str(co)], capture_output=True, text=True).stdout
    lds = [int(x) for x in re.findall(r"\.group_segment_fixed_size:\s+(\d+)", notes)]
    assert len(lds) == 4 and max(lds) <= 160 * 1024


@pytest.mark.parametrize("inst", ["tvr::admm_tvl_kernel<20, 2, 20, true, false>", "tvr::admm_tvl_kernel<18, 3, 7, false, true>", "tvr::fista_tvl_kernel<20, 2, 20, true, false>",
                                  "tvr::tv_update_coop_kernel<20, 2, true, false>", "tvr::tv_update_coop_kernel<6, 2, false, true>", "tvr::tv_bi_rolled_kernel<20>"])
def test_time_varying_lds_form_compiles_within_the_lds_and_without_scratch(inst, tmp_path):
    """admm_tvl_kernel.inc (time-varying ADMM / FISTA for plants past the register file: the instance's factors - the packed triangles of S_l - in the
    LDS) is always run-time specialised, as the concatenation admm_tvr.hip hands to hiprtc: compiled here out of process with the library's options.
    The 20-state plant of BASELINE configs[3] at N = 20 must leave room for FOUR instances in a CU's 160 KB of LDS (one per SIMD), and the ITERATION -
    everything between the first and the last MFMA - must not touch scratch memory (the first version hoisted its LDS reads into registers and spilled
    1 KB per lane, a later one parked 280 addresses there; the set-up may park what it likes: a few hundred bytes, once per solve).  The cooperative
    update phase: four workgroups per CU, no scratch at all."""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spcies_amd", "csrc")
    src = "\n".join(open(os.path.join(here, f)).read() for f in ("tv_update_kernel.inc", "admm_tvr_kernel.inc", "admm_tvl_kernel.inc"))
    src += f"\nnamespace spcies {{ __device__ void *spcies_keep_ = (void *)&{inst}; }}\n"
    p = tmp_path / "tvl.hip"
    p.write_text(src)
    import sys
    co = tmp_path / "tvl.co"
    r = subprocess.run([sys.executable, "-c", _COMPILE, HIPRTC, str(p), str(co), "-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-amdgpu-mfma-vgpr-form"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    notes = subprocess.run([READELF, "--notes", str(co)], capture_output=True, text=True).stdout
    lds = [int(x) for x in re.findall(r"\.group_segment_fixed_size:\s+(\d+)", notes)]
    scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", notes)]
    assert len(lds) == 1 and 4 * lds[0] <= 160 * 1024 and scratch[0] <= 512, (lds, scratch)
    dis = subprocess.run([os.path.join(os.path.dirname(READELF), "llvm-objdump"), "-d", str(co)], capture_output=True, text=True).stdout.split("\n")
    mfma = [i for i, line in enumerate(dis) if "v_mfma" in line]
    if mfma:  # (the solve kernels)
        loop = dis[mfma[0]:mfma[-1] + 1]
        assert sum("scratch_" in line for line in loop) <= 2, sum("scratch_" in line for line in loop)
    else:
        assert scratch[0] == 0, scratch


@pytest.mark.parametrize("family,inst", [("admm", "admm_stream_kernel<7, 3, true, true>"), ("fista", "fista_stream_kernel<7, 3, false, true>"),
                                         ("eadmm", "eadmm_stream_kernel<7, 3, true>"), ("tv", "admm_tv_update_kernel<7, 3, true, true>"),
                                         # round 5: the other switches of the template at any plant size (ensure_stream_rtc) - time-varying
                                         # (with the ROLLED update phase past n = 16: the register form does not compile at n = 20), GEN, ELLIP
                                         ("tv", "admm_tv_update_kernel<20, 4, true>"), ("fista", "fista_tv_update_kernel<17, 3, false>"),
                                         ("admm", "admm_stream_kernel<9, 2, true, true, true>"), ("fista", "fista_stream_kernel<7, 3, false, true, true>"),
                                         ("admm", "admm_stream_kernel<7, 3, false, true, false, false, true>"),
                                         ("admm", "admm_stream_kernel<7, 3, true, true, false, true, true>")])
def test_run_time_specialised_stream_sources_compile(family, inst, tmp_path):
    """The kernel texts the library hands to hiprtc for plant sizes without a build-time instantiation (spcies_hip.hip ensure_stream_rtc,
    admm_tvr.hip) are the concatenation of .inc files that are also #included at build time: here the concatenation itself is compiled,
    out of process, for one odd plant size per family - a text that only works behind the headers of the build would show."""
    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spcies_amd", "csrc")
    parts = ["admm_dev.inc", "tv_update_kernel.inc", "admm_stream_kernel.inc"]
    if family == "fista":
        parts.append("fista_stream_kernel.inc")
    if family == "eadmm":
        parts.append("eadmm_stream_kernel.inc")
    src = "\n".join(open(os.path.join(here, f)).read() for f in parts)
    # the signature is taken from the template itself: an explicit instantiation through a function pointer of the deduced type
    src += f"\nnamespace spcies {{ __device__ void *spcies_keep_ = (void *)&{inst}; }}\n"
    p = tmp_path / f"stream_{family}.hip"  # (tmp_path is per test case)
    p.write_text(src)
    import sys
    co = tmp_path / f"stream_{family}.co"
    r = subprocess.run([sys.executable, "-c", _COMPILE, HIPRTC, str(p), str(co)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    notes = subprocess.run([READELF, "--notes", str(co)], capture_output=True, text=True).stdout
    assert inst.split("<")[0] in notes
