"""CPU: compile instantiations of the time-varying LDS-form module (tv_update_kernel.inc + admm_tvr_kernel.inc + admm_tvl_kernel.inc, the text
admm_tvr.hip hands to hiprtc) out of process with the library's options and print what the code object says about each kernel: registers, LDS,
scratch (and, with --isa DIR, the disassembly).  No GPU needed.
    python tools/tvl_compile.py "tvr::admm_tvl_kernel<20, 2, 20, true, false>" "tvr::fista_tvl_kernel<20, 2, 20, true, false>" [-D...] [--isa /tmp/x]"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spcies_amd", "csrc")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
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
    sys.exit(log.value.decode()[:4000])
rt.hiprtcGetCodeSize(prog, C.byref(n))
code = C.create_string_buffer(n.value)
rt.hiprtcGetCode(prog, code)
open(sys.argv[3], "wb").write(code.raw)
"""


def main():
    args = sys.argv[1:]
    isa = None
    if "--isa" in args:
        i = args.index("--isa")
        isa = args[i + 1]
        del args[i:i + 2]
    inc = os.path.join(HERE, "admm_tvl_kernel.inc")
    if "--inc" in args:  # a working copy of admm_tvl_kernel.inc
        i = args.index("--inc")
        inc = args[i + 1]
        del args[i:i + 2]
    vgpr_form = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
    if "--no-vgpr-form" in args:  # (experiment: MFMA results in the accumulation registers)
        args.remove("--no-vgpr-form")
        vgpr_form = []
    insts = [a for a in args if not a.startswith("-")]
    flags = [a for a in args if a.startswith("-")]
    src = "\n".join(open(f).read() for f in (os.path.join(HERE, "tv_update_kernel.inc"), os.path.join(HERE, "admm_tvr_kernel.inc"), inc))
    src += "\nnamespace spcies { __device__ void *spcies_keep_[] = {" + ", ".join(f"(void *)&{i}" for i in insts) + "}; }\n"
    d = isa or tempfile.mkdtemp()
    os.makedirs(d, exist_ok=True)
    p, co = os.path.join(d, "tvl.hip"), os.path.join(d, "tvl.co")
    open(p, "w").write(src)
    r = subprocess.run([sys.executable, "-c", _COMPILE, HIPRTC, p, co, "-mllvm", "-pragma-unroll-threshold=1000000", *vgpr_form, *flags],
                       capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        print(f"{g('name')[:90]}: vgpr {g('vgpr_count')} agpr {blk.split()[0].strip(':')} sgpr {g('sgpr_count')} lds {g('group_segment_fixed_size')} scratch {g('private_segment_fixed_size')}")
    if isa:
        open(os.path.join(d, "tvl.s"), "w").write(subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True).stdout)
        print("ISA:", os.path.join(d, "tvl.s"))


if __name__ == "__main__":
    main()
