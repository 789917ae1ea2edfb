#!/usr/bin/env python3
"""Collects the regions marked `// [rtc-begin]` ... `// [rtc-end]` of the MFMA4 headers into one raw string
literal (mfma4_rtc_src.inc): the source mfma4_rtc.hpp hands to hiprtc when a controller's shape is not among
the kernels instantiated at build time.  usage: gen_rtc_src.py admm_mfma.hpp admm_mfma4.hpp > mfma4_rtc_src.inc"""
import sys

out = ["// (hiprtc pre-includes its own HIP device declarations: no #include)", "namespace spcies {"]
for path in sys.argv[1:]:
    keep = False
    for line in open(path):
        if line.startswith("// [rtc-begin]"):
            keep = True
            continue
        if line.startswith("// [rtc-end]"):
            keep = False
            continue
        if keep:
            out.append(line.rstrip("\n"))
out.append("}  // namespace spcies")
src = "\n".join(out)
assert ')RTCSRC"' not in src
sys.stdout.write('R"RTCSRC(' + src + '\n)RTCSRC"\n')
