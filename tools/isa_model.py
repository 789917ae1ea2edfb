#!/usr/bin/env python3
"""Static issue model of a matrix-pipe kernel's steady-state loop, from `hipcc -S` output (no GPU needed).

usage: tools/isa_model.py file.s [--kernel REGEX] [--loop-depth D] [--skip-cold]

Rules measured on MI355X (profiles/r03_microbench_issue.txt, one wavefront per SIMD): v_mfma_f64_4x4x4 16 ticks; a vector-ALU
instruction of any kind 4 ticks in a run, the first one behind an MFMA +8 (pipe switch); v_accvgpr_read 8; s_nop / s_waitcnt /
scalar instructions and the first LDS read behind an MFMA hide in its other issue slots.  The script finds the innermost loop
of the kernel with the most MFMAs, drops the basic blocks that only run while the residual checks are live (fall-through
blocks `; %bb.N` behind a conditional branch, except the first one) when --skip-cold is given, and prints the histogram and
the modelled ticks per trip.
"""
import re, sys, argparse, collections

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("--kernel", default=None)
    ap.add_argument("--skip-cold", action="store_true")
    ap.add_argument("--range", default=None, help="first:last line of the region (1-based) instead of the automatic loop search")
    ap.add_argument("--dump-runs", action="store_true")
    a = ap.parse_args()
    lines = open(a.file).read().split("\n")
    # kernel extent
    start, end = 0, len(lines)
    if a.kernel:
        rx = re.compile(a.kernel)
        for i, l in enumerate(lines):
            if re.match(r"^_Z\w+:", l) and rx.search(l):
                start = i
                break
        for i in range(start, len(lines)):
            if "s_endpgm" in lines[i]:
                end = i
                break
    if a.range:
        lo, hi = (int(x) for x in a.range.split(":"))
    else:
        # the deepest loop: lines whose block comments carry the largest Depth=
        depth = collections.Counter()
        for i in range(start, end):
            m = re.search(r"Depth=(\d+)", lines[i])
            if m and ("in Loop" in lines[i] or "Loop Header" in lines[i]):
                depth[int(m.group(1))] += 1
        D = max(depth) if depth else 1
        idx = [i for i in range(start, end) if re.search(r"Depth=%d\b" % D, lines[i])]
        lo, hi = idx[0] + 1, idx[-1] + 1
        # extend to the backward branch that closes the loop
        for i in range(hi, end):
            if re.search(r"s_cbranch|s_branch", lines[i]):
                hi = i + 1
                break
    body = lines[lo - 1:hi]
    hot = []
    cold = False
    seen_first_cold = False
    for l in body:
        s = l.strip()
        if re.match(r"^(\.LBB\d+_\d+:)", s):
            cold = False
        elif s.startswith("; %bb.") and a.skip_cold:
            if seen_first_cold:
                cold = True
            seen_first_cold = True
        if not s or s.startswith(";") or s.startswith(".") :
            continue
        if not cold:
            hot.append(s.split(";")[0].strip())
    hist = collections.Counter()
    ticks = 0
    runs = []
    run = 0
    prev_mfma = False
    for ins in hot:
        op = ins.split()[0]
        if op.startswith("v_mfma"):
            cls = "mfma"; ticks += 16
            if run: runs.append(run); run = 0
            prev_mfma = True
        elif op.startswith("v_accvgpr_read"):
            cls = "acc_read"; ticks += 8 + (8 if prev_mfma else 0); run += 1; prev_mfma = False
        elif op.startswith("v_accvgpr_write"):
            cls = "acc_write"; ticks += 4 + (8 if prev_mfma else 0); run += 1; prev_mfma = False
        elif op.startswith("v_"):
            f64 = "f64" in op
            cls = "valu_f64" if f64 else "valu_other"
            if "dpp" in ins: cls = "valu_dpp"
            ticks += 4 + (8 if prev_mfma else 0); run += 1; prev_mfma = False
        elif op.startswith("ds_"):
            cls = "lds"
        elif op.startswith("s_nop"):
            cls = "s_nop"
        elif op.startswith("s_waitcnt"):
            cls = "s_waitcnt"
        elif op.startswith("s_barrier"):
            cls = "s_barrier"
        elif op.startswith("s_"):
            cls = "salu"
        elif op.startswith(("buffer_", "global_", "scratch_", "flat_")):
            cls = "vmem:" + op.split("_")[0] + ("_st" if "store" in op else "_ld")
        else:
            cls = "other:" + op
        hist[cls] += 1
        hist["op:" + op] += 0
    if run: runs.append(run)
    print(f"region lines {lo}..{hi}, hot instructions {len(hot)}")
    for k in sorted(k for k in hist if not k.startswith("op:")):
        print(f"  {k:14s} {hist[k]}")
    print(f"  runs of vector instructions: {len(runs)}  (1-2 long: {sum(1 for r in runs if r <= 2)})")
    print(f"  modelled ticks per trip: {ticks}  (MFMA alone {16 * hist['mfma']})")
    ops = collections.Counter(i.split()[0] for i in hot if i.split()[0].startswith("v_") and not i.startswith("v_mfma"))
    print("  vector ops:", ", ".join(f"{k} {v}" for k, v in ops.most_common(24)))
    if a.dump_runs:
        print("  run lengths:", runs)

if __name__ == "__main__":
    main()
