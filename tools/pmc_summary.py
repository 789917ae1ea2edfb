"""Summarise the separate FETCH_SIZE / WRITE_SIZE rocprofv3 passes of tools/profile_bench.sh.
usage: python tools/pmc_summary.py gpurun_out/prof_<tag> > profiles/<tag>_pmc_summary.txt
`mean` excludes the first dispatch of each kernel (cold L2 / first-touch page faults), which is listed as `first`."""
import collections, csv, glob, sys

root = sys.argv[1]
print("# rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py --variant ... (separate passes; values are per "
      "dispatch, KB; mean over all dispatches but the first)")
for ctr, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(f"{root}/{sub}/*/*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != ctr:
                continue
            key = (r["Kernel_Name"][:90], r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
            agg.setdefault(key, []).append(float(r["Counter_Value"]))
    for (name, vg, ag, lds, scr), v in agg.items():
        rest = v[1:] if len(v) > 1 else v
        print(f"{ctr}  n={len(v)} mean={sum(rest) / len(rest):.6g} KB first={v[0]:.6g} KB  kernel={name}  vgpr={vg} agpr={ag} "
              f"lds={lds} scratch={scr}")
