#!/usr/bin/env python3
"""gpurun_out/parity_margins.json (written by `pytest tests -m gpu`, tests/_margins.py) -> a table for profiles/rNN_parity_margins.txt:
per test and variant the worst absolute differences against the oracle and the share of the bar they use."""
import json
import sys

rows = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/parity_margins.json"))
cols = ("du", "dz", "dv", "dlam", "lam_scale", "k_differs", "bar", "frac_of_bar", "frac_of_flat_bar")
print(f"# {len(rows)} comparisons; frac_of_bar = worst |difference| / the bar the test asserts (scaled bars included), "
      f"frac_of_flat_bar = worst |d(u, z, v)| / 1e-10 with no scaling")
print("test | helper | variant | " + " | ".join(cols))
worst = {}
for r in rows:
    print(" | ".join([r["test"].replace("tests/", ""), r["helper"], r["variant"]] +
                     [("%.2e" % r[c]) if c in r else "-" for c in cols]))
    for c in ("frac_of_bar", "frac_of_flat_bar"):
        if c in r and r[c] > worst.get(c, (0, ""))[0]:
            worst[c] = (r[c], r["test"])
for c, (val, test) in worst.items():
    print(f"# worst {c}: {val:.3f} in {test}")
