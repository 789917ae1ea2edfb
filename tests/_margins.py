"""Worst observed differences of the parity helpers, per test: what the relaxed bars of test_parity_gpu.py really lend.

Every non-bit-exact comparison calls `record(...)` with the worst ABSOLUTE differences it saw and the worst ratio to its bar
(`frac_of_bar` <= 1 is what the assert next to it checks).  `conftest.pytest_sessionfinish` writes the collection to
`gpurun_out/parity_margins.json` (merged back from the GPU box by gpurun); `tools/summarise_margins.py` turns it into the
committed `profiles/rNN_parity_margins.txt`."""
import json
import os

_ROWS = {}


def _test_id():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]


def record(helper, variant, **values):
    """Keep the worst value per (test, helper, variant, key)."""
    key = (_test_id(), helper, str(variant))
    row = _ROWS.setdefault(key, {})
    for name, val in values.items():
        val = float(val)
        if name not in row or val > row[name]:
            row[name] = val


def dump(root):
    if not _ROWS:
        return None
    out = os.path.join(root, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_margins.json")
    rows = [dict(test=t, helper=h, variant=v, **vals) for (t, h, v), vals in sorted(_ROWS.items())]
    with open(path, "w") as f:
        json.dump(rows, f, indent=0)
    return path
