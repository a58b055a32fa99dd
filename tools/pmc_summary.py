#!/usr/bin/env python
"""Per-kernel averages of rocprofv3 --pmc counters from one or more results .db files (per wave where sensible)."""
import collections
import re
import sqlite3
import sys

res = {}
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    for did, name, grid, wg, cname, val in cur.execute(
            "select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value from counters_collection"):
        key = (re.sub(r"cobevt::|void ", "", name)[:52], grid)
        e = res.setdefault(key, collections.defaultdict(float))
        e[cname] += val
        e["_n_" + cname] += 1
        e["_waves"] = grid / 64.0
for key, e in res.items():
    waves = e["_waves"]
    print("%s grid=%d waves=%d" % (key[0], key[1], waves))
    out = []
    for c in sorted(k for k in e if not k.startswith("_")):
        per_dispatch = e[c] / max(e["_n_" + c], 1) * (1 if True else 1)
        out.append("%s/wave=%.4g" % (c.replace("SQ_", ""), per_dispatch / waves))
    print("    " + "  ".join(out))
