#!/usr/bin/env python
"""HBM bytes per launch per kernel family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the eager bench.
Usage: pmc_traffic.py <fetch results .db> <write results .db> > profiles/pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are reported in kilobytes; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950
(128-byte requests tallied as 64 bytes); WRITE_SIZE is uncalibrated."""
import collections
import json
import sqlite3
import sys

FAMILIES = [("conv3x3_strips_kernel", "conv3x3"), ("conv3x3_kernel", "conv3x3"), ("basicblock_kernel", "basicblock"), ("dsblock_kernel", "basicblock"),
            ("gemm_rows_kernel", "gemm_rows"), ("row_chain_kernel", "row_chain"), ("attn_gather_kernel", "attention"),
            ("igemm_kernel", "igemm"), ("stem7x7_kernel", "stem7x7"), ("stem_pool_kernel", "stem7x7")]


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return None


def collect(path, counter):
    cur = sqlite3.connect(path).cursor()
    per = collections.defaultdict(lambda: [0.0, 0])
    for did, name, cname, val in cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
        fam = family(name)
        if fam is None or cname != counter:
            continue
        per[fam][0] += val
        per[fam][1] += 1
    return per


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python bench.py --steps 3 --warmup 1 "
                "--no-graph --no-cpu-baseline --no-roofline --frames-in-flight 1` (eager frames, bf16, 5 agents); per-launch "
                "averages per kernel family; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests as "
                "64 B); WRITE_SIZE uncalibrated; both counters are in KB",
       "bf16": {}, "detail_bf16": {}}
for fam in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(fam, [0.0, 0])
    w, nw = write.get(fam, [0.0, 0])
    fb = f * 1024.0 / max(nf, 1)
    wb = w * 1024.0 / max(nw, 1)
    out["bf16"][fam] = int(2 * fb + wb)
    out["detail_bf16"][fam] = {"fetch_bytes_per_launch_raw": int(fb), "fetch_bytes_per_launch_x2": int(2 * fb),
                               "write_bytes_per_launch": int(wb), "hbm_bytes_per_launch": int(2 * fb + wb),
                               "launches_sampled": nf}
print(json.dumps(out, indent=1))
