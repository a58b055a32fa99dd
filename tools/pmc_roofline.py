#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of tools/pmc_collect.sh per kernel family (and per attention launch shape):
HBM bytes per launch (FETCH_SIZE doubled for gfx950 as MI355X_MICROARCH.md prescribes + WRITE_SIZE, both reported in KB)
and MFMA / VALU activity (SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, summed over the chip; SQ_BUSY_CYCLES is per
shader engine, 32 of them; MfmaUtil = MFMA_BUSY / (4 SIMDs x 256 CUs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32).
Usage: pmc_roofline.py <dir with fetch/ write/ sq/ sub-directories> > profiles/pmc_rNN.json"""
import collections
import glob
import json
import os
import re
import sqlite3
import sys

FAMILIES = [("conv3x3_strips_kernel", "conv3x3"), ("conv3x3_kernel", "conv3x3"), ("basicblock_kernel", "basicblock"), ("dsblock_kernel", "basicblock"),
            ("bottleneck_kernel", "bottleneck"), ("gemm_rows3_kernel", "gemm_rows"), ("gemm_rows_kernel", "gemm_rows"),
            ("gemm_rows2_kernel", "gemm_rows"), ("bev_query_kernel", "gemm_rows"), ("row_chain64_kernel", "row_chain"),
            ("proj_chain128_kernel", "row_chain"), ("proj_chain_k_kernel", "row_chain"), ("row_chain_kernel", "row_chain"), ("swap_stage_kernel", "swap_stage"), ("attn_resident_kernel", "attention"),
            ("attn_gather_kernel", "attention"), ("igemm_kernel", "igemm"), ("stem7x7_kernel", "stem7x7"),
            ("stem_pool_kernel", "stem7x7")]
N_SE, N_SIMD = 32, 4 * 256


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return None


def rows(path):
    dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True)
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        for r in cur.execute("select dispatch_id, kernel_name, grid_size, workgroup_size, counter_name, value, duration "
                             "from counters_collection"):
            yield r


def collect(path):
    """{family or attention shape: {counter: [sum, n], '_dur': [sum ns, n]}}"""
    per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    seen = set()
    for did, name, grid, wg, cname, val, dur in rows(path):
        fam = family(name)
        if fam is None:
            continue
        keys = [fam]
        if fam == "attention":
            short = re.sub(r"^.*(attn_\w+_kernel<[^>]*>).*$", r"\1", name)
            keys.append("attention|%s|%d wgs" % (short, grid // max(wg, 1)))
        for k in keys:
            per[k][cname][0] += val
            per[k][cname][1] += 1
            if (k, did) not in seen:
                seen.add((k, did))
                per[k]["_dur"][0] += dur
                per[k]["_dur"][1] += 1
    return per


def main():
    base = sys.argv[1]
    fetch, write, sq = collect(os.path.join(base, "fetch")), collect(os.path.join(base, "write")), collect(os.path.join(base, "sq"))
    out = {"_note": "rocprofv3 --pmc passes (tools/pmc_collect.sh) over eager bf16 frames of the 5-agent bench workload; per-launch "
                    "averages.  hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; FETCH_SIZE doubled: gfx950 tallies 128-B "
                    "requests as 64 B, MI355X_MICROARCH.md; WRITE_SIZE uncalibrated).  mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / "
                    "(1024 SIMDs x SQ_BUSY_CYCLES / 32 SEs); valu_per_mfma = SQ_INSTS_VALU / SQ_INSTS_MFMA where collected.",
           "families": {}, "attention_launches": {}}
    for k in sorted(set(fetch) | set(write) | set(sq)):
        e = {}
        f, w = fetch.get(k, {}).get("FETCH_SIZE"), write.get(k, {}).get("WRITE_SIZE")
        if f and f[1]:
            e["fetch_bytes_x2"] = int(2 * f[0] * 1024.0 / f[1])
        if w and w[1]:
            e["write_bytes"] = int(w[0] * 1024.0 / w[1])
        if "fetch_bytes_x2" in e and "write_bytes" in e:
            e["hbm_bytes"] = e["fetch_bytes_x2"] + e["write_bytes"]
        s = sq.get(k, {})
        avg = {c: v[0] / v[1] for c, v in s.items() if v[1] and c != "_dur"}
        if "_dur" in s and s["_dur"][1]:
            e["avg_duration_us_profiled"] = round(s["_dur"][0] / s["_dur"][1] / 1e3, 2)
            e["launches_sampled"] = s["_dur"][1]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("SQ_BUSY_CYCLES"):
            e["mfma_util"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * avg["SQ_BUSY_CYCLES"] / N_SE), 4)
        if avg.get("SQ_INSTS_MFMA") and "SQ_INSTS_VALU" in avg:
            e["valu_per_mfma"] = round(avg["SQ_INSTS_VALU"] / avg["SQ_INSTS_MFMA"], 2)
        if avg.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_VALU" in avg:
            e["valu_active_frac_of_wave_cycles"] = round(avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"], 4)
        e["counters_per_launch"] = {c: round(v, 1) for c, v in sorted(avg.items())}
        (out["attention_launches"] if k.startswith("attention|") else out["families"])[k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
