#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite) into a per-kernel table (name, calls, total,
avg, %), optionally listing the N longest dispatches.  Usage: rocprof_summary.py results.db [--top N] [--frames F]"""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace("cobevt::", "")
    n = re.sub(r"void ", "", n)
    return n[:110]


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0
    frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 1
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, "
                            "accum_vgpr_count, scratch_size from kernels order by start"))
    agg = collections.OrderedDict()
    for n, s, e, gx, gy, gz, wx, lds, vg, ag, sc in rows:
        a = agg.setdefault(short(n), [0, 0.0, vg, ag, lds, sc])
        a[0] += 1
        a[1] += (e - s) / 1e3
    tot = sum(a[1] for a in agg.values())
    print("%d dispatches, %.1f us of kernel time (%.1f us per frame over %d frames)" % (len(rows), tot, tot / frames, frames))
    print("%10s %6s %6s %9s  %4s %4s %6s %5s  %s" % ("total_us", "calls", "%", "avg_us", "vgpr", "agpr", "lds", "scr", "kernel"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%10.1f %6d %6.2f %9.2f  %4d %4d %6d %5d  %s" % (a[1], a[0], 100 * a[1] / tot, a[1] / a[0], a[2], a[3], a[4], a[5], k))
    if top:
        print("\nlongest %d dispatches:" % top)
        for n, s, e, gx, gy, gz, wx, lds, vg, ag, sc in sorted(rows, key=lambda r: r[1] - r[2])[:top]:
            print("%9.1f us  grid %7d x%d x%d  wg %d  %s" % ((e - s) / 1e3, gx, gy, gz, wx, short(n)))


if __name__ == "__main__":
    main()
