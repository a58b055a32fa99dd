#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace results .db (rocpd sqlite) into a per-kernel table (name, calls, total,
avg, %), optionally listing the N longest dispatches, a per-(kernel, grid) table (--by-grid) and the dispatch timeline
of the last graph replay (--timeline N: the last N dispatches with start offsets, gaps and overlap).
Usage: rocprof_summary.py results.db [--top N] [--frames F] [--by-grid] [--timeline N [--densest]] [--busy MS]"""
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
    if "--steady" in sys.argv:
        # only the steady state: everything from the start of the N-th last frame on (a frame starts with its stem launch).  A
        # whole-trace table also counts model set-up (one small copy per uploaded weight plan, eager warm-up frames), which
        # reads like "copies per frame" when divided by the frame count
        nlast = int(sys.argv[sys.argv.index("--steady") + 1])
        # --skip-last M: leave out the M frames at the end of the trace (bench.py appends the one-frame-at-a-time leg, 55 frames,
        # behind the pipelined timed loop)
        skip = int(sys.argv[sys.argv.index("--skip-last") + 1]) if "--skip-last" in sys.argv else 0
        stems = [r[1] for r in rows if short(r[0]).startswith("stem_pool_kernel") or short(r[0]).startswith("stem7x7_kernel")]
        if len(stems) > nlast + skip:
            t0 = stems[-(nlast + skip)]
            t1 = stems[-skip] if skip else None
            rows = [r for r in rows if r[1] >= t0 and (t1 is None or r[1] < t1)]
            print("steady state only: %d frames of the trace%s" % (nlast, ", ending %d frames before its end" % skip if skip else ""))
    agg = collections.OrderedDict()
    for n, s, e, gx, gy, gz, wx, lds, vg, ag, sc in rows:
        a = agg.setdefault(short(n), [0, 0.0, vg, ag, lds, sc])
        a[0] += 1
        a[1] += (e - s) / 1e3
    tot = sum(a[1] for a in agg.values())
    # the per-frame normaliser: a CoBEVT frame launches the stem (+ pool) kernel exactly once, so its dispatch count IS the
    # number of traced frames; --frames only overrides that when the trace has no such kernel (operator-level workloads).  A
    # disagreement between the two is reported instead of silently dividing by the wrong number
    once = [a[0] for k, a in agg.items() if k.startswith("stem_pool_kernel") or k.startswith("stem7x7_kernel")]
    if once:
        counted = sum(once)
        if "--frames" in sys.argv and frames != counted:
            print("note: --frames %d ignored, the trace holds %d frames (stem launches)" % (frames, counted))
        frames = counted
    print("%d dispatches, %.1f us of kernel time (%.1f us per frame over %d frames)" % (len(rows), tot, tot / frames, frames))
    print("%10s %6s %6s %9s  %4s %4s %6s %5s  %s" % ("total_us", "calls", "%", "avg_us", "vgpr", "agpr", "lds", "scr", "kernel"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%10.1f %6d %6.2f %9.2f  %4d %4d %6d %5d  %s" % (a[1], a[0], 100 * a[1] / tot, a[1] / a[0], a[2], a[3], a[4], a[5], k))
    if "--by-grid" in sys.argv:
        g = collections.OrderedDict()
        for n, s, e, gx, gy, gz, wx, lds, vg, ag, sc in rows:
            a = g.setdefault((short(n)[:60], gx * gy * gz // max(wx, 1)), [0, 0.0])
            a[0] += 1
            a[1] += (e - s) / 1e3
        print("\nper (kernel, workgroups), us per frame:")
        for k, a in sorted(g.items(), key=lambda kv: -kv[1][1]):
            print("%9.1f us/frame %6.2f calls/frame %8.2f us avg  %6d wgs  %s" % (a[1] / frames, a[0] / frames, a[1] / a[0], k[1], k[0]))
    if "--timeline" in sys.argv:
        nlast = int(sys.argv[sys.argv.index("--timeline") + 1])
        last = rows[-nlast:]
        if "--densest" in sys.argv:          # the N-dispatch window with the smallest span (= a HIP-graph replay)
            best = min(range(0, len(rows) - nlast), key=lambda i: rows[i + nlast - 1][2] - rows[i][1])
            last = rows[best:best + nlast]
        t0 = last[0][1]
        busy_end = t0
        idle = 0.0
        print("\ntimeline of the last %d dispatches (start us, dur us, gap since all earlier kernels ended):" % nlast)
        for n, s, e, gx, gy, gz, wx, lds, vg, ag, sc in last:
            gap = (s - busy_end) / 1e3
            if gap > 0:
                idle += gap
            print("%9.1f %7.1f %6.1f  %6d wgs  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, gx * gy * gz // max(wx, 1), short(n)[:70]))
            busy_end = max(busy_end, e)
        print("span %.1f us, idle (no kernel running) %.1f us" % ((busy_end - t0) / 1e3, idle))
    if "--busy" in sys.argv:
        # steady-state occupancy of the device over the last MS milliseconds of the trace: how much of the wall time has at
        # least one kernel running, the average number of kernels in flight, and the CU-weighted occupancy (a dispatch
        # with W workgroups is taken to occupy min(W, 256) / 256 of the chip for its duration - an upper bound)
        ms = float(sys.argv[sys.argv.index("--busy") + 1])
        skip = float(sys.argv[sys.argv.index("--busy-skip") + 1]) if "--busy-skip" in sys.argv else 0.0   # ms before the end
        t_end = rows[-1][2] - skip * 1e6
        win = [r for r in rows if t_end - ms * 1e6 <= r[1] and r[2] <= t_end]
        t_end = win[-1][2]
        t0 = win[0][1]
        span = (t_end - t0) / 1e3
        busy_end, union, gaps = t0, 0.0, []
        for n, s, e, gx, gy, gz, wx, lds, vg, ag, sc in win:
            if s > busy_end:
                gaps.append((s - busy_end) / 1e3)
                union += 0.0
                busy_end_prev = busy_end
            union += max(0.0, (e - max(s, busy_end)) / 1e3)
            busy_end = max(busy_end, e)
        total = sum((r[2] - r[1]) / 1e3 for r in win)
        cu = sum((r[2] - r[1]) / 1e3 * min(r[3] * r[4] * r[5] // max(r[6], 1), 256) / 256.0 for r in win)
        print("\nlast %.1f ms: %d dispatches over %.1f us; a kernel is running %.1f %% of the time (%d idle gaps, %.1f us in total, "
              "longest %.1f us); kernels in flight on average %.2f; CU-weighted occupancy <= %.1f %%" %
              (ms, len(win), span, 100.0 * union / span, len(gaps), sum(gaps), max(gaps) if gaps else 0.0, total / span, 100.0 * cu / span))
        big = sorted(gaps, reverse=True)[:10]
        print("largest idle gaps (us): %s" % ", ".join("%.1f" % g for g in big))
    if top:
        print("\nlongest %d dispatches:" % top)
        for n, s, e, gx, gy, gz, wx, lds, vg, ag, sc in sorted(rows, key=lambda r: r[1] - r[2])[:top]:
            print("%9.1f us  grid %7d x%d x%d  wg %d  %s" % ((e - s) / 1e3, gx, gy, gz, wx, short(n)))


if __name__ == "__main__":
    main()
