"""Idle gaps of the GPU timeline from a rocprofv3 --kernel-trace CSV: for each dispatch the time between the end of the latest earlier dispatch and
its own start; summed per (previous kernel -> kernel).  Usage: python tools/exp/gap_report.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys, collections


def short(n):
    for p in ("void ", "(anonymous namespace)::"):
        n = n.replace(p, "")
    return n.split("(")[0][:70]


def main():
    d = sys.argv[1]
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows))
    # keep the last third of the trace (steady state: the timed passes)
    t_lo = ev[0][0] + (ev[-1][1] - ev[0][0]) * 2 // 3
    ev = [e for e in ev if e[0] >= t_lo]
    busy_end, gaps, tot_gap, tot_busy = ev[0][1], collections.defaultdict(lambda: [0, 0.0]), 0.0, 0.0
    prev = ev[0][2]
    big = []
    for s, e, n in ev[1:]:
        g = max(0, s - busy_end)
        gaps[(prev, n)][0] += 1
        gaps[(prev, n)][1] += g / 1e3
        tot_gap += g / 1e3
        if g > 200000:
            big.append((g / 1e3, prev, n))
        if e > busy_end:
            tot_busy += (e - max(s, busy_end)) / 1e3
            busy_end, prev = e, n
    span = (ev[-1][1] - ev[0][0]) / 1e3
    print("window %.1f ms: busy %.1f ms, idle %.1f ms over %d dispatches (%.1f us idle per dispatch)" % (span / 1e3, tot_busy / 1e3, tot_gap / 1e3, len(ev), tot_gap / len(ev)))
    for (p, n), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %9.1f us total %5d x %7.1f us   %s -> %s" % (g, c, g / c, p, n))
    # per kernel: the gap in front of it and its own duration
    per = collections.defaultdict(lambda: [0, 0.0, 0.0])
    be = ev[0][1]
    for s_, e_, n_ in ev[1:]:
        per[n_][0] += 1
        per[n_][1] += max(0, s_ - be) / 1e3
        per[n_][2] += (e_ - s_) / 1e3
        be = max(be, e_)
    print("per kernel: count, mean gap in front, mean duration")
    for n_, (c, g, dur) in sorted(per.items(), key=lambda kv: -kv[1][1])[:30]:
        print("  %5d x gap %8.1f us  dur %8.1f us  %s" % (c, g / c, dur / c, n_))
    hist = collections.Counter()
    be = ev[0][1]
    for s_, e_, n_ in ev[1:]:
        g = max(0, s_ - be) / 1e3
        hist[min(int(g // 5) * 5, 100)] += 1
        be = max(be, e_)
    print("gap histogram (us):", sorted(hist.items()))
    print("gaps > 200 us:")
    for g, p, n in sorted(big, reverse=True)[:20]:
        print("  %9.1f us  %s -> %s" % (g, p, n))


main()
