"""Gaps of the dependent chain by what runs beside them, from a rocprofv3 kernel trace CSV: for every chain kernel the time between the end of the
previous chain kernel and its own start ("gap": boundary + admission), and its duration, grouped by the consumer kernel running at its start.
Usage: trace_gaps.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
CH = ('skel_hist', 'skel_k2_wide', 'skel_k2', 'skel_rank')
CO = ('skel_fillseq', 'skel_fill', 'sweep_hist', 'p3r_scan', 'p3r_combine', 'p3r_emit', 'transpose32', 'fillBuffer', 'copyBuffer')
def nm(r, names):
    for k in names:
        if k in r['Kernel_Name']: return k
    return None
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r) for r in rows)
cons = [(s, e, nm(r, CO)) for s, e, r in ev if nm(r, CO)]
chain = [(s, e, nm(r, CH)) for s, e, r in ev if nm(r, CH)]
st = collections.defaultdict(lambda: [[], []])
prev_end = None
for s, e, n in chain:
    beside = 'alone'
    for cs, ce, cn in cons:
        if cs <= s < ce: beside = cn; break
    if prev_end is not None and s - prev_end < 200000:      # same batch (a batch boundary is a longer pause)
        st[(beside, n)][0].append((s - prev_end) / 1000.0)
    st[(beside, n)][1].append((e - s) / 1000.0)
    prev_end = e
print("%-14s %-13s %6s %10s %10s %10s %10s" % ("beside", "chain kernel", "n", "gap mean", "gap median", "dur mean", "dur median"))
for (b, n), (g, d) in sorted(st.items()):
    g.sort(); d.sort()
    if not g: continue
    print("%-14s %-13s %6d %10.2f %10.2f %10.2f %10.2f" % (b, n, len(d), sum(g) / len(g), g[len(g) // 2], sum(d) / len(d), d[len(d) // 2]))
tot = chain[-1][1] - chain[0][0]
busy = sum(e - s for s, e, n in chain)
print("chain span %.1f ms, chain kernels running %.1f ms (%.0f %%), %d launches: %.2f us per launch span, %.2f us kernel time" % (tot / 1e6, busy / 1e6, 100.0 * busy / tot, len(chain), tot / 1e3 / len(chain), busy / 1e3 / len(chain)))
