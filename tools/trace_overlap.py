"""Chain-kernel durations by what runs beside them, from a rocprofv3 kernel trace CSV.  Usage: trace_overlap.py <kernel_trace.csv>"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
KN = ('skel_fill', 'skel_k2_wide', 'skel_k2', 'skel_rank', 'skel_hist', 'sweep_hist', 'p3r_scan', 'p3r_combine', 'p3r_emit', 'transpose32')
def nm(r):
    for k in KN:
        if k in r['Kernel_Name']: return k
    return 'other'
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), nm(r)) for r in rows]
ev.sort()
cons = [(s, e, n) for s, e, n in ev if n in ('skel_fill', 'sweep_hist', 'p3r_scan', 'p3r_emit')]
stats = collections.defaultdict(list)
for s, e, n in ev:
    if n not in ('skel_hist', 'skel_k2_wide', 'skel_k2', 'skel_rank'): continue
    beside = 'alone'
    for cs, ce, cn in cons:
        if cs <= s < ce: beside = cn; break
    stats[(beside, n)].append((e - s) / 1000.0)
for (b, n), v in sorted(stats.items()):
    v.sort()
    print('beside %-11s %-13s n=%5d  mean %8.1f us  median %8.1f  max %8.1f' % (b, n, len(v), sum(v) / len(v), v[len(v) // 2], v[-1]))
for n in ('skel_fill', 'sweep_hist'):
    v = [(e - s) / 1000.0 for s, e, x in ev if x == n]
    if v: print('%-11s n=%d mean %.1f us' % (n, len(v), sum(v) / len(v)))
