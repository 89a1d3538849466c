"""Coarse timeline of a rocprofv3 kernel trace: consumer kernels and runs of chain kernels.  Usage: trace_timeline.py <kernel_trace.csv> [t0_ms t1_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
KN = ('skel_fill', 'skel_k2_wide', 'skel_k2', 'skel_rank', 'skel_hist', 'sweep_hist', 'p3r_scan', 'p3r_combine', 'p3r_emit', 'transpose32', 'synth', 'skel_keys', 'hist_fold', 'bump', 'scan_u64', 'pack3_offsets', 'fillBuffer', 'copyBuffer')
def nm(r):
    for k in KN:
        if k in r['Kernel_Name']: return k
    return r['Kernel_Name'][:24]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), nm(r), r.get('Queue_Id', '?')) for r in rows)
t0 = ev[0][0]
lo = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0
hi = float(sys.argv[3]) * 1e6 if len(sys.argv) > 3 else 1e18
chain = ('skel_hist', 'skel_k2_wide', 'skel_k2', 'skel_rank')
out = []
run = None
for s, e, n, q in ev:
    if n in chain:
        if run and s - run[1] < 20000: run[1] = e; run[2] += 1
        else:
            if run: out.append((run[0], run[1], 'CHAIN x%d' % run[2], '1'))
            run = [s, e, 1]
    else:
        out.append((s, e, n, q))
if run: out.append((run[0], run[1], 'CHAIN x%d' % run[2], '1'))
for s, e, n, q in sorted(out):
    if lo <= s - t0 <= hi and (e - s > 20000 or n.startswith('CHAIN')):
        print('%10.3f ms  +%9.3f ms  q%s %s' % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
