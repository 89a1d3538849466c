"""every kernel of a rocprofv3 kernel trace in a window, in start order: tools/trace_dump.py <kernel_trace.csv> <from_kernel_substring> [count=120]
(the window starts at the LAST-but-3rd kernel whose name contains the substring: the steady state, not the set-up)"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '?')) for r in rows)
key = sys.argv[2]; n = int(sys.argv[3]) if len(sys.argv) > 3 else 120
idx = [i for i, e in enumerate(ev) if key in e[2]]
if not idx:
    sys.exit("no kernel matches " + key)
i0 = idx[max(0, len(idx) - 4)]
t0 = ev[i0][0]
def short(nm):
    nm = nm.replace('pbwtk::', '').replace('void ', '')
    return nm[:60]
for s, e, nm, q in ev[max(0, i0 - 6): i0 + n]:
    print('%10.1f us  +%9.1f us  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, short(nm)))
