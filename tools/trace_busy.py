"""Where the device's time goes in a window of a rocprofv3 kernel trace: per kernel the summed durations, and how much of the window had
0, 1, 2, 3+ kernels running.  Usage: trace_busy.py <kernel_trace.csv> [frac_lo frac_hi] (window as fractions of the trace, default 0.5 0.9)
or trace_busy.py <kernel_trace.csv> @<kernel substring> <n>: the window from the start of the n-th last launch of that kernel to the start of its
last launch (n whole batches of a steady loop)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
T0, T1 = ev[0][0], max(e for _, e, _ in ev)
if len(sys.argv) > 2 and sys.argv[2].startswith('@'):
    anchors = [s for s, _, n in ev if sys.argv[2][1:] in n]
    nb = int(sys.argv[3])
    lo, hi = anchors[-1 - nb], anchors[-1]
    print("%d launches of %s: %.3f ms each" % (nb, sys.argv[2][1:], (hi - lo) / nb / 1e6))
else:
    flo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    fhi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
    lo, hi = T0 + flo * (T1 - T0), T0 + fhi * (T1 - T0)
def short(n):
    n = n.replace('void ', '').replace('pbwtk::', '')
    return n.split('(')[0][:44]
per = collections.Counter(); cnt = collections.Counter()
pts = []
for s, e, n in ev:
    s2, e2 = max(s, lo), min(e, hi)
    if e2 <= s2: continue
    per[short(n)] += e2 - s2; cnt[short(n)] += 1
    pts.append((s2, 1)); pts.append((e2, -1))
pts.sort()
depth = 0; last = lo; hist = collections.Counter()
for t, d in pts:
    hist[min(depth, 3)] += t - last; last = t; depth += d
hist[0] += hi - last
W = hi - lo
print("window %.1f ms; kernels running: none %.1f%%  one %.1f%%  two %.1f%%  three+ %.1f%%" % (W / 1e6, *(100 * hist[i] / W for i in range(4))))
print("summed kernel durations %.1f ms = %.2fx the window" % (sum(per.values()) / 1e6, sum(per.values()) / W))
for n, v in per.most_common(16):
    print("  %-46s %8.2f ms  %5.1f%% of window  n=%d  avg %.1f us" % (n, v / 1e6, 100 * v / W, cnt[n], v / cnt[n] / 1e3))
