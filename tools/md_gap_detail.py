"""the kernels around ONE long hole of the panel-side chain of -matchDynamic (rocprofv3 kernel trace): usage md_gap_detail.py <kernel_trace.csv> [which]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
qk = "Queue_Id" if "Queue_Id" in rows[0] else None
sk = "Stream_Id" if "Stream_Id" in rows[0] else None
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get(qk, "?") if qk else "?", r.get(sk, "?") if sk else "?") for r in rows)
chain = [e for e in ev if any(t in e[2] for t in ("skel_hist_kernel<4, true>", "skel_k2_wide_kernel", "skel_rank_kernel<2, 0, true>"))]
gaps = sorted(((chain[i + 1][0] - chain[i][1], chain[i][1], chain[i + 1][0]) for i in range(len(chain) - 1)), reverse=True)
g, a, b = gaps[min(which, len(gaps) - 1)]
print("hole of %.1f us; kernels overlapping [hole start - 300 us, hole end + 100 us], times relative to the hole's start (us):" % (g / 1e3))
def short(n): return n.replace("void ", "").replace("pbwtk::", "").split("(")[0][:46]
for s, e, n, q, st in ev:
    if e > a - 300000 and s < b + 100000 and not any(t in n for t in ("skel_hist_kernel<4, true>", "skel_k2_wide_kernel", "skel_rank_kernel<2, 0, true>")) or (abs(s - b) < 1000 or abs(e - a) < 1000):
        print("  q%-3s s%-3s %9.1f -> %9.1f  (%8.1f us)  %s" % (q, st, (s - a) / 1e3, (e - a) / 1e3, (e - s) / 1e3, short(n)))
