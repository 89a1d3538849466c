"""Summarise a rocprofv3 kernel trace (CSV) of bench.py as a per-batch timeline: when the chain of each
batch starts/ends, and when the consumer-stream kernels of the previous batch run beside it.
usage: timeline.py <kernel_trace.csv> [first_batch] [n_batches]"""
import csv, sys, collections
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
def short(n):
    n = n.replace("pbwtk::", "").replace("void ", "")
    return n.split("(")[0]
CHAIN = ("skel_hist", "skel_k2", "skel_rank", "step", "prepare")
# batches are delimited by transpose32_kernel on the chain stream
starts = [i for i, r in enumerate(rows) if "transpose32" in r[2]]
b0 = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t0 = rows[starts[b0]][0]
lo, hi = starts[b0], starts[min(b0 + nb, len(starts) - 1)]
agg = []
for r in rows[lo:hi]:
    nm = short(r[2])
    is_chain = any(nm.startswith(c) for c in CHAIN)
    if agg and agg[-1][0] == nm and (is_chain or nm.startswith("fill")):
        agg[-1][2] = r[1]; agg[-1][3] += 1; agg[-1][4] += r[1] - r[0]
    elif agg and is_chain and agg[-1][5] and agg[-1][0].startswith("skel") and nm.startswith("skel"):
        agg[-1][0] = "skel_round*"; agg[-1][2] = r[1]; agg[-1][3] += 1; agg[-1][4] += r[1] - r[0]
    else:
        agg.append([nm, r[0], r[1], 1, r[1] - r[0], is_chain, r[3]])
for nm, s, e, n, busy, ic, q in agg:
    print("%-10s q%-3s %-28s start %9.1f us  end %9.1f us  n=%4d  busy %8.1f us" % ("CHAIN" if ic else "", q, nm[:28], (s - t0) / 1e3, (e - t0) / 1e3, n, busy / 1e3))
