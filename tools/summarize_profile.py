"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>_*.  Usage: summarize_profile.py r01"""
import collections
import csv
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)
shutil.copy(os.path.join(src, "trace", "bench_kernel_stats.csv"), os.path.join("profiles", tag + "_kernel_stats.csv"))


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return d


rows = []
for name in ("calib_fetch/calib", "calib_write/calib", "pmc_fetch/bench", "pmc_write/bench", "pmc_sq/bench"):
    if not os.path.exists(os.path.join(src, name + "_counter_collection.csv")):
        continue
    for (k, c), v in sorted(agg(os.path.join(src, name + "_counter_collection.csv")).items(), key=lambda kv: -sum(kv[1])):
        rows.append((name.split("/")[0], k, c, len(v), sum(v) / len(v)))
with open(os.path.join("profiles", tag + "_pmc_summary.csv"), "w") as f:
    f.write("pass,kernel,counter,dispatches,mean_value_KiB\n")
    for r in rows:
        f.write('%s,"%s",%s,%d,%.3f\n' % r)
# calibration factors: known bytes / reported
cal = {}
for p, k, c, n, m in rows:
    if k.startswith("calib_copy4"):
        cal[c] = (256 << 20) * 4 / 1024.0 / m
step = {c: m for p, k, c, n, m in rows if k.startswith("void pbwtk::step") and p.startswith("pmc")}
print("calibration (true/reported): ", cal)
fetch = step.get("FETCH_SIZE", 0) * cal.get("FETCH_SIZE", 1) * 1024
write = step.get("WRITE_SIZE", 0) * cal.get("WRITE_SIZE", 1) * 1024
print("step_kernel HBM-side traffic per launch: fetch %.0f B + write %.0f B = %.0f B" % (fetch, write, fetch + write))
sq = {c: m for p, k, c, n, m in rows if k.startswith("void pbwtk::step") and p == "pmc_sq"}
if sq:
    with open(os.path.join("profiles", tag + "_sq.txt"), "w") as f:
        f.write("rocprofv3 PMC (SQ block), chain kernel, mean per dispatch:\n")
        for c in sorted(sq):
            f.write("  %-20s %.1f\n" % (c, sq[c]))
        if sq.get("SQ_WAVES"):
            w = sq["SQ_WAVES"]
            f.write("per wave: VALU %.0f  SALU %.0f  LDS %.0f instructions; wave-cycles (quad-cycle units) %.0f, of which waiting %.0f (%.0f%%), issuing %.0f\n"
                    % (sq.get("SQ_INSTS_VALU", 0) / w, sq.get("SQ_INSTS_SALU", 0) / w, sq.get("SQ_INSTS_LDS", 0) / w, sq.get("SQ_WAVE_CYCLES", 0) / w,
                       sq.get("SQ_WAIT_ANY", 0) / w, 100.0 * sq.get("SQ_WAIT_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1), sq.get("SQ_ACTIVE_INST_ANY", 0) / w))
    print(open(os.path.join("profiles", tag + "_sq.txt")).read())
with open(os.path.join("profiles", tag + "_traffic.txt"), "w") as f:
    f.write("rocprofv3 PMC, separate passes (FETCH_SIZE, WRITE_SIZE), units KiB; calibration on tools/pmc_calib.hip\n")
    f.write("(1 GiB copy with 4 B/lane coalesced accesses, the step kernel's pattern): true/reported = %s\n" % cal)
    f.write("step_kernel per launch: FETCH_SIZE %.1f KiB x %.3f, WRITE_SIZE %.1f KiB x %.3f => %.0f bytes HBM-side traffic\n"
            % (step.get("FETCH_SIZE", 0), cal.get("FETCH_SIZE", 1), step.get("WRITE_SIZE", 0), cal.get("WRITE_SIZE", 1), fetch + write))
