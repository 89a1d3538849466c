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
import json
CHAIN = ("void pbwtk::skel_hist_kernel", "void pbwtk::skel_k2_kernel", "void pbwtk::skel_rank_kernel")


def chain_counter(counter, passname):
    """mean per dispatch of each chain kernel (hist, k2, rank) for one counter"""
    out = {}
    for p, k, c, n, m in rows:
        if p == passname and c == counter:
            for ck in CHAIN:
                if k.startswith(ck):
                    out[ck.split("::")[1]] = m
    return out


fetch_k, write_k = chain_counter("FETCH_SIZE", "pmc_fetch"), chain_counter("WRITE_SIZE", "pmc_write")
print("calibration (true/reported): ", cal)
cf, cw = cal.get("FETCH_SIZE", 1), cal.get("WRITE_SIZE", 1)
per_round = sum(fetch_k.values()) * cf * 1024 + sum(write_k.values()) * cw * 1024
per_launch = per_round / max(len(fetch_k), 1)
with open(os.path.join("profiles", tag + "_traffic.txt"), "w") as f:
    f.write("rocprofv3 PMC, separate passes (FETCH_SIZE, WRITE_SIZE), units KiB; calibration on tools/pmc_calib.hip\n")
    f.write("(1 GiB copy with 4 B/lane coalesced accesses): true/reported = %s\n" % cal)
    f.write("skeleton chain, one round = 8 sites = %d launches (M = 100000):\n" % len(fetch_k))
    for kname in fetch_k:
        f.write("  %-20s FETCH_SIZE %9.1f KiB x %.3f   WRITE_SIZE %9.1f KiB x %.3f\n" % (kname, fetch_k[kname], cf, write_k.get(kname, 0), cw))
    f.write("=> %.0f bytes HBM-side traffic per round, %.0f per launch (algorithmic: 16.125 B x M x 8 sites = %.0f per round)\n"
            % (per_round, per_launch, 16.125 * 100000 * 8))
print(open(os.path.join("profiles", tag + "_traffic.txt")).read())
json.dump({"100000": {"with_d": True, "kernel": "skeleton chain (skel_hist/k2/rank), mean over the launches of a round",
                      "bytes_per_launch": int(per_launch),
                      "source": "profiles/%s_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x%.1f per the calibration kernel tools/pmc_calib.hip)" % (tag, cf)}},
          open(os.path.join("profiles", "traffic.json"), "w"), indent=1)
sq_rows = [(k, c, m) for p, k, c, n, m in rows if p == "pmc_sq" and any(k.startswith(ck) for ck in CHAIN)]
if sq_rows:
    with open(os.path.join("profiles", tag + "_sq.txt"), "w") as f:
        f.write("rocprofv3 PMC (SQ block), chain kernels, mean per dispatch:\n")
        for ck in CHAIN:
            sq = {c: m for k, c, m in sq_rows if k.startswith(ck)}
            if not sq.get("SQ_WAVES"):
                continue
            w = sq["SQ_WAVES"]
            f.write("%s: waves %.0f; per wave: VALU %.0f  SALU %.0f  LDS %.0f instructions; wave-cycles (quad-cycle units) %.0f, of which waiting %.0f (%.0f%%), issuing %.0f\n"
                    % (ck.split("::")[1], w, sq.get("SQ_INSTS_VALU", 0) / w, sq.get("SQ_INSTS_SALU", 0) / w, sq.get("SQ_INSTS_LDS", 0) / w, sq.get("SQ_WAVE_CYCLES", 0) / w,
                       sq.get("SQ_WAIT_ANY", 0) / w, 100.0 * sq.get("SQ_WAIT_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1), sq.get("SQ_ACTIVE_INST_ANY", 0) / w))
    print(open(os.path.join("profiles", tag + "_sq.txt")).read())
