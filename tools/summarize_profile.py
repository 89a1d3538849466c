"""Condense gpurun_out/<tag>/ (tools/profile_round.sh) into profiles/<tag>_*.  Usage: summarize_profile.py r02"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join("gpurun_out", tag)
os.makedirs("profiles", exist_ok=True)


def copy_stats(sub, stem, dst):
    p = os.path.join(src, sub, stem + "_kernel_stats.csv")
    if os.path.exists(p):
        shutil.copy(p, os.path.join("profiles", dst))
        return True
    return False


copy_stats("trace", "bench", tag + "_kernel_stats.csv")                 # bench.py --steps 2 --warmup 1 (configs[2] width)
copy_stats("wide_trace", "wide", tag + "_wide_kernel_stats.csv")        # tools/wide_bench.py 1000000 2048 hp (north-star width)
copy_stats("qs_trace", "qs", tag + "_matchdynamic_kernel_stats.csv")    # tools/qsweep_bench.py 1000000 10000 4096
copy_stats("wide_alone", "wide", tag + "_wide_chain_alone_kernel_stats.csv")   # tools/wide_bench.py 1000000 4096 none
copy_stats("shard_trace", "shard", tag + "_posshard_1rank_kernel_stats.csv")   # bench.py --mode posshard --haps 1000000 --steps 1 (one rank)
for extra in ("overlap.txt", "matchdynamic_busy.txt", "consumer_pricing.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join("profiles", tag + "_" + extra))


def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return d


rows = []
for name in ("calib_fetch/calib", "calib_write/calib", "pmc_fetch/bench", "pmc_write/bench", "pmc_sq/bench", "wide_fetch/wide", "wide_write/wide", "wide_sq/wide"):
    p = os.path.join(src, name + "_counter_collection.csv")
    if not os.path.exists(p):
        continue
    for (k, c), v in sorted(agg(p).items(), key=lambda kv: -sum(kv[1])):
        rows.append((name.split("/")[0], k, c, len(v), sum(v) / len(v)))
with open(os.path.join("profiles", tag + "_pmc_summary.csv"), "w") as f:
    f.write("pass,kernel,counter,dispatches,mean_value\n")
    for r in rows:
        f.write('%s,"%s",%s,%d,%.3f\n' % r)
cal = {}
for p, k, c, n, m in rows:
    if k.startswith("calib_copy4"):
        cal[c] = (256 << 20) * 4 / 1024.0 / m                      # known bytes (1 GiB copy, 4 B/lane) / reported KiB
cf, cw = cal.get("FETCH_SIZE", 2.0), cal.get("WRITE_SIZE", 1.0)
CHAIN = ("skel_hist_kernel", "skel_k2_kernel", "skel_k2_local_kernel", "skel_k2_wide_kernel", "skel_rank_kernel", "skel_team_kernel", "skel_onepass_kernel")
CONS = ("skel_totals_kernel", "skel_fill_kernel", "skel_fillseq_kernel", "skel_fillprep_kernel", "sweep_hist_kernel", "p3r_scan_kernel", "p3r_combine_kernel", "p3r_emit_kernel", "transpose32_kernel")


def counters(passname, counter, names):
    out = {}
    for p, k, c, n, m in rows:
        if p == passname and c == counter:
            for nm in names:
                if ("::" + nm) in k:
                    out[nm] = out.get(nm, 0.0) + m * n          # total over the dispatches of the run
                    out[nm + "#n"] = out.get(nm + "#n", 0) + n
    return out


def avg_ns(stats_csv):
    """kernel name fragment -> average duration (ns) from a rocprofv3 --stats summary (the kernel-trace pass of the same command)"""
    out = {}
    if os.path.exists(stats_csv):
        for r in csv.DictReader(open(stats_csv)):
            for nm in CHAIN + CONS:
                if ("::" + nm) in r["Name"]:
                    calls, tot = int(r["Calls"]), float(r["TotalDurationNs"])
                    c0, t0 = out.get(nm, (0, 0.0))
                    out[nm] = (c0 + calls, t0 + tot)
    return {k: v[1] / max(v[0], 1) for k, v in out.items()}


with open(os.path.join("profiles", tag + "_traffic.txt"), "w") as f:
    f.write("rocprofv3 PMC, separate passes (FETCH_SIZE, WRITE_SIZE), units KiB; calibration on tools/pmc_calib.hip\n")
    f.write("(1 GiB copy with 4 B/lane coalesced accesses): true/reported = %s  (FETCH_SIZE x2 on gfx950 as MI355X_MICROARCH.md says)\n\n" % cal)
    result = {}
    f.write("us / GB/s: the kernel's average duration in the kernel-trace pass of the same command and its HBM-side bytes per dispatch over it\n\n")
    for label, fp, wp, M, sites, stats in (("configs[2] width, bench.py --steps 2 --warmup 1 (M = 100000, 3 x 8192 sites)", "pmc_fetch", "pmc_write", 100000, 3 * 8192, tag + "_kernel_stats.csv"),
                                           ("north-star width, tools/wide_bench.py (M = 1000000, 512 + 2048 sites)", "wide_fetch", "wide_write", 1000000, 2560, tag + "_wide_kernel_stats.csv")):
        dur = avg_ns(os.path.join("profiles", stats))
        fe, wr = counters(fp, "FETCH_SIZE", CHAIN + CONS), counters(wp, "WRITE_SIZE", CHAIN + CONS)
        if not fe:
            continue
        f.write(label + "\n")
        tot = 0.0
        chain_bytes, chain_launches = 0.0, 0
        for nm in CHAIN + CONS:
            if nm not in fe:
                continue
            b = (fe[nm] * cf + wr.get(nm, 0.0) * cw) * 1024
            tot += b
            if nm in CHAIN:
                chain_bytes += b; chain_launches += fe[nm + "#n"]
            per = b / max(fe[nm + "#n"], 1)
            rate = ("  %9.1f us  %7.0f GB/s" % (dur[nm] / 1e3, per / dur[nm])) if nm in dur else ""
            f.write("  %-22s %6d dispatches  fetch %12.0f KiB x %.2f  write %12.0f KiB x %.2f  = %8.1f MB%s\n"
                    % (nm, fe[nm + "#n"], fe[nm], cf, wr.get(nm, 0.0), cw, b / 1e6, rate))
        alg = 16.125 * M * sites
        f.write("  => %.1f MB HBM-side traffic for %d sites = %.2f MB/site; algorithmic 16.125 B x M = %.2f MB/site: ratio %.2f\n"
                % (tot / 1e6, sites, tot / 1e6 / sites, 16.125 * M / 1e6, tot / alg))
        if chain_launches:
            f.write("  chain: %.0f bytes per launch (%d launches)\n\n" % (chain_bytes / chain_launches, chain_launches))
            result[str(M)] = {"with_d": True, "kernel": "the chain's launches, mean (one-launch round: skel_onepass_kernel; wider panels: skel_hist / k2 / rank)",
                              "bytes_per_launch": int(chain_bytes / chain_launches),
                              "source": "profiles/%s_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x%.1f per the calibration kernel tools/pmc_calib.hip)" % (tag, cf)}
    if result:
        json.dump(result, open(os.path.join("profiles", "traffic.json"), "w"), indent=1)
print(open(os.path.join("profiles", tag + "_traffic.txt")).read())

with open(os.path.join("profiles", tag + "_sq.txt"), "w") as f:
    f.write("rocprofv3 PMC (SQ block), mean per dispatch; wave-cycles in the counter's quad-cycle units\n")
    for label, pn in (("configs[2] width (M = 100000)", "pmc_sq"), ("north-star width (M = 1000000)", "wide_sq")):
        d = collections.defaultdict(dict)
        for p, k, c, n, m in rows:
            if p == pn:
                d[k][c] = m
        if not d:
            continue
        f.write("\n" + label + "\n")
        for k in sorted(d):
            if not any(("::" + nm) in k for nm in CHAIN + CONS):
                continue
            sq = d[k]
            w = sq.get("SQ_WAVES", 0)
            if not w:
                continue
            f.write("%-64s waves %9.0f; per wave: VALU %6.0f SALU %6.0f LDS %5.0f; wave-cycles %8.0f, waiting %8.0f (%.0f%%), issuing %7.0f\n"
                    % (k[:64], w, sq.get("SQ_INSTS_VALU", 0) / w, sq.get("SQ_INSTS_SALU", 0) / w, sq.get("SQ_INSTS_LDS", 0) / w, sq.get("SQ_WAVE_CYCLES", 0) / w,
                       sq.get("SQ_WAIT_ANY", 0) / w, 100.0 * sq.get("SQ_WAIT_ANY", 0) / max(sq.get("SQ_WAVE_CYCLES", 1), 1), sq.get("SQ_ACTIVE_INST_ANY", 0) / w))
# the VALU / SALU issue bound of every kernel against its measured duration: a wave64 VALU instruction occupies its SIMD for 2 cycles (SIMD-32),
# a SALU instruction the scalar unit of its CU for 1 (one scalar unit per SIMD: 4 per CU, 1 024 per chip); 1 024 SIMDs at CLOCK_GHZ.  "issue bound" = the time the
# chip would need if nothing but those instructions were issued, perfectly spread; a kernel at 50 % of it can only get faster by executing FEWER instructions.
CLOCK_GHZ, NSIMD = 2.4, 1024
with open(os.path.join("profiles", tag + "_sq.txt"), "a") as f:
    f.write("\nissue bound per dispatch = waves x (2 x VALU + SALU) cycles / (%d SIMDs x %.1f GHz), against the kernel's average duration in the kernel-trace pass of the same command\n" % (NSIMD, CLOCK_GHZ))
    for label, pn, stats_sub, stem in (("configs[2] width (M = 100000)", "pmc_sq", "trace", "bench"), ("north-star width (M = 1000000)", "wide_sq", "wide_trace", "wide")):
        d = collections.defaultdict(dict)
        for pp, k, c, n, m in rows:
            if pp == pn:
                d[k][c] = m
        dur = {}
        sp = os.path.join(src, stats_sub, stem + "_kernel_stats.csv")
        if os.path.exists(sp):
            for r in csv.DictReader(open(sp)):
                dur[r["Name"]] = float(r["AverageNs"]) / 1e3
        if not d or not dur:
            continue
        f.write("\n" + label + "\n")
        for k in sorted(d):
            if not any(("::" + nm) in k for nm in CHAIN + CONS):
                continue
            sq = d[k]; w = sq.get("SQ_WAVES", 0)
            if not w or k not in dur:
                continue
            valu_us = sq.get("SQ_INSTS_VALU", 0) * 2 / (NSIMD * CLOCK_GHZ * 1e3)
            salu_us = sq.get("SQ_INSTS_SALU", 0) * 1 / (NSIMD * CLOCK_GHZ * 1e3)
            f.write("%-64s measured %8.1f us; VALU issue %7.1f us, SALU issue %7.1f us, together %7.1f us = %3.0f%% of measured\n"
                    % (k[:64], dur[k], valu_us, salu_us, valu_us + salu_us, 100.0 * (valu_us + salu_us) / dur[k]))
print(open(os.path.join("profiles", tag + "_sq.txt")).read())
