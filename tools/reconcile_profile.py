"""tools/reconcile_profile.py <tag>: gpurun_out/<tag>/{bench.json, trace/bench_kernel_stats.csv, box.txt} -> profiles/<tag>_bench_and_profile.json
(the bench line and the kernel profile from the same box in the same gpurun, reconciled)."""
import csv, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join("gpurun_out", tag)
line = json.load(open(os.path.join(src, "bench.json")))
CHAIN = ("skel_hist_kernel", "skel_k2_kernel", "skel_k2_local_kernel", "skel_k2_wide_kernel", "skel_rank_kernel", "skel_team_kernel", "skel_onepass_kernel")
stats = {}
p = os.path.join(src, "trace", "bench_kernel_stats.csv")
for r in csv.DictReader(open(p)):
    for nm in CHAIN + ("skel_fillseq_kernel", "skel_fill_kernel", "sweep_hist_kernel", "p3r_scan_kernel", "p3r_emit_kernel", "p3r_combine_kernel", "transpose32_kernel", "skel_fillprep_kernel"):
        if ("::" + nm) in r["Name"]:
            c, t = stats.get(nm, (0, 0.0))
            stats[nm] = (c + int(r["Calls"]), t + float(r["TotalDurationNs"]))
avg_us = {k: v[1] / v[0] / 1e3 for k, v in stats.items()}
prof_line = None                                            # the profiled run prints its own bench line (slower: the profiler time-stamps every dispatch)
for ln in open(os.path.join(src, "trace.log"), errors="replace"):
    if ln.startswith("{") and '"metric"' in ln:
        prof_line = json.loads(ln)
M = line["config"]["haplotypes"]; S = line["config"]["sites_per_step"]
chain_sum = sum(avg_us[k] for k in CHAIN if k in avg_us)                    # one launch of each chain kernel = one round of 8 sites
rounds_per_step = S / 8
alg_per_round = 16.125 * M * 8
frac_csv = alg_per_round / (chain_sum * 1e-6) / 1e9 / 8000.0
out = {"box": open(os.path.join(src, "box.txt")).read().split("\n")[:6], "command": "python bench.py --steps 20 --warmup 5 (then rocprofv3 --kernel-trace --stats around the same command with --no-cpu --no-1m)",
       "bench_line": {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config")},
       "roofline_of_the_line": line["roofline"], "north_star_width": {k: v for k, v in (line.get("north_star_width") or {}).items() if k != "roofline"},
       "many_panels": line.get("many_panels"), "match_dynamic": {k: v for k, v in (line.get("match_dynamic") or {}).items() if k != "query_sharding_one_rank_share"},
       "kernel_avg_us_profiled_run": avg_us, "kernel_calls_profiled_run": {k: v[0] for k, v in stats.items()},
       "chain_kernel_sum_per_round_us": chain_sum, "rounds_per_step": rounds_per_step,
       "chain_kernel_time_per_step_ms": chain_sum * rounds_per_step / 1e3, "ms_per_step_of_the_line": line["ms_per_step"],
       "fits": chain_sum * rounds_per_step / 1e3 <= line["ms_per_step"],
       "profiled_run": None if prof_line is None else {"ms_per_step": prof_line["ms_per_step"], "value": prof_line["value"], "roofline_frac": prof_line["roofline"]["frac"],
                                                        "us_per_launch": prof_line["roofline"]["us_per_launch"],
                                                        "chain_kernel_time_fits_its_own_step": chain_sum * rounds_per_step / 1e3 <= prof_line["ms_per_step"]},
       "roofline_frac_from_csv_kernel_time_only": frac_csv, "roofline_frac_of_the_line_gaps_included": line["roofline"]["frac"],
       "note": "the line's frac divides by HIP-event time of the chain INCLUDING launch gaps (3 launches per round); the CSV figure divides by the kernels' own durations in the PROFILED run, "
               "which is a different execution of the same command on the same box: rocprofv3 time-stamps every dispatch, the 3-4 us chain kernels come out ~0.2 us longer each and the "
               "profiled run's own ms_per_step (profiled_run) is the figure the CSV sums have to fit inside"}
json.dump(out, open(os.path.join("profiles", tag + "_bench_and_profile.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("chain_kernel_sum_per_round_us", "chain_kernel_time_per_step_ms", "ms_per_step_of_the_line", "fits", "roofline_frac_from_csv_kernel_time_only", "roofline_frac_of_the_line_gaps_included")}, indent=1))
