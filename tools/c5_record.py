"""tools/c5_record.py <full.json> <streamed_1m.json> <bench.json> -> profiles/r04_c5_full.json
BASELINE configs[4] at its own length: the 1 M x 10 M build + maxWithin with the panel generated per step, the same job at 1 M sites
(incl. pack3) beside the resident-panel run of bench.py's north_star_width, and what agrees."""
import json, sys
last = lambda p: json.loads(open(p).read().strip().split("\n")[-1])
full, one, bench = last(sys.argv[1]), last(sys.argv[2]), last(sys.argv[3])
ns = bench["north_star_width"]
out = {"configs4_full_length": full,
       "streamed_1m_sites_same_job_as_north_star_width": one,
       "resident_1m_sites_north_star_width": {k: ns[k] for k in ("haplotypes", "sites_timed", "seconds", "us_per_site", "value", "within_reports_hist_total")},
       "hist_total_streamed_equals_resident": one["within_reports_hist_total"] == ns["within_reports_hist_total"] and one["sites_timed"] == ns["sites_timed"],
       "streamed_over_resident_time": one["us_per_site"] / ns["us_per_site"],
       "note": "the streamed runs generate the panel (pbwtamd_synth_device, 1 M haplotypes per column) beside the job, one 8 192-site step ahead: the generator is a full-chip "
               "kernel of its own (~1 us per column after round 4's rewrite, 3.6 before), so the streamed rate is the job's rate beside a third throughput kernel, not "
               "within 3 % of the resident run; the histogram of the 1 M-site streamed job equals the resident run's exactly (same seed, same n_total). "
               "The 10 M-site run's mid-pass snapshot is NOT comparable with a 1 M-site panel's total: a panel that ends reports every open match at its last site (k == N sweep), "
               "a longer panel does not at that site."}
json.dump(out, open("profiles/r04_c5_full.json", "w"), indent=1)
print(json.dumps({k: out[k] for k in ("hist_total_streamed_equals_resident", "streamed_over_resident_time")}), full["seconds"], full["us_per_site"])
