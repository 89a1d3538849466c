"""bench.py's many_panels object on its own: python tools/many_bench.py [P] [M] [sites]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pbwt_amd, bench
dev = torch.device("cuda:0")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
sites = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
opts = pbwt_amd.OPT_WITH_D | pbwt_amd.OPT_WITHIN_HIST | pbwt_amd.OPT_PACK3
o = bench.many_panels(torch, pbwt_amd, dev, opts, 0, M, P=P, sites=sites)
print("P %d M %d: %.3e site*haps/s over all panels, %.3f us/site/panel, hist total %d" % (P, M, o["value"], o["us_per_site_per_panel"], o["within_reports_hist_total"]))
