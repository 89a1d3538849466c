#!/usr/bin/env python
"""-matchDynamic alone at the north-star width (bench.py's match_dynamic leg), A/B over environment switches of a measurement build.
usage: python tools/md_bench.py [sites] [repeats]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pbwt_amd, bench
sites = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
for r in range(reps):
    out = bench.match_dynamic(torch, pbwt_amd, dev, 0, sites=sites)
    print(json.dumps({k: out[k] for k in out if k in ("us_per_site", "records", "value", "sites")}), flush=True)
