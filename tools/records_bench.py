"""the record-sink object of the bench line on its own: python tools/records_bench.py  (bench.py's match_records)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pbwt_amd, bench
dev = torch.device("cuda:0")
print(json.dumps(bench.match_records(torch, pbwt_amd, dev, 0), indent=1))
