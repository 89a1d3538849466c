#!/usr/bin/env python
"""bench.py — PBWT build + maxWithin throughput on MI355X (sites*haplotypes/sec) with the chain
kernel's achieved algorithmic HBM GB/s against the gfx950 roofline, beside a CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] ITSELF): synthetic founder-mosaic panel, M = 100,000 haplotypes x N = K*S sites; one
"step" = S = 50,000 consecutive sites pushed through the hot path with the panel's bit columns already
resident in HBM: per site pbwtCursorForwardsAD (a[], d[]) + the matchMaximalWithin sweep (histogram
sink, the reference's -stats mode) + pack3 encoding of the PBWT column.  The timed region is ONE pass
from site 0 to site N (the k == N sweep included): with the default (and the driver's) K = 20 steps that
is the 100,000 x 1,000,000 panel of configs[2], whole; the W warm-up steps run a separate, untimed pass over
the first W*S sites of the same panel.  tests/test_gpu_z_c3_full.py runs this very pass (same seed, same
calls) under the oracle, block by block from the device's own checkpoints, and pins its histogram total
(tests/golden/c3_full.json), which this script compares its own total with.

Multi-GPU (⑤), two modes:
  --mode replicas (default): ranks process independent panels (different chromosomes = seeds) with no data-path
      collective: weak scaling, value = total site*haps over all ranks / max time.
  --mode posshard: ONE panel of K*S sites with the RECURRENCE position-sharded over the ranks (BASELINE configs[3]: pbwt_amd/posshard.py,
      csrc/pbwt_shard.inc): rank g owns a range of positions of the sorted order; per round of 8 sites one row exchange and the
      scatter as peer stores over xGMI (hipIpc mappings), consumers sharded by site inside every batch; one all-reduce (RCCL) of
      the histogram at the end, inside the timed region.  Strong scaling, value = M*K*S / max time.  Default M = 1,000,000.
  --mode siteblock: ONE panel of K*S sites sharded by site blocks (pbwt_amd/siteblock.py): every rank
      runs the chain alone up to its block, then the full hot path over its block; one all-reduce (RCCL) of the
      histogram at the end, inside the timed region.  Strong scaling, value = M*K*S / max time; bounded by
      t_full / t_chain because the recurrence itself stays serial (1.2x at M = 100 k, 2.1x at M = 1 M on 8 GPUs).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # hipIpc (the position-sharded rings) needs the dmabuf IPC mode on this driver stack; read when HSA initialises

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_SITEHAP = 16.125        # r a,d + w a,d (4 B each) + 1 allele bit
HBM_PEAK_GBPS = 8000.0                # MI355X_MICROARCH.md: 8 TB/s HBM3E
MANY_PANELS_P = 8                     # the `many_panels` object: a fixed number of panels (no best-of-tried)
BOUNDARY_US = 1.54                    # measured cost of one dependent kernel boundary on MI355X, empty kernels (profiles/r01_probes.txt)


def latency_bound(alg_bytes_per_launch, us_per_launch):
    """site k+1 needs the whole order of site k, so every launch of the chain pays one device-wide dependency; the
    reachable HBM fraction is bounded by (time the launch's algorithmic bytes take at the peak rate) / (one kernel boundary).
    Reported next to the measured fraction: frac / latency_bound_frac says how much of what the dependency allows is reached."""
    data_us = alg_bytes_per_launch / (HBM_PEAK_GBPS * 1e9) * 1e6
    return {"boundary_us": BOUNDARY_US, "data_us_per_launch": data_us, "latency_bound_frac": min(1.0, data_us / BOUNDARY_US),
            "measured_us_per_launch": us_per_launch,
            "note": "upper bound on frac for a chain of dependent launches: data time of one launch at 8 TB/s / the 1.54 us a dependent kernel boundary costs"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--haps", type=int, default=None, help="M, haplotypes in the panel (default 100000; 1000000 in --mode posshard)")
    ap.add_argument("--backend", default=os.environ.get("PBWT_BENCH_BACKEND", "nccl"), help="torch.distributed backend (nccl = RCCL; gloo lets several ranks share ONE GPU)")
    ap.add_argument("--sites-per-step", type=int, default=50000, help="S, sites per step (a multiple of 8); K*S = the panel's sites: 20 x 50 000 = configs[2]")
    ap.add_argument("--batch", type=int, default=512, help="sites per device batch (graph length)")
    ap.add_argument("--kind", type=int, default=0, help="0 founder mosaic, 1 iid")
    ap.add_argument("--no-within", action="store_true")
    ap.add_argument("--no-pack3", action="store_true")
    ap.add_argument("--cpu-sites", type=int, default=32768, help="sites of the same panel timed on the CPU oracle")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-1m", action="store_true", help="skip the measurements at the north-star width (1M haplotypes)")
    ap.add_argument("--ns-sites", type=int, default=1000000, help="sites of the north-star job (1M haplotypes x this many sites; the bit panel, 125 KB per site, is resident)")
    ap.add_argument("--ps-sites", type=int, default=None, help="sites of the position-sharded job attached at N > 1 (default: --ns-sites, i.e. the north-star job itself)")
    ap.add_argument("--ps-timeout", type=float, default=900.0, help="seconds after which the attached position-sharded job is given up (the line is printed without it)")
    ap.add_argument("--no-posshard", action="store_true", help="N > 1: do not attach the position-sharded job")
    ap.add_argument("--stream-panel", action="store_true", help="run ONLY the north-star job with the panel generated per step into a column ring (configs[4] at its own length: --ns-sites 10000000); prints that object")
    ap.add_argument("--with-queries", action="store_true", help="--stream-panel: configs[4] whole — build + maxWithin + pack3 + -matchDynamic of 10 000 queries in ONE streamed pass (pbwtamd_match_sweep_stream)")
    ap.add_argument("--ns-no-pack3", action="store_true", help="--stream-panel: build + maxWithin without the pack3 consumer (10 M sites of .pbwt bytes do not fit beside the job)")
    ap.add_argument("--own-stream", action="store_true", help="let the engine create its own (high-priority) chain stream instead of torch's current stream")
    ap.add_argument("--mode", default=os.environ.get("PBWT_BENCH_MODE", "replicas"), choices=["replicas", "siteblock", "posshard"],
                    help="multi-GPU mode: independent panels per rank (weak) or one panel sharded by site blocks (strong)")
    ap.add_argument("--panels", type=int, default=1, help="independent panels (chromosomes) advanced by the same chain launches on this GPU (pbwtamd_pass_advance_many); default 1 = the named config")
    args = ap.parse_args()
    if args.haps is None:
        args.haps = 1000000 if args.mode == "posshard" else 100000
    return args


def cpu_baseline(args, first_cols):
    """CPU baseline on a bounded sample of the same workload, 1 thread (the reference is
    single-threaded): the REAL reference (oracle/_ref, compiled in place from the reference's own
    sources; kind "reference") when its prebuilt library travelled with the repo, else the oracle's
    C restatement (kind "port").  Build with d + pack3, then the -stats maxWithin sweep."""
    import oracle
    M = args.haps
    n = first_cols.shape[0]
    if oracle.ref() is not None and not args.no_within:
        tb, tw = oracle.ref_time_build_and_within(first_cols, M)
        kind, what = "reference", "richarddurbin/pbwt compiled from its own sources (oracle/_ref)"
    else:
        t0 = time.perf_counter()
        b = oracle.build_bitcols(first_cols, M, with_d=True, want_csum=False)
        t1 = time.perf_counter()
        if not args.no_within:
            oracle.max_within_hist(b["yz"], M, n)
        tb, tw = t1 - t0, time.perf_counter() - t1
        kind, what = "port", "oracle C restatement"
    model, ncpu = host_cpu()
    return {"value": M * n / (tb + tw), "unit": "site*haps/s", "cores": 1, "kind": kind, "host_cpu": model, "host_logical_cpus": ncpu,
            "sample": "first %d sites of the same %d-haplotype panel, %s: build(WriteForwardsAD + pack3) %.2fs + -stats maxWithin %.2fs"
                      % (n, M, what, tb, tw)}


def verified_total(M, N, args, unit, got):
    """tests/golden/c3_full.json: the histogram total of THIS pass (same seed, same pbwt_amd calls) as tests/test_gpu_z_c3_full.py got it with every
    block of sites checked against the oracle from the device's own checkpoints.  None when the run is not that configuration."""
    try:
        g = json.load(open(os.path.join(ROOT, "tests", "golden", "c3_full.json")))
    except Exception:
        return None
    if (M, N, args.kind, unit) != (g["haplotypes"], g["sites"], g["kind"], 0) or args.no_within or args.panels != 1:
        return None
    return {"expected": g["within_reports_hist_total"], "matches": got == g["within_reports_hist_total"], "source": "tests/golden/c3_full.json (tests/test_gpu_z_c3_full.py)"}


def host_cpu():
    """model string and logical CPUs of the host the cpu_baseline ran on (BASELINE.md section 4 asks for both)"""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return model, os.cpu_count()


def north_star_streamed(torch, pbwt_amd, dev, opts, kind, M=1000000, sites=10000000, batch=512, step=8192, snapshot_at=1000000):
    """BASELINE configs[4] at its own length (1 M haplotypes x 10 M sites): the bit panel would be 1.25 TB, so it is GENERATED per step of 8 192 sites
    (pbwtamd_synth_device on a second engine's stream — the same counter-based generator, column by column) into a two-slot column ring, one step
    ahead of the chain; events order the generator behind the chain's last read of a slot and the chain behind the generator.  Same seed and pass
    structure as north_star_width(), so the histogram total after the first `snapshot_at` sites (read once, mid-pass) is comparable with that run's."""
    sites = (sites // batch) * batch                        # as north_star_width(): the same n_total for the same --ns-sites, so the totals are comparable
    s_chain, s_gen = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    eng = pbwt_amd.Engine(M, batch_sites=batch, device=dev.index, stream=s_chain.cuda_stream)
    gen = pbwt_amd.Engine(M, batch_sites=8, device=dev.index, stream=s_gen.cuda_stream)        # only its generator and its stream are used
    n_total = sites + batch
    look = 8
    ring = [torch.empty((step + look, eng.wpc), dtype=torch.int32, device=dev) for _ in range(2)]
    ev_gen = [torch.cuda.Event() for _ in range(2)]; ev_read = [torch.cuda.Event() for _ in range(2)]
    torch.cuda.synchronize()

    def generate(slot, k0):                                 # columns k0 .. k0 + step + look - 1 (clipped to the panel) into ring[slot]
        n = min(step + look, n_total - k0)
        s_gen.wait_event(ev_read[slot])
        gen.synth_device(ring[slot].data_ptr(), k0, n, seed=0x1A2B3C, kind=kind)
        ev_gen[slot].record(s_gen)

    for sl in range(2):
        ev_read[sl].record(s_chain)
    eng.pass_begin(n_total)
    generate(0, 0)
    s_chain.wait_event(ev_gen[0])
    eng.pass_advance(ring[0].data_ptr(), batch, batch + look, opts)          # warm-up batch (untimed), as in north_star_width()
    eng.sync()
    # the timed region: steps of `step` sites starting at site `batch`; slot i % 2 holds the columns of step i
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k, i = batch, 0
    generate(1, k)
    snap = None
    while k < n_total:
        n = min(step, n_total - k)
        slot = (i + 1) % 2
        s_chain.wait_event(ev_gen[slot])
        eng.pass_advance(ring[slot].data_ptr(), n, min(n + look, n_total - k), opts)
        ev_read[slot].record(s_chain)                        # the chain's last read of this slot (transpose + key gathers) is enqueued
        k += n; i += 1
        if k < n_total:
            generate((i + 1) % 2, k)
        if snap is None and k >= snapshot_at + batch:
            t_s = time.perf_counter()
            snap = {"sites": k, "within_reports_hist_total": int(eng.get_hist(n_total + 1).sum()), "seconds_so_far": time.perf_counter() - t0}
            snap["snapshot_cost_s"] = time.perf_counter() - t_s
    eng.pass_end(opts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hist = eng.get_hist(n_total + 1)
    eng.close(); gen.close()
    return {"haplotypes": M, "sites_timed": sites, "seconds": dt, "value": M * sites / dt, "unit": "site*haps/s", "us_per_site": 1e6 * dt / sites,
            "within_reports_hist_total": int(hist.sum()), "snapshot": snap,
            "whole_job_frac_of_hbm_peak": ALG_BYTES_PER_SITEHAP * M * sites / dt / 1e9 / HBM_PEAK_GBPS,
            "panel": "generated per %d-site step into a two-slot column ring (%.1f GB instead of %.0f GB resident), one step ahead of the chain"
                     % (step, 2 * (step + look) * ((M + 31) // 32) * 4 / 1e9, n_total * ((M + 31) // 32) * 4 / 1e9),
            "pack3": bool(opts & pbwt_amd.OPT_PACK3)}


def configs4_streamed_with_queries(torch, pbwt_amd, dev, kind, M=1000000, Q=10000, sites=10000000, batch=512, pack3=True):
    """BASELINE configs[4] at its own length, ALL of it in one pass: 1 M haplotypes x `sites` sites build + -maxWithin (histogram) + pack3 + -matchDynamic of a
    Q-haplotype query panel (matchSequencesSweep, pbwtMatch.c:363-443) through the STREAMED entry point pbwtamd_match_sweep_stream — the library asks for both
    panels' bit columns a device batch at a time (generated here on a second stream, one batch ahead: panel and queries are the two parts of ONE (M + Q)-wide
    synthetic panel, shared founders), hands the records over per batch (counted, not kept) and the pack3 bytes leave through pbwtamd_drain_packed into a
    pinned buffer (counted).  Nothing of either panel ever exists beyond two batches of columns."""
    assert M % 32 == 0
    sites = (sites // batch) * batch
    s_gen = torch.cuda.Stream(device=dev)
    eng = pbwt_amd.Engine(M, batch_sites=batch, device=dev.index)
    full = pbwt_amd.Engine(M + Q, batch_sites=8, device=dev.index, stream=s_gen.cuda_stream)      # only its generator and its stream are used
    wq = pbwt_amd.wpc_for(Q)
    look = 8
    stage = [torch.empty((batch + look, full.wpc), dtype=torch.int32, device=dev) for _ in range(2)]
    pcols = [torch.zeros((batch + look, eng.wpc), dtype=torch.int32, device=dev) for _ in range(2)]
    qcols = [torch.zeros((batch + look, wq), dtype=torch.int32, device=dev) for _ in range(2)]
    nqw = (Q + 31) // 32
    ev = [torch.cuda.Event() for _ in range(2)]
    state = {"next": 0, "records": 0, "bytes": 0, "calls": 0, "t_snap": None}
    pinned = torch.empty(64 << 20, dtype=torch.uint8).pin_memory().numpy() if pack3 else None

    def generate(slot, k0):
        n = min(batch + look, sites - k0)
        with torch.cuda.stream(s_gen):
            full.synth_device(stage[slot].data_ptr(), k0, n, seed=0x3D, kind=kind)
            pcols[slot][:n, : M // 32] = stage[slot][:n, : M // 32]
            qcols[slot][:n, :nqw] = stage[slot][:n, M // 32: M // 32 + nqw]
            if Q % 32:
                qcols[slot][:n, nqw - 1] &= (1 << (Q % 32)) - 1
            ev[slot].record(s_gen)

    def cols(site0, ncols):
        slot = (site0 // batch) % 2
        if state["calls"] == 0:
            generate(slot, site0)
        ev[slot].synchronize()                              # this batch's columns are complete
        if site0 + batch < sites:
            generate(slot ^ 1, site0 + batch)               # the next batch's are generated while this one runs (its slot's last reader, two batches back, is done)
        if pinned is not None and site0:
            state["bytes"] += len(eng.drain_packed(pinned))
        state["calls"] += 1
        return pcols[slot].data_ptr(), qcols[slot].data_ptr()

    def on_records(arr):
        state["records"] += len(arr)

    popts = pbwt_amd.OPT_WITHIN_HIST | (pbwt_amd.OPT_PACK3 if pack3 else 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _, nom, tot = eng.match_sweep_stream(sites, Q, cols, on_records=on_records, panel_opts=popts)
    if pinned is not None:
        state["bytes"] += len(eng.drain_packed(pinned))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hist = eng.get_hist(sites + 1)
    eng.close(); full.close()
    alg = (ALG_BYTES_PER_SITEHAP * (M + Q) * sites + 16.0 * state["records"]) / dt / 1e9
    return {"haplotypes": M, "queries": Q, "sites": sites, "seconds": dt, "us_per_site": 1e6 * dt / sites, "value": M * sites / dt, "unit": "panel site*haps/s",
            "records": int(state["records"]), "no_match_events": int(nom), "nTot": int(tot[0]), "totLen": int(tot[1]),
            "within_reports_hist_total": int(hist.sum()), "pack3_bytes": int(state["bytes"]), "pack3": bool(pack3),
            "achieved_GBps": alg, "frac_of_hbm_peak": alg / HBM_PEAK_GBPS,
            "note": "ONE pass: build (ForwardsAD) + maxWithin histogram + pack3 + matchSequencesSweep; both panels generated per %d-site batch on a second stream "
                    "(pbwtamd_match_sweep_stream), records and .pbwt bytes handed over per batch; set-up and the closing tails included" % batch}


def north_star_width(torch, pbwt_amd, dev, opts, kind, M=1000000, sites=1000000, batch=512, step=8192, want_hist=False):
    """the north-star job itself: build + -maxWithin (histogram sink) + pack3 on 1 M haplotypes x `sites` sites (default: all
    1 M sites; the 125 GB bit panel is resident in HBM before the timed region), with its own roofline object.  The headline
    `value` stays BASELINE configs[2]; the histogram of a prefix is pinned to the oracle by
    tests/test_gpu_z_configs.py::test_bench_north_star_width_path (this same function)."""
    sites = (sites // batch) * batch
    eng = pbwt_amd.Engine(M, batch_sites=batch, device=dev.index)
    n_total = sites + batch
    panel = torch.empty((n_total, eng.wpc), dtype=torch.int32, device=dev)
    eng.synth_device(panel.data_ptr(), 0, n_total, seed=0x1A2B3C, kind=kind)
    eng.sync()
    col = lambda k: panel.data_ptr() + k * eng.wpc * 4
    eng.pass_begin(n_total)
    eng.pass_advance(col(0), batch, batch + 8, opts)                     # warm-up batch
    eng.sync()
    ms0, n0 = eng.chain_timing(); s0 = eng.chain_sites()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = batch
    while k < n_total:
        n = min(step, n_total - k)
        eng.pass_advance(col(k), n, min(n + 8, n_total - k), opts)
        k += n
    eng.pass_end(opts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms1, n1 = eng.chain_timing(); s1 = eng.chain_sites()
    us = 1e3 * (ms1 - ms0) / max(n1 - n0, 1)
    spl = (s1 - s0) / max(n1 - n0, 1)
    ach = ALG_BYTES_PER_SITEHAP * M * spl / (us * 1e-6) / 1e9
    hist = eng.get_hist(n_total + 1)
    alg_pl = ALG_BYTES_PER_SITEHAP * M * spl
    out = {"haplotypes": M, "sites_timed": sites, "seconds": dt, "value": M * sites / dt, "unit": "site*haps/s", "us_per_site": 1e6 * dt / sites,
           "within_reports_hist_total": int(hist.sum()),
           "whole_job_achieved_GBps": ALG_BYTES_PER_SITEHAP * M * sites / dt / 1e9,
           "whole_job_frac_of_hbm_peak": ALG_BYTES_PER_SITEHAP * M * sites / dt / 1e9 / HBM_PEAK_GBPS,
           "roofline": {"bound": "hbm", "kernel": "skeleton chain: skel_hist_kernel + skel_k2_wide_kernel + skel_rank_kernel, 3 launches per 8 sites",
                        "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                        "us_per_launch": us, "sites_per_launch": spl, "latency_bound": latency_bound(alg_pl, us),
                        "note": "chain launches only (HIP events around the dependent chain, launch gaps included); whole_job_* = the job's algorithmic bytes over wall time, consumers included"}}
    if want_hist:
        out["hist"] = hist
        out["packed"] = eng.get_packed() if (opts & pbwt_amd.OPT_PACK3) else None
        out["panel"] = panel
    eng.close()
    return out


def many_panels(torch, pbwt_amd, dev, opts, kind, M, P=2, sites=32768, batch=512, step=8192):
    """whole-genome throughput: P independent panels (chromosomes) of M haplotypes advanced by the SAME chain launches
    (pbwtamd_pass_advance_many: grid.y = panel).  Below ~250 k haplotypes a chain launch costs its 3-4 us whatever runs inside it, so the
    panels share that cost — until chain plus consumers fill the GPU: at 100 k haplotypes P=2 is the best point measured (1.4x one panel; P=4, 8 give
    1.2x, chain-only 2.2x), at 10 k-25 k P=16 gives 2.5-3x.  A labelled secondary object, never `value`; every panel's output is pinned to the oracle by
    tests/test_gpu_parity.py::test_many_panels_per_launch."""
    shared = torch.cuda.Stream(device=dev)                  # a null stream handle would give every engine a stream of its own
    st = shared.cuda_stream
    engs = [pbwt_amd.Engine(M, batch_sites=batch, device=dev.index, stream=st) for _ in range(P)]
    n_total = sites + batch
    bufs = [torch.empty((n_total, engs[0].wpc), dtype=torch.int32, device=dev) for _ in range(P)]
    for p in range(P):
        engs[p].synth_device(bufs[p].data_ptr(), 0, n_total, seed=0x77AA00 + p, kind=kind)
        engs[p].sync()
        engs[p].pass_begin(n_total)
    rb = engs[0].wpc * 4
    adv = lambda k, n: pbwt_amd.pass_advance_many(engs, [b.data_ptr() + k * rb for b in bufs], n, min(n + 8, n_total - k), opts)
    adv(0, batch)                                            # warm-up batch
    for e in engs:
        e.sync()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = batch
    while k < n_total:
        n = min(step, n_total - k)
        adv(k, n)
        k += n
    for e in engs:
        e.pass_end(opts)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = int(sum(int(e.get_hist(n_total + 1).sum()) for e in engs))
    for e in engs:
        e.close()
    return {"panels": P, "haplotypes_per_panel": M, "sites_timed": sites, "value": P * M * sites / dt, "unit": "site*haps/s over all panels",
            "us_per_site_per_panel": 1e6 * dt / sites / P, "us_per_site_all_panels": 1e6 * dt / sites, "within_reports_hist_total": tot,
            "note": "pbwtamd_pass_advance_many: whole sets of eight panels — the team-persistent chain, one launch per batch, panel p on XCD p; otherwise — every launch of the chain covers all panels (grid.y = panel); consumers per panel"}


def match_dynamic(torch, pbwt_amd, dev, kind, M=1000000, Q=10000, sites=65536, batch=512):
    """configs[4]'s second half: `-matchDynamic` of a Q-haplotype query panel against an M-wide panel (pbwtamd_match_sweep,
    matchSequencesSweep pbwtMatch.c:363-443), host-buffer entry point: packed panels in, records out.  Panel and queries
    are the two parts of ONE synthetic panel (shared founders: matches run for many sites, as with real data).
    Secondary measurement; the record stream at this shape is pinned to the oracle by tests/test_gpu_z_configs.py."""
    assert M % 32 == 0
    full = pbwt_amd.Engine(M + Q, batch_sites=batch, device=dev.index)
    cols = torch.empty((sites, full.wpc), dtype=torch.int32, device=dev)
    full.synth_device(cols.data_ptr(), 0, sites, seed=0x3D, kind=kind)
    full.sync(); full.close()
    ep = pbwt_amd.Engine(M, batch_sites=batch, device=dev.index)
    eq = pbwt_amd.Engine(Q, batch_sites=batch, device=dev.index)
    pc = torch.zeros((sites, ep.wpc), dtype=torch.int32, device=dev); pc[:, : M // 32] = cols[:, : M // 32]
    nqw = (Q + 31) // 32
    qc = torch.zeros((sites, eq.wpc), dtype=torch.int32, device=dev); qc[:, :nqw] = cols[:, M // 32: M // 32 + nqw]
    if Q % 32:
        qc[:, nqw - 1] &= (1 << (Q % 32)) - 1
    del cols
    torch.cuda.synchronize()
    packed = []
    for eng, c in ((ep, pc), (eq, qc)):
        eng.pass_begin(sites); eng.pass_advance(c.data_ptr(), sites, sites, pbwt_amd.OPT_PACK3); eng.pass_end(pbwt_amd.OPT_PACK3)
        packed.append(eng.get_packed())
    del pc, qc
    eq.close()
    pz, qz = packed
    best = None
    for _ in range(1 if sites > 16384 else 2):               # (short runs: the first call pays the allocations)
        t0 = time.perf_counter()
        recs, nom, tot = ep.match_sweep(pz, sites, qz, Q)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    # one rank's share when the QUERIES are sharded over G GPUs (pbwt_amd/queryshard.py; every rank repeats the panel side):
    # the job's wall time at G ranks is the time of Q / G queries here — measured, not modelled, but on this one GPU
    from pbwt_amd import queryshard as qsh
    shard = {}
    for G in (2, 4, 8):
        lo, hi = qsh.plan_ranges(Q, G)[0]
        t0 = time.perf_counter()
        qsh.run_range(ep, pz, sites, qz, Q, lo, hi)
        shard[str(G)] = {"us_per_site": 1e6 * (time.perf_counter() - t0) / sites, "queries_per_rank": hi - lo}
    for G in shard:
        shard[G]["speedup_vs_1"] = 1e6 * best / sites / shard[G]["us_per_site"]
    ep.close()
    alg = (ALG_BYTES_PER_SITEHAP * (M + Q) * sites + 16.0 * len(recs)) / best / 1e9      # panel step + query step + 16 B per report
    return {"haplotypes": M, "queries": Q, "sites": sites, "us_per_site": 1e6 * best / sites, "records": int(len(recs)), "no_match_events": int(nom),
            "value": M * sites / best, "unit": "panel site*haps/s", "achieved_GBps": alg, "frac_of_hbm_peak": alg / HBM_PEAK_GBPS,
            "query_sharding_one_rank_share": shard,
            "note": "host-buffer entry point (packed panels in host memory in, records out), one call over all the sites (set-up and the closing tails included)"}


def position_sharded_selfcheck(torch, pdist, pbwt_amd, dev, rank, world, backend, kind, M=70000, N=264, batch=128):
    """before the big job, on whatever hardware this is: a small panel through the position-sharded engine against the SAME panel through the
    plain engine on rank 0 (histogram, .pbwt bytes, final a/d).  On a multi-GPU node this is the first thing that exercises the peer stores and
    flag barriers over xGMI; a mismatch is reported and the big job is skipped."""
    from pbwt_amd import posshard as ps
    opts = pbwt_amd.OPT_WITH_D | pbwt_amd.OPT_WITHIN_HIST | pbwt_amd.OPT_PACK3 | pbwt_amd.OPT_CHECKSUM
    eng = pbwt_amd.Engine(M, batch_sites=batch, device=dev.index)
    ps.setup(eng, rank, world)
    panel = torch.empty((N, eng.wpc), dtype=torch.int32, device=dev)
    eng.synth_device(panel.data_ptr(), 0, N, seed=0x9051, kind=kind)
    eng.sync()
    ps.run(eng, lambda k: panel.data_ptr() + k * eng.wpc * 4, N, opts)
    hist = ps.reduce_hist(eng.get_hist(N + 1), device=dev if backend == "nccl" else None)
    cs = ps.gather_checksums(eng, 0, N + 1)
    yz = ps.gather_packed(eng)
    a, d = eng.get_state()
    eng.close()
    ok, detail = True, None
    if rank == 0:
        ref = pbwt_amd.Engine(M, batch_sites=batch, device=dev.index)
        ref.pass_begin(N); ref.pass_advance(panel.data_ptr(), N, N, opts); ref.pass_end(opts)
        a0, d0 = ref.get_state()
        ra, rd, _ = ref.get_checksums(0, N + 1)
        same = {"hist": bool(np.array_equal(hist, ref.get_hist(N + 1))), "pack3_bytes": bool(np.array_equal(yz, ref.get_packed())),
                "final_a": bool(np.array_equal(a, a0)), "final_d": bool(np.array_equal(d, d0))}
        if cs is not None:                                   # every site's order-sensitive checksum of (a, d): WHERE the sharded chain first leaves the plain one
            bad = np.nonzero((np.asarray(cs[0]) != ra) | (np.asarray(cs[1]) != rd))[0]
            same["every_site_checksums"] = bool(bad.size == 0)
            if bad.size:
                same["first_differing_site"] = int(bad[0])
        ok = all(v for k, v in same.items() if k != "first_differing_site")
        detail = same
        ref.close()
    ok = bool(pdist.max_over_ranks(0.0 if ok else 1.0, device=dev if backend == "nccl" else None) == 0.0)
    return {"ok": ok, "haplotypes": M, "sites": N, "equal_to_the_plain_engine": detail}


def position_sharded_job(torch, pdist, pbwt_amd, dev, rank, world, backend, opts, kind, M=1000000, sites=1000000, batch=512, step=8192):
    """BASELINE configs[3] / the north star's multi-GPU form: ONE panel of M haplotypes x `sites` sites, the RECURRENCE position-sharded over the
    ranks (pbwt_amd/posshard.py, csrc/pbwt_shard.inc).  Same panel (seed, kind), same pass structure (one untimed warm-up batch, then the rest)
    and same option set as north_star_width(), so `within_reports_hist_total` must equal that object's from the N = 1 run when `sites` is the same."""
    from pbwt_amd import posshard as ps
    red = dev if backend == "nccl" else None
    sites = (sites // batch) * batch
    eng = pbwt_amd.Engine(M, batch_sites=batch, device=dev.index)
    ps.setup(eng, rank, world)
    n_total = sites + batch
    panel = torch.empty((n_total, eng.wpc), dtype=torch.int32, device=dev)      # replicated: every rank holds the panel's columns
    eng.synth_device(panel.data_ptr(), 0, n_total, seed=0x1A2B3C, kind=kind)
    eng.sync()
    col = lambda k: panel.data_ptr() + k * eng.wpc * 4
    eng.pass_begin(n_total)
    eng.pass_advance(col(0), batch, batch + 8, opts)
    eng.sync()
    ms0, n0 = eng.chain_timing(); s0 = eng.chain_sites()
    pdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = batch
    while k < n_total:
        n = min(step, n_total - k)
        eng.pass_advance(col(k), n, min(n + 8, n_total - k), opts)
        k += n
    eng.pass_end(opts)
    hist = ps.reduce_hist(eng.get_hist(n_total + 1), device=red)              # the one collective of the mode, inside the timed region
    pdist.barrier(); torch.cuda.synchronize()
    dt = pdist.max_over_ranks(time.perf_counter() - t0, device=red)
    ms1, n1 = eng.chain_timing(); s1 = eng.chain_sites()
    us = 1e3 * (ms1 - ms0) / max(n1 - n0, 1); spl = (s1 - s0) / max(n1 - n0, 1)
    lo, hi = eng.shard_range(rank)
    ach = ALG_BYTES_PER_SITEHAP * (hi - lo) * spl / (us * 1e-6) / 1e9
    ndev = len({int(x) for x in _gather_ints(pdist, torch, dev.index if dev.index is not None else 0, red)})
    st = eng.shard_stats()                                      # what rank 0's chain spent waiting for its peers (rows of a round, flag barriers), whole engine life (warm-up batch included)
    rounds = max(st["row_waits"], 1)
    eng.close()
    del panel
    return {"haplotypes": M, "sites_timed": sites, "n_ranks": world, "devices": ndev, "backend": backend, "seconds": dt, "value": M * sites / dt, "unit": "site*haps/s",
            "us_per_site": 1e6 * dt / sites, "scaling": "strong", "within_reports_hist_total": int(hist.sum()),
            "whole_job_achieved_GBps": ALG_BYTES_PER_SITEHAP * M * sites / dt / 1e9,
            "whole_job_frac_of_hbm_peak": ALG_BYTES_PER_SITEHAP * M * sites / dt / 1e9 / (HBM_PEAK_GBPS * max(ndev, 1)),
            "validated_on": ("%d devices" % ndev) if ndev > 1 else "ranks sharing ONE device (peer stores never left the GPU): correctness only, not a scaling result",
            "roofline": {"bound": "hbm", "kernel": "sharded skeleton chain: skel_hist_kernel + skel_k2s_kernel + skel_rank_shard_kernel + shard_xbar_kernel, 4 launches per 8 sites",
                         "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None, "us_per_launch": us, "sites_per_launch": spl,
                         "positions_of_rank0": [lo, hi], "note": "rank 0's chain over its own range of positions (launch gaps and peer waits included)"},
            "exchange": "per round of 8 sites: one row of 256 (count, carry) per rank + the scatter as peer stores (hipIpc / xGMI), 2 flag barriers; consumers sharded "
                        "by site inside every batch (bulk pulls); one all-reduce of the histogram at the end",
            "exchange_wait_rank0": {"row_wait_us_per_round": st["row_wait_us"] / rounds, "barrier_wait_us_per_round": st["barrier_us"] / rounds, "rounds": st["row_waits"], "barriers": st["barriers"],
                                    "chain_us_per_round": us * (8.0 / max(spl, 1e-9)),
                                    "note": "time rank 0's chain kernels spent waiting for the peers' rows and inside the flag barriers (device wall clock, pbwtamd_shard_stats), "
                                            "beside the chain's launch-to-launch time per round: the exchange's share of a round"}}


def ipcprobe_report(world, spread):
    """tools/ipcprobe (hipIpc peer scatter + flag barrier + read-back between `world` processes, the sharded chain's own traffic pattern on plain hipMalloc rings) across
    the same devices, BEFORE the sharded engine runs: its "mismatches" are words a peer wrote that the owner did not see — a platform visibility failure told apart from a
    defect of the chain.  Best effort: None when the probe is not built or does not come back."""
    import subprocess
    probe = os.path.join(ROOT, "tools", "ipcprobe")
    if not os.path.exists(probe):
        return {"error": "tools/ipcprobe not built"}
    try:
        pr = subprocess.run([probe, str(world), "200", "0"], capture_output=True, text=True, timeout=120,
                            env=dict(os.environ, IPCPROBE_SPREAD="1" if spread else "0", HSA_ENABLE_IPC_MODE_LEGACY="0"))
        lines = [ln for ln in pr.stdout.splitlines() if "mismatches" in ln or "staggered barrier" in ln or ln.startswith("ipcprobe:")]
        stale = sum(int(ln.split("mismatches")[1].split(",")[0]) for ln in lines if "mismatches" in ln)
        return {"rc": pr.returncode, "stale_words": stale, "processes": world, "one_process_per_device": bool(spread), "lines": lines[:8]}
    except Exception as ex:
        return {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}


def _gather_ints(pdist, torch, value, device):
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [value]
    t = torch.tensor([int(value)], dtype=torch.int64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def host_entry_points(torch, pbwt_amd, panel, M, sites=16384, batch=512):
    """PCIe-inclusive rates of the host-buffer entry points on the first `sites` columns of the same panel (reported
    next to the headline, never part of `value`): pbwtamd_build (columns in host memory -> .pbwt bytes) and the read side
    pbwtamd_max_within in -stats mode (packed panel up, decode, ForwardsReadAD chain, sweep, histogram back)"""
    bits = panel[:sites].cpu().numpy().view(np.uint32)
    eng = pbwt_amd.Engine(M, batch_sites=batch)
    eng.build(bits[:1024], with_d=False)                                 # warm-up (allocations)
    t0 = time.perf_counter(); b = eng.build(bits, with_d=False); t1 = time.perf_counter()
    eng.max_within(b["yz"], sites, mode="hist"); t2 = time.perf_counter()
    eng.max_within(b["yz"], sites, mode="hist"); t3 = time.perf_counter()
    eng.close()
    return {"sites": sites, "build_site_haps_per_s": M * sites / (t1 - t0), "read_maxwithin_stats_site_haps_per_s": M * sites / (t3 - t2),
            "note": "caller buffers in ordinary host memory (columns in and .pbwt bytes out through the engine's pinned staging buffers), transfers included, first call at this size"}


def match_records(torch, pbwt_amd, dev, kind, with_ref=True):
    """The reference's DEFAULT -maxWithin sink: every maximal match as a record / a MATCH text line (pbwtMatch.c:46-49,133-134), not the -stats histogram the
    headline uses.  configs[0] (2 000 x 20 000) and configs[1] (10 000 x 100 000), BASELINE.md section 4's last bullet: R, seconds and
    (16.125 M N + 16 R) / t for (a) pbwtamd_max_within with the record sink (packed panel in host memory in, records out), (b) the CLI
    `pbwt -read f -maxWithin > /dev/null` (file read, device pass, text formatting), beside (c) the reference's own pbwtLongMatches writing the same text
    to /dev/null on one host core (oracle/_ref; skipped when it did not travel).  The text itself is pinned by tests/test_cli.py::test_config0_full_size..."""
    import subprocess
    import tempfile
    out = {}
    cli = os.path.join(ROOT, "pbwt_amd", "pbwt")
    for name, M, N in (("configs[0]", 2000, 20000), ("configs[1]", 10000, 100000)):
        eng = pbwt_amd.Engine(M, batch_sites=512)
        buf = torch.zeros((N, eng.wpc), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        eng.synth_device(buf.data_ptr(), 0, N, seed=0x5EED + M, kind=kind)
        eng.sync()
        bits = buf.cpu().numpy().view(np.uint32)
        yz = eng.build(bits, with_d=False)["yz"]
        eng.max_within(yz, N, mode="hist")                               # warm-up (allocations of the read side)
        t0 = time.perf_counter(); rec = eng.max_within(yz, N, mode="records"); t1 = time.perf_counter()
        R = int(len(rec))
        alg = ALG_BYTES_PER_SITEHAP * M * N + 16.0 * R
        o = {"haplotypes": M, "sites": N, "records": R, "record_sink_seconds": t1 - t0, "record_sink_GBps": alg / (t1 - t0) / 1e9,
             "record_sink_site_haps_per_s": M * N / (t1 - t0)}
        del rec
        eng.close()
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "p.pbwt")
            ident = np.arange(M, dtype="<i4").tobytes()
            open(f, "wb").write(b"PBW3" + np.array([M, N], "<i4").tobytes() + ident + ident + np.array([len(yz)], "<i8").tobytes() + b"    " + np.asarray(yz, np.uint8).tobytes())
            if os.path.exists(cli):
                t0 = time.perf_counter()
                r = subprocess.run([cli, "-read", f, "-maxWithin"], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
                t1 = time.perf_counter()
                if r.returncode == 0:
                    o["cli_text_to_devnull_seconds"] = t1 - t0
                    o["cli_text_GBps"] = alg / (t1 - t0) / 1e9
                else:
                    o["cli_error"] = r.stderr.decode(errors="replace")[-200:]
            if with_ref:
                try:
                    import oracle
                    if oracle.ref() is not None:
                        t0 = time.perf_counter(); oracle.ref_max_within_file(yz, M, N, "/dev/null"); t1 = time.perf_counter()
                        o["reference_text_to_devnull_seconds_1core"] = t1 - t0
                        o["speedup_record_sink_vs_reference"] = (t1 - t0) / o["record_sink_seconds"]
                except Exception as ex:
                    o["reference_error"] = "%s: %s" % (type(ex).__name__, ex)
        out[name] = o
    out["note"] = ("-maxWithin with the RECORD sink (the reference's default: one MATCH line per maximal match), host-buffer entry point, transfers and set-up included; "
                   "GBps = (16.125 B x M x N + 16 B x R) / seconds (SURVEY 8(d)); the CLI line also reads the .pbwt file, starts a process and formats the text")
    return out


def run_siteblock(args, torch, pdist, pbwt_amd, dev, rank, world):
    """--mode siteblock: one panel, sharded by site blocks across the ranks (strong scaling)"""
    from pbwt_amd import siteblock as sb
    M, S, K, Wm = args.haps, args.sites_per_step, args.steps, args.warmup
    N = K * S
    eng = pbwt_amd.Engine(M, batch_sites=args.batch, device=dev.index)
    panel = torch.empty((N, eng.wpc), dtype=torch.int32, device=dev)          # every rank holds the panel's columns
    eng.synth_device(panel.data_ptr(), 0, N, seed=0x5EED0001, kind=args.kind)
    eng.sync()
    opts = pbwt_amd.OPT_WITH_D | (0 if args.no_within else pbwt_amd.OPT_WITHIN_HIST) | (0 if args.no_pack3 else pbwt_amd.OPT_PACK3)
    col = lambda k: panel.data_ptr() + k * eng.wpc * 4
    # warm-up = the calibration of rho = t_chain / t_full on this GPU (untimed), agreed between the ranks
    rho = sb.calibrate_rho(eng, col, N, opts, args.batch, nbatches=max(2, min(8, Wm * S // args.batch)))
    rho = pdist.max_over_ranks(rho, device=dev)
    blocks = sb.plan_blocks(N, world, rho, align=args.batch)
    pdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    sb.run_block(eng, col, N, blocks[rank], opts, is_last=(rank == world - 1), step=S)
    hist = sb.reduce_hist(eng.get_hist(N + 1), device=dev)                     # the one collective of the mode
    pdist.barrier(); torch.cuda.synchronize()
    dt = pdist.max_over_ranks(time.perf_counter() - t0, device=dev)
    ms, nl = eng.chain_timing(); sites = eng.chain_sites()
    us = 1e3 * ms / max(nl, 1); spl = sites / max(nl, 1)
    ach = ALG_BYTES_PER_SITEHAP * M * spl / (us * 1e-6) / 1e9
    out = {"metric": "sites*haplotypes/sec PBWT build + maxWithin", "value": M * N / dt, "unit": "site*haps/s",
           "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
           "config": {"workload": "configs[2]: ONE panel of %d haplotypes x %d sites, build (ForwardsAD + pack3) + maxWithin (hist sink), "
                                  "site-block sharded over %d ranks" % (M, N, world),
                      "haplotypes": M, "sites_per_step": S, "sites_timed": N, "device_batch_sites": args.batch, "mode": "siteblock",
                      "rho_t_chain_over_t_full": rho, "blocks": blocks, "collective": "all-reduce of the histogram (RCCL), once"},
           "roofline": {"bound": "hbm", "kernel": "skeleton chain: skel_hist_kernel + skel_k2_kernel + skel_rank_kernel, 3 launches per 8 sites",
                        "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                        "us_per_launch": us, "sites_per_launch": spl, "note": "rank 0's chain (prefix + block)"},
           "within_reports_hist_total": int(hist.sum())}
    if rank == 0:
        print(json.dumps(out))
    pdist.finish()


def run_posshard(args, torch, pdist, pbwt_amd, dev, rank, world):
    """--mode posshard: one panel, the recurrence position-sharded across the ranks (strong scaling; BASELINE configs[3])"""
    from pbwt_amd import posshard as ps
    M, S, K, Wm = args.haps, args.sites_per_step, args.steps, args.warmup
    N = (K + Wm) * S
    eng = pbwt_amd.Engine(M, batch_sites=args.batch, device=dev.index)
    ps.setup(eng, rank, world)
    if N * eng.wpc * 4 > 120e9:
        raise SystemExit("bench.py --mode posshard keeps the panel's bit columns resident: %d sites x %d haplotypes = %.0f GB per rank; use fewer --steps"
                         % (N, M, N * eng.wpc * 4 / 1e9))
    panel = torch.empty((N, eng.wpc), dtype=torch.int32, device=dev)          # replicated: every rank holds the panel's columns
    eng.synth_device(panel.data_ptr(), 0, N, seed=0x5EED0001, kind=args.kind)
    eng.sync()
    opts = pbwt_amd.OPT_WITH_D | (0 if args.no_within else pbwt_amd.OPT_WITHIN_HIST) | (0 if args.no_pack3 else pbwt_amd.OPT_PACK3)
    col = lambda k: panel.data_ptr() + k * eng.wpc * 4
    eng.pass_begin(N)
    for i in range(Wm):                                       # untimed warm-up steps of the same pass
        eng.pass_advance(col(i * S), S, min(S + 8, N - i * S), opts)
    eng.sync()
    ms0, n0 = eng.chain_timing(); s0 = eng.chain_sites()
    pdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Wm, Wm + K):
        eng.pass_advance(col(i * S), S, min(S + 8, N - i * S), opts)
    eng.pass_end(opts)
    hist = ps.reduce_hist(eng.get_hist(N + 1), device=dev if args.backend == "nccl" else None)   # the one collective of the mode
    pdist.barrier(); torch.cuda.synchronize()
    dt = pdist.max_over_ranks(time.perf_counter() - t0, device=dev if args.backend == "nccl" else None)
    ms1, n1 = eng.chain_timing(); s1 = eng.chain_sites()
    us = 1e3 * (ms1 - ms0) / max(n1 - n0, 1); spl = (s1 - s0) / max(n1 - n0, 1)
    lo, hi = eng.shard_range(rank)
    ach = ALG_BYTES_PER_SITEHAP * (hi - lo) * spl / (us * 1e-6) / 1e9
    out = {"metric": "sites*haplotypes/sec PBWT build + maxWithin", "value": M * K * S / dt, "unit": "site*haps/s",
           "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
           "config": {"workload": "configs[3]: ONE panel of %d haplotypes x %d sites, build (ForwardsAD + pack3) + maxWithin (hist sink), "
                                  "the recurrence position-sharded over %d ranks" % (M, K * S, world),
                      "haplotypes": M, "sites_per_step": S, "sites_timed": K * S, "device_batch_sites": args.batch, "mode": "posshard",
                      "positions_of_rank0": [lo, hi], "backend": args.backend,
                      "exchange": "per round of 8 sites: one row of 256 (count, carry) per rank + the scatter as peer stores (hipIpc / xGMI), 2 flag barriers; "
                                  "consumers sharded by site inside every batch (bulk pulls); one all-reduce of the histogram at the end"},
           "roofline": {"bound": "hbm", "kernel": "sharded skeleton chain: skel_hist_kernel + skel_k2s_kernel + skel_rank_shard_kernel + shard_xbar_kernel, 4 launches per 8 sites",
                        "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                        "us_per_launch": us, "sites_per_launch": spl, "note": "rank 0's chain over its own range of positions (launch gaps and peer waits included)"},
           "whole_job_achieved_GBps": ALG_BYTES_PER_SITEHAP * M * K * S / dt / 1e9,
           "whole_job_frac_of_hbm_peak": ALG_BYTES_PER_SITEHAP * M * K * S / dt / 1e9 / (HBM_PEAK_GBPS * world),
           "within_reports_hist_total": int(hist.sum())}
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    pdist.finish()


def main():
    args = parse()
    # before torch initialises HIP: the launching thread and the runtime's helper threads on the physical cores of the GPU's NUMA node (pbwt_amd/pin.py; the
    # "slow mode" — one fresh process in five 10 % slower — follows their placement).  PBWTAMD_PIN=0: off
    from pbwt_amd.pin import pin_for_gpu
    pinned_cpus = pin_for_gpu(int(os.environ.get("LOCAL_RANK", "0")))
    import torch
    from pbwt_amd import dist as pdist
    rank, local, world = pdist.env_world()
    if args.backend == "nccl":
        dev = torch.device("cuda", local if world > 1 else 0)
        torch.cuda.set_device(dev)
        pdist.init("nccl", device_id=dev)  # backend "nccl" is RCCL on ROCm; only barrier + max-reduce (and the modes' one all-reduce) use it
    else:                                  # gloo: several ranks may share one GPU (how the one-GPU test box runs the sharded modes)
        dev = torch.device("cuda", local if (world > 1 and torch.cuda.device_count() > local) else 0)
        torch.cuda.set_device(dev)
        pdist.init(args.backend)
    import pbwt_amd
    if args.stream_panel and args.with_queries:
        out = configs4_streamed_with_queries(torch, pbwt_amd, dev, args.kind, sites=args.ns_sites, pack3=not args.ns_no_pack3)
        print(json.dumps(out), flush=True)
        return pdist.finish()
    if args.stream_panel:
        o = pbwt_amd.OPT_WITH_D | pbwt_amd.OPT_WITHIN_HIST | (0 if args.ns_no_pack3 else pbwt_amd.OPT_PACK3)
        out = north_star_streamed(torch, pbwt_amd, dev, o, args.kind, sites=args.ns_sites)
        print(json.dumps(out), flush=True)
        return pdist.finish()
    if args.mode == "siteblock":
        return run_siteblock(args, torch, pdist, pbwt_amd, dev, rank, world)
    if args.mode == "posshard":
        return run_posshard(args, torch, pdist, pbwt_amd, dev, rank, world)

    M, S, K, Wm = args.haps, args.sites_per_step, args.steps, args.warmup
    if S % 8:
        raise SystemExit("--sites-per-step must be a multiple of 8 (the chain's radix step spans 8 sites)")
    n_total = K * S                                         # the panel: the timed pass covers ALL of it, site 0 to site N
    stream = torch.cuda.current_stream().cuda_stream
    if args.panels > 1:                                     # the fused launches need every panel's engine on ONE (non-null) stream
        shared = torch.cuda.Stream(device=dev)
        stream = shared.cuda_stream
    eng = pbwt_amd.Engine(M, batch_sites=args.batch, device=dev.index, stream=None if args.own_stream else stream)
    wpc = eng.wpc
    # the panel, resident in HBM before the timed region (bit-packed, original haplotype order)
    panel = torch.empty((n_total, wpc), dtype=torch.int32, device=dev)
    unit = pdist.units_for_rank(world, rank, world)[0]      # one independent panel per rank (weak scaling)
    eng.synth_device(panel.data_ptr(), 0, n_total, seed=pdist.panel_seed(0x5EED0001, unit), kind=args.kind)
    eng.sync()
    extra = []                                              # --panels P: P-1 more independent panels (chromosomes) on this GPU, same stream
    for pi in range(1, args.panels):
        e2 = pbwt_amd.Engine(M, batch_sites=args.batch, device=dev.index, stream=None if args.own_stream else stream)
        p2 = torch.empty((n_total, wpc), dtype=torch.int32, device=dev)
        e2.synth_device(p2.data_ptr(), 0, n_total, seed=pdist.panel_seed(0x5EED0001, world * pi + unit), kind=args.kind)
        e2.sync()
        extra.append((e2, p2))
    if extra and args.own_stream:
        raise SystemExit("--panels needs the engines on one stream (drop --own-stream)")
    opts = pbwt_amd.OPT_WITH_D
    if not args.no_within:
        opts |= pbwt_amd.OPT_WITHIN_HIST
    if not args.no_pack3:
        opts |= pbwt_amd.OPT_PACK3
    row_bytes = wpc * 4
    engines = [eng] + [e2 for e2, _ in extra]
    panels = [panel] + [p2 for _, p2 in extra]

    def step(i):
        k = i * S
        avail = min(S + 8, n_total - k)    # look-ahead columns: the chain's radix step spans 8 sites (2 for the two-site path)
        if extra:                          # ONE launch of the chain covers every panel (pbwtamd_pass_advance_many)
            pbwt_amd.pass_advance_many(engines, [p.data_ptr() + k * row_bytes for p in panels], S, avail, opts)
        else:
            eng.pass_advance(panel.data_ptr() + k * row_bytes, S, avail, opts)

    # warm-up: W untimed steps as a pass of their own over the panel's first sites (allocations, code objects, clocks)
    ms_w, n_w, w_left = 0.0, 0, Wm
    while w_left > 0:
        nw = min(w_left, K)
        for e in engines:
            e.pass_begin(n_total)
        for i in range(nw):
            step(i)
        for e in engines:
            e.pass_stop()
        m_, n_ = eng.chain_timing(); ms_w += m_; n_w += n_
        w_left -= nw
    for e in engines:                      # the timed pass: cursor creation (pbwtCursorCreate, pbwtCore.c:420-445) is set-up, outside the timed region
        e.pass_begin(n_total)
    eng.sync()

    def barrier():
        pdist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    eng.pass_end(opts)                     # includes the k == N sweep; synchronises the stream
    for e2, _ in extra:
        e2.pass_end(opts)
    barrier()
    dt = time.perf_counter() - t0
    ms_all, n_all = eng.chain_timing()
    dt = pdist.max_over_ranks(dt, device=dev)

    chain_ms, chain_n = ms_all, n_all
    chain_sites = eng.chain_sites()
    sites_per_launch = chain_sites / max(chain_n, 1)
    us_per_launch = 1e3 * chain_ms / max(chain_n, 1)
    alg_bytes_per_launch = ALG_BYTES_PER_SITEHAP * M * sites_per_launch
    achieved = alg_bytes_per_launch / (us_per_launch * 1e-6) / 1e9
    hist = eng.get_hist(n_total + 1)
    first = panel[:min(args.cpu_sites, n_total)].cpu().numpy().view(np.uint32) if (rank == 0 and world == 1 and not args.no_cpu) else None
    hep = host_entry_points(torch, pbwt_amd, panel, M) if (rank == 0 and world == 1 and not args.no_1m and n_total >= 16384) else None
    traffic, traffic_src = None, None
    try:                                   # HBM-side bytes per launch measured with rocprofv3 PMC (separate run)
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(str(M))
        if tj and tj.get("with_d"):
            traffic, traffic_src = tj["bytes_per_launch"], tj["source"]
    except Exception:
        pass
    out = {
        "metric": "sites*haplotypes/sec PBWT build + maxWithin",
        "value": world * args.panels * K * S * M / dt,
        "unit": "site*haps/s",
        "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": 1e3 * dt / K,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": "configs[2]%s: %d haplotypes x %d sites, ONE pass from site 0 to site N: build (ForwardsAD + pack3) + maxWithin (hist sink, k == N sweep included)"
                               % (" itself" if (M == 100000 and K * S == 1000000) else " width", M, K * S),
                   "haplotypes": M, "sites_per_step": S, "sites_timed": K * S, "device_batch_sites": args.batch,
                   "panel": "founder-mosaic" if args.kind == 0 else "iid", "within": not args.no_within,
                   "pack3": not args.no_pack3, "units_per_rank": "independent panel per rank", "panels_per_gpu": args.panels},
        "roofline": {"bound": "hbm", "kernel": ("one-launch round: skel_onepass_kernel, 1 launch per 8 sites (key totals per batch: skel_totals_kernel)" if sites_per_launch > 7.5 else
                                "skeleton chain: skel_hist_kernel + skel_k2_kernel + skel_rank_kernel, 3 launches per 8 sites" if sites_per_launch > 2.5 else
                                "step2_kernel<WITH_D> (two sites per launch)" if sites_per_launch > 1.5 else "step1_kernel<WITH_D,GATHER>"), "achieved": achieved, "peak": HBM_PEAK_GBPS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                     "alg_bytes_per_launch": alg_bytes_per_launch, "us_per_launch": us_per_launch,
                     "launches": int(chain_n), "sites_per_launch": sites_per_launch,
                     "latency_bound": latency_bound(alg_bytes_per_launch, us_per_launch),
                     "latency_bound_frac": latency_bound(alg_bytes_per_launch, us_per_launch)["latency_bound_frac"],
                     "note": "one launch = sites_per_launch sites; duration = HIP-event time of the dependent launch chain / launches, i.e. including launch gaps"},
        "within_reports_hist_total": int(hist.sum()),
        "verified_hist_total": verified_total(M, K * S, args, unit, int(hist.sum())),
        "launch_us_level_warmup": 1e3 * ms_w / max(n_w, 1),      # the chain's us per launch over the warm-up steps (the slow mode shows here first: 4.3 against 3.9 at three launches per round)
        "pinned_cpus": (len(pinned_cpus) if pinned_cpus else 0),
    }
    if hep:
        out["host_entry_points"] = hep
    if rank == 0 and world == 1 and not args.no_1m and args.panels == 1:
        # ONE fixed point, P = 8 (eight chromosomes of a cohort side by side; from six panels on pbwtamd_pass_advance_many runs the team-persistent chain,
        # panel p on XCD p — round 5).  Not a maximum over tried configurations.
        try:                                                # (a secondary object must not take the line with it)
            out["many_panels"] = many_panels(torch, pbwt_amd, dev, opts, args.kind, M, P=MANY_PANELS_P)
            out["many_panels"]["speedup_vs_one_panel"] = out["many_panels"]["value"] / out["value"]
        except Exception as ex:
            out["many_panels"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if rank == 0 and world == 1 and not args.no_1m:
        del panel
        torch.cuda.empty_cache()
        free_b = torch.cuda.mem_get_info(dev)[0]
        ns_sites = min(args.ns_sites, int((free_b - 24e9) // 125000) // 512 * 512)     # leave room for the engine's rings (~10 GB) and the query sweep
        try:
            out["north_star_width"] = north_star_width(torch, pbwt_amd, dev, opts, args.kind, sites=max(ns_sites, 4096))
        except Exception as ex:
            out["north_star_width"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        torch.cuda.empty_cache()
        try:
            out["match_dynamic"] = match_dynamic(torch, pbwt_amd, dev, args.kind)
        except Exception as ex:
            out["match_dynamic"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if rank == 0 and world == 1 and not args.no_1m:
        try:
            out["match_records"] = match_records(torch, pbwt_amd, dev, args.kind, with_ref=not args.no_cpu)
        except Exception as ex:
            out["match_records"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args, first)
    if world > 1 and not args.no_posshard and args.panels == 1:
        # N > 1: after the replicas measurement (independent panels, `value`), the SAME process group runs the north star's multi-GPU form — ONE
        # 1 M-haplotype panel with the recurrence position-sharded over the ranks (BASELINE configs[3]) — and attaches it as `position_sharded`.
        # Never `value`.  It has only ever run with the ranks sharing one GPU, so it is fenced: a small self-check against the plain engine first,
        # exceptions caught, and a watchdog that prints the line without it if it does not come back.
        import copy
        import threading
        done = threading.Event()
        fallback = copy.deepcopy(out)                       # the line as it stands now: the watchdog never touches `out` while the main thread fills it in

        def watchdog():
            if done.wait(args.ps_timeout):
                return
            code = 3                                         # the attached job hung (a collective or a peer wait that never came back): not a clean exit
            try:
                if rank == 0:
                    fallback["position_sharded"] = {"error": "gave up after %.0f s (--ps-timeout)" % args.ps_timeout}
                    print(json.dumps(fallback), flush=True)
                    code = 0                                 # rank 0 printed the replicas line (a valid measurement, with the failure of the attachment in it)
            finally:
                os._exit(code)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            del panel
            torch.cuda.empty_cache()
            rpd = max(1, world // max(1, min(world, torch.cuda.device_count())))       # ranks per device (gloo on the one-GPU box: all of them)
            free_b = torch.cuda.mem_get_info(dev)[0] / rpd
            ps_sites = args.ps_sites if args.ps_sites else args.ns_sites
            ps_sites = min(ps_sites, int((free_b - 24e9 / rpd) // 125000) // 512 * 512)
            red = dev if args.backend == "nccl" else None
            ps_sites = int(-pdist.max_over_ranks(-float(ps_sites), device=red))         # the same job on every rank: the smallest
            probe = ipcprobe_report(world, spread=torch.cuda.device_count() >= world) if rank == 0 else None
            pdist.barrier()
            check = position_sharded_selfcheck(torch, pdist, pbwt_amd, dev, rank, world, args.backend, args.kind)
            if not check["ok"]:
                out["position_sharded"] = {"error": "self-check failed: the sharded engine differs from the plain engine on a 70 000 x 264 panel (equal_to_the_plain_engine says where)", "selfcheck": check}
            elif ps_sites < 512:
                out["position_sharded"] = {"error": "not enough free device memory for the replicated panel", "selfcheck": check}
            else:
                out["position_sharded"] = position_sharded_job(torch, pdist, pbwt_amd, dev, rank, world, args.backend, opts, args.kind, sites=ps_sites)
                out["position_sharded"]["selfcheck"] = check
            out["position_sharded"]["ipcprobe"] = probe
        except Exception as ex:                              # the secondary object must not cost the line
            out["position_sharded"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:500])}
        done.set()
    if rank == 0:
        print(json.dumps(out), flush=True)
    pdist.finish()


if __name__ == "__main__":
    main()
