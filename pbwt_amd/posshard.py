"""Position sharding of ONE panel across the GPUs of a node — the host side (BASELINE configs[3]).

The device side is `pbwt_amd/csrc/pbwt_shard.inc` behind the C ABI (`pbwtamd_shard_init / _connect`, then the ordinary
`pbwtamd_pass_*` calls): rank g owns a contiguous range of positions of the sorted order and runs the chain of
pbwtCursorForwardsAD (pbwtCore.c:485-508) over it; per round of 8 sites the ranks exchange one row of 256 (count, carry)
pairs and store every element of the new order straight into its owner's memory (peer stores through hipIpc mappings:
xGMI on a multi-GPU node).  What this module does:

  setup(eng)              all-gather of the ranks' handle blobs (torch.distributed, any backend) + connect
  plan_rounds(nr, world)  which rounds of a batch a rank consumes (fill + maxWithin sweep + pack3 + checksums)
  reduce_hist / gather_checksums / gather_packed
                          the once-per-job collectives: histograms add, checksums add (a rank's are zero for the sites
                          it did not consume), pack3 blocks interleave by (batch, rank) into PBWT.yz
  merge_packed            the interleaving itself (pure numpy: tested on the CPU)

One process per GPU (torchrun); several ranks may share one device — that is how the GPU test box (one GPU) runs it."""
import numpy as np


def tile_bounds(n_tiles, world):
    """first tile of every rank (and the end): the split pbwtamd_shard_init makes"""
    return [g * n_tiles // world for g in range(world + 1)]


def plan_rounds(nr, world):
    """rounds [lo, hi) of a batch of nr rounds (8 sites each) that each rank consumes"""
    return [(g * nr // world, (g + 1) * nr // world) for g in range(world)]


def setup(eng, rank=None, world=None):
    """make `eng` one rank of a position-sharded panel: every rank calls this with an engine of the same M and batch"""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    import os, sys
    # hipIpc export / open of the rings needs the dmabuf IPC mode on this driver stack (the legacy mode fails with hipIpcGetMemHandle: invalid
    # argument); the variable is read when the HSA runtime initialises, i.e. before the first HIP call of the process — bench.py and the test
    # workers export it themselves, a caller embedding this module has to as well
    if world > 1 and os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") != "0":
        print("[posshard %d] warning: HSA_ENABLE_IPC_MODE_LEGACY is not 0 in this process; hipIpc handles may not export (set it before HIP initialises)" % rank, file=sys.stderr, flush=True)
    tr = (lambda m: print("[posshard %d] %s" % (rank, m), file=sys.stderr, flush=True)) if os.environ.get("PBWTAMD_SHARD_TRACE") else (lambda m: None)
    blob = eng.shard_init(rank, world)
    if world > 1:
        blobs = [None] * world
        tr("all-gather of the handle blobs")
        dist.all_gather_object(blobs, blob)
        tr("connect")
        eng.shard_connect(blobs)
        tr("connected")
    return rank, world


def reduce_hist(hist, device=None):
    from .siteblock import reduce_hist as rh
    return rh(hist, device=device)


def gather_checksums(eng, k_first, n, dst=0):
    """sum over the ranks (mod 2^64) of the per-site checksums: every site was consumed by exactly one rank"""
    import torch.distributed as dist
    mine = np.stack(eng.get_checksums(k_first, n))
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return mine
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(mine, parts, dst=dst)
    if parts is None:
        return None
    out = np.zeros_like(mine)
    for p in parts:
        out += p                                            # uint64: wraps like the device's atomicAdd
    return out


def merge_packed(parts):
    """parts: per rank (site0[], nsites[], byte_end[], bytes) as `Engine.shard_blocks()` + `Engine.get_packed()` give them.
    Returns the panel's yz: the blocks of all ranks in site order (pack3 runs never span columns, pbwtCore.c:254-267)."""
    blocks = []
    for s0, ns, be, yz in parts:
        start = 0
        for i in range(len(s0)):
            blocks.append((int(s0[i]), int(ns[i]), yz[start:int(be[i])]))
            start = int(be[i])
        if start != len(yz):
            raise ValueError("merge_packed: %d bytes beyond the last block" % (len(yz) - start))
    blocks.sort(key=lambda b: b[0])
    k = blocks[0][0] if blocks else 0
    for s0, ns, _ in blocks:
        if s0 != k:
            raise ValueError("merge_packed: the blocks do not tile the sites (gap or overlap at site %d)" % k)
        k += ns
    return np.concatenate([b[2] for b in blocks]) if blocks else np.zeros(0, np.uint8)


def gather_packed(eng, dst=0):
    """the panel's pack3 bytes on rank `dst` (None elsewhere)"""
    import torch.distributed as dist
    s0, ns, be = eng.shard_blocks()
    mine = (s0, ns, be, eng.get_packed())
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return merge_packed([mine])
    parts = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(mine, parts, dst=dst)
    return None if parts is None else merge_packed(parts)


def run(eng, col_ptr, N, opts, step=8192, lookahead=8):
    """the whole pass on this rank: every rank is handed the same columns (col_ptr(k) -> device address of bit column k)"""
    eng.pass_begin(N)
    k = 0
    while k < N:
        n = min(step, N - k)
        eng.pass_advance(col_ptr(k), n, min(n + lookahead, N - k), opts)
        k += n
    eng.pass_end(opts)
