"""multi-GPU plumbing for bench.py: one process per GPU over torch.distributed (RCCL on ROCm).

The PBWT site recurrence does not shard across GPUs without a per-site exchange that costs more
than the step itself, so ranks take INDEPENDENT units (panels / chromosomes): no
data-path collective, only a barrier around the timed region and a max-reduction of the elapsed
time.  These helpers are backend-agnostic so the N>1 logic is testable on CPU with gloo."""
import os


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment (1 process = 1 GPU)"""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, device_id=None):
    """initialise the default process group when WORLD_SIZE > 1; returns (rank, world)"""
    rank, _local, world = env_world()
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            kw = {}
            if device_id is not None:
                kw["device_id"] = device_id
            dist.init_process_group(backend, **kw)
    return rank, world


def units_for_rank(n_units, rank, world):
    """contiguous block of independent units (panels) for this rank: weak scaling gives each rank
    n_units/world of them; with n_units == world every rank takes exactly one"""
    per = n_units // world
    extra = n_units % world
    lo = rank * per + min(rank, extra)
    return list(range(lo, lo + per + (1 if rank < extra else 0)))


def panel_seed(base_seed, unit):
    """distinct synthetic panel per unit"""
    return base_seed + unit


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device=None):
    """max of a python float over all ranks (the job's elapsed time)"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def finish():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
