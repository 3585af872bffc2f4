"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU).

The headline configuration (5-keyframe / 2 k-landmark windows) shards by WINDOW: windows are independent problems, so
ranks take disjoint slices and there is NO data-path collective ("replicas only").  torch.distributed is
used for the barrier around the timed region, the max-over-ranks time and for gathering per-window reports.
"""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(n_items, rank, world):
    """Contiguous, balanced slice [lo, hi) of n_items for `rank`: sizes differ by at most one, union = everything."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def init(backend=None, device_index=None):
    """Initialise the default process group from the torchrun environment; returns the torch.distributed module or
    None for a single process."""
    rank, local_rank, world = env_rank_world()
    if world == 1:
        return None
    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local_rank if device_index is None else device_index)
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return dist


def max_over_ranks(dist, value, device="cpu"):
    if dist is None:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_objects(dist, obj):
    """List with every rank's object (rank order) on all ranks."""
    if dist is None:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def init_shard_comm(dist, ctx):
    """Give `ctx` the RCCL communicator of a landmark-sharded solve (limo_ctx_comm_init): rank 0 creates the unique id,
    torch.distributed broadcasts its bytes, every rank joins.  Single process: a one-rank communicator."""
    rank, _, world = env_rank_world()
    if dist is None:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)
        return
    box = [ctx.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctx.comm_init(box[0], rank, world)
