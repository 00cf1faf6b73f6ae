"""Utterance-level sharding across ranks (one process per GPU).

The layer-0 path has no cross-utterance state, so a job
of `total` utterances is block-partitioned over `world` ranks and every rank
processes its shard independently: no data-path collective.  torch.distributed
(RCCL on the GPU box, gloo in the CPU tests) is used only for the barrier and
for reducing the timed interval / frame counts."""


def shard_range(total, world, rank):
    """Contiguous block of utterance indices owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def sweep_f0(u, total, f_lo=80.0, f_hi=400.0):
    """BASELINE.json configs[2]: log sweep over the WHOLE job's utterance list."""
    return f_lo * (f_hi / f_lo) ** (u / max(total - 1, 1))


def reduce_timing(dt_seconds, frames, device=None):
    """(max over ranks of dt, sum over ranks of frames); identity without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dt_seconds, frames
    if dist.get_backend() == "gloo":
        device = None                                  # gloo reduces host tensors
    t = torch.tensor([dt_seconds], dtype=torch.float64, device=device)
    n = torch.tensor([frames], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())
