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


def init_timing_group(rank, world, device=None, backend=None, log=None):
    """Process group for the only collectives of the bench (a barrier and the MAX / SUM of two scalars).  RCCL
    (backend "nccl") first; if it cannot come up on this node -- or $LLSM_BENCH_BACKEND=gloo asks for it -- the same
    reductions run over gloo on host tensors.  The fallback does NOT go back through env://: under torchrun that
    rendezvous is the agent's store (a client connection to MASTER_PORT; another port has no server and every rank
    waits forever), so rank 0 opens a TCPStore of its own on MASTER_PORT + 1 and the group is built on it -- the same
    under torchrun and under a plain RANK / WORLD_SIZE launch.  Returns the backend in use."""
    import datetime
    import os
    import sys
    import torch
    import torch.distributed as dist
    backend = backend or os.environ.get("LLSM_BENCH_BACKEND", "nccl")
    log = log or (lambda m: print(m, file=sys.stderr, flush=True))
    if backend == "nccl":
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            probe = torch.zeros(1, device=device); dist.all_reduce(probe); torch.cuda.synchronize()
            return "nccl"
        except Exception as e:                            # noqa: BLE001
            log(f"rank {rank}: RCCL did not come up ({e!r}); timing reductions over gloo")
            if dist.is_initialized():
                dist.destroy_process_group()
    port = int(os.environ.get("MASTER_PORT", "29500")) + 1
    store = dist.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), port, world, is_master=(rank == 0),
                          timeout=datetime.timedelta(seconds=300))
    dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
    return "gloo"


def gather_rank_devices(rank, world, local):
    """[{rank, device, name, pci_bus_id, uuid}] of every rank (all_gather_object; identity without a group)."""
    import torch
    import torch.distributed as dist
    me = {"rank": rank, "device": local, "name": None, "pci_bus_id": None, "uuid": None}
    if torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(local)
        me["name"] = pr.name
        dom, bus, devid = (getattr(pr, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
        if bus is not None:
            me["pci_bus_id"] = f"{dom or 0:04x}:{bus:02x}:{devid or 0:02x}"
        u = getattr(pr, "uuid", None)
        me["uuid"] = str(u) if u is not None else None
    if not (world > 1 and dist.is_available() and dist.is_initialized()):
        return [me]
    out = [None] * world
    dist.all_gather_object(out, me)
    return out
