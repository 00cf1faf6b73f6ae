"""Utterance-level sharding across ranks (one process per GPU).

The layer-0 path has no cross-utterance state, so a job
of `total` utterances is partitioned over `world` ranks and every rank
processes its shard independently: no data-path collective.  torch.distributed
(RCCL on the GPU box, gloo in the CPU tests) is used only for the barrier and
for reducing the timed interval / frame counts.

Two partitions: contiguous blocks (`shard_range`: the in-process fan-out, where
neighbouring utterances share a download buffer) and the strided one the bench
uses (`shard_strided`): a job sorted by F0 -- the sweep of
BASELINE.json configs[2] -- costs ~40 % more per utterance at 80 Hz than at 400 Hz
(`utt_cost`), so contiguous blocks of it leave the step time to the rank that
drew the low end."""
import math


def shard_range(total, world, rank):
    """Contiguous block of utterance indices owned by `rank` (sizes differ by <= 1)."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def shard_strided(total, world, rank):
    """Utterances rank, rank + world, ... (round-robin): every rank sees the whole range of a sorted job."""
    return range(rank, total, world)


def utt_cost(f0, fs=44100.0, thop=0.005, maxnhar=100, nch=4, nhe=4, nfrm=200):
    """Modelled device cost of one analysed + resynthesised utterance of constant F0, in flops of the direct
    formulation (bench.py frame_alg): the harmonic analysis and the envelope analysis scale with
    window x harmonics, the resynthesis with the harmonic count, the transforms / filters / smoother not at all."""
    hw = int(round(fs / f0 * 4.0 / 2.0)) * 2
    nh = min(int(math.floor(fs / f0 / 2.0)), maxnhar)
    nwin = 2 * int(round(thop * fs))
    fft = lambda n: 5.0 * n * math.log2(n)
    ana = 4.0 * hw * nh + nch * 4.0 * hw * nhe + 2.0 * nh * nwin + 3 * fft(2048) + fft(1024) + 0.07e6
    syn = 2.0 * nh * nwin + 0.03e6 + 2 * fft(1024) + 0.02e6
    return nfrm * (ana + syn)


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


def gather_rank_times(ms, rank, world):
    """[ms of rank 0, ms of rank 1, ...] on every rank (all_gather_object; [ms] without a group): the bench prints
    it beside the MAX so that an imbalance between ranks is visible instead of hidden in the maximum."""
    import torch.distributed as dist
    if not (world > 1 and dist.is_available() and dist.is_initialized()):
        return [float(ms)]
    out = [None] * world
    dist.all_gather_object(out, float(ms))
    return out


class RcclUnavailable(RuntimeError):
    """RCCL did not come up on a node that has a device for every rank, and nobody asked for gloo."""


def init_timing_group(rank, world, device=None, backend=None, log=None, strict=False):
    """Process group for the only collectives of the bench (a barrier and the MAX / SUM of two scalars).  RCCL
    (backend "nccl") first; if it cannot come up on this node -- or $LLSM_BENCH_BACKEND=gloo asks for it -- the same
    reductions run over gloo on host tensors.  strict (the bench passes: the node has at least `world` devices): a
    failing RCCL raises RcclUnavailable instead of falling back -- the first real N-GPU run must not pass quietly on
    gloo; only an EXPLICIT LLSM_BENCH_BACKEND=gloo (or backend="gloo") selects it there.  The fallback does NOT go back through env://: under torchrun that
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
            if dist.is_initialized():
                dist.destroy_process_group()
            if strict:
                raise RcclUnavailable(f"rank {rank}: RCCL did not come up on a node with a device for each of the {world} ranks "
                                      f"({e!r}); set LLSM_BENCH_BACKEND=gloo to time over gloo on purpose") from e
            log(f"rank {rank}: RCCL did not come up ({e!r}); timing reductions over gloo")
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
