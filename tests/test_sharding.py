"""N > 1 path on CPU: two gloo ranks shard an utterance list, process their
shards independently (here: the index plan stands in for the GPU work) and
reduce timing the way bench.py does."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libllsm2_amd.sharding import reduce_timing, shard_range, shard_strided, sweep_f0, utt_cost  # noqa: E402


def test_shard_range_partitions():
    for total in (0, 1, 7, 8, 1024, 8192, 1000):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                seen += list(shard_range(total, world, r))
            assert seen == list(range(total))
            sizes = [len(shard_range(total, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_shard_strided_partitions_and_balances_the_sweep():
    """the bench's partition: disjoint, complete, and -- on the F0-sorted sweep of BASELINE.json configs[2] -- equal
    modelled cost per rank to well under 5 % (contiguous blocks of the same list: 36 % over the mean at 8 ranks)"""
    for total in (0, 1, 7, 8, 1000, 8192):
        for world in (1, 2, 3, 4, 8):
            seen = sorted(u for r in range(world) for u in shard_strided(total, world, r))
            assert seen == list(range(total))
            sizes = [len(shard_strided(total, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    for world in (2, 4, 8):
        total = 1024 * world
        cost = [sum(utt_cost(sweep_f0(u, total)) for u in shard_strided(total, world, r)) for r in range(world)]
        assert max(cost) / min(cost) < 1.05, cost
        blocks = [sum(utt_cost(sweep_f0(u, total)) for u in shard_range(total, world, r)) for r in range(world)]
        assert max(blocks) / min(blocks) > 1.3                  # what the strided partition removes
    # the model follows the kernels: cost falls with F0, 80 Hz costs 1.5 - 2.5 x 400 Hz
    assert utt_cost(80.0) > utt_cost(120.0) > utt_cost(400.0) and 1.5 < utt_cost(80.0) / utt_cost(400.0) < 2.5


def test_sweep_endpoints():
    assert abs(sweep_f0(0, 8192) - 80.0) < 1e-9 and abs(sweep_f0(8191, 8192) - 400.0) < 1e-9
    assert all(sweep_f0(u + 1, 100) > sweep_f0(u, 100) for u in range(99))


def _worker(rank, world, port, total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = list(shard_range(total, world, rank))
    f0 = [sweep_f0(u, total) for u in mine]
    # each rank "processes" its shard: frames = 200 per utterance; fake per-rank time
    dt, frames = reduce_timing(0.5 + 0.25 * rank, 200 * len(mine))
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, f0))
    dist.barrier()
    if rank == 0:
        torch.save({"dt": dt, "frames": frames, "gathered": gathered}, out)
    dist.destroy_process_group()


def test_two_rank_gloo_sharding(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "res.pt")
    total = 37
    mp.spawn(_worker, args=(2, port, total, out), nprocs=2, join=True)
    res = torch.load(out, weights_only=False)
    assert res["frames"] == 200 * total
    assert abs(res["dt"] - 0.75) < 1e-12                      # MAX over ranks
    idx = res["gathered"][0][0] + res["gathered"][1][0]
    assert idx == list(range(total))                          # disjoint, complete, ordered
    f0 = res["gathered"][0][1] + res["gathered"][1][1]
    assert np.allclose(f0, [sweep_f0(u, total) for u in range(total)])


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
def test_bench_launcher_spawns_n_ranks(backend):
    """`python bench.py --gpus 2` with no torchrun environment must start 2 ranks itself
    (--gpus used to be parsed and ignored).  CPU-only plumbing check: with the default backend RCCL is
    tried and declined (no GPU here) and the group lands on gloo; LLSM_BENCH_BACKEND=gloo goes there directly."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["LLSM_BENCH_BACKEND"] = backend
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--utts", "128",
                          "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["frames"] == 2 * 128 * 200 and abs(res["max_dt"] - 0.02) < 1e-12
    assert res["rank_ms_per_step"] == [10.0, 20.0]               # every rank's own time, all-gathered (not only the MAX)
    assert res["my_utts_head"] == [0, 2, 4, 6]                   # strided partition: rank 0 owns the even utterances
    cs, cb = res["sweep_cost_strided"], res["sweep_cost_blocks"] # modelled cost of each rank's share of the sweep
    assert max(cs) / min(cs) < 1.05 < max(cb) / min(cb)
    pl = res["placement"]                                        # who ran where, as the bench line reports it
    assert pl["world_size"] == 2 and pl["backend"] == "gloo" and [r["rank"] for r in pl["ranks"]] == [0, 1]
    assert [r["device"] for r in pl["ranks"]] == [0, 1]          # LOCAL_RANK of each self-spawned rank


def test_bench_under_torchrun_falls_back_to_gloo_when_rccl_is_unavailable():
    """The driver's launch line (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`) on a box
    where RCCL cannot come up (here: no GPU): the group set-up of the bench (libllsm2_amd.sharding.init_timing_group)
    must land on gloo and finish.  Under torchrun the env:// rendezvous is the agent's store, so a fall-back that went
    back through env:// on another port waited forever (found on a 1-GPU box with two ranks sharing the device); it
    now builds the gloo group on a store of its own."""
    import json
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LLSM_BENCH_BACKEND")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--utts", "5", "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["frames"] == 2 * 5 * 200 and abs(res["max_dt"] - 0.02) < 1e-12
    assert "timing reductions over gloo" in out.stderr          # RCCL was tried first and declined


def test_bench_refuses_world_size_mismatch():
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)


def test_product_fanout_queue_covers_every_utterance_once():
    """The in-process fan-out of llsm_analyze_batch / llsm_synthesize_batch (csrc/capi.cpp): blocks of utterances
    pulled from one queue by several workers -- every utterance exactly once, blocks contiguous, several workers
    actually used.  Device-less workers (llsm_fanout_selftest): runs here."""
    import ctypes as C
    import libllsm2_amd as llsm
    L = llsm.load()
    L.llsm_fanout_plan.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.llsm_fanout_selftest.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int)]
    starts = (C.c_int * 64)()
    assert L.llsm_fanout_plan(1000, 256, starts, 64) == 4 and list(starts[:4]) == [0, 256, 512, 768]
    assert L.llsm_fanout_plan(0, 256, starts, 64) == 0 and L.llsm_fanout_plan(5, 0, starts, 64) == 5
    L.llsm_gpu_set_fanout(0, 0, 7)
    try:
        for n, workers in ((100, 4), (7, 3), (1, 2), (64, 8)):
            owner = (C.c_int * n)()
            assert L.llsm_fanout_selftest(n, workers, owner) == 0
            o = np.array(owner[:])
            assert np.all(o >= 0) and np.all(o < workers)
            for b0 in range(0, n, 7):                              # a block is handled by one worker
                assert len(set(o[b0:b0 + 7])) == 1
    finally:
        L.llsm_gpu_set_fanout(-1, -1, -1)


def _selftest(env_extra, gpus=2):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LLSM_BENCH_BACKEND",
                                                             "LLSM_BENCH_ASSUME_DEVICES")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--utts", "16",
                           "--launcher-selftest"], env=env, capture_output=True, text=True, timeout=240)


def test_bench_fails_when_rccl_is_down_on_a_node_that_has_the_devices():
    """on a node with a device for every rank, `bench.py --gpus N` must FAIL (non-zero exit) when RCCL
    does not come up -- not pass with placement.backend = "gloo" and a line on stderr.  Here (no GPU) RCCL always fails;
    LLSM_BENCH_ASSUME_DEVICES=2 makes the selftest apply the rule of a 2-device node."""
    out = _selftest({"LLSM_BENCH_ASSUME_DEVICES": "2"})
    assert out.returncode != 0, out.stdout[-500:]
    assert "RCCL did not come up" in out.stderr and "LLSM_BENCH_BACKEND=gloo" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]          # and no result line either


def test_bench_times_over_gloo_only_on_explicit_request():
    """... unless gloo is asked for by name: then the same node runs, and the top-level JSON says which backend timed it
    and how far apart the ranks were."""
    import json
    out = _selftest({"LLSM_BENCH_ASSUME_DEVICES": "2", "LLSM_BENCH_BACKEND": "gloo"})
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["backend"] == "gloo" and res["placement"]["backend"] == "gloo"
    assert abs(res["rank_ms_spread"] - 2.0) < 1e-12 and res["rank_ms_per_step"] == [10.0, 20.0]
