#!/usr/bin/env python
"""bench.py -- layer-0 analyse+resynthesise throughput on MI355X.

Metric (BASELINE.json): frames/sec (layer-0 analyze+synth, 44.1 kHz, 5 ms hop).
One "step" = one pass of the hot path (llsm_gpu_batch_analyze followed by
llsm_gpu_batch_synthesize) over one batch that is already resident in HBM.
Workload at N=1 = BASELINE.json configs[1]: 1024 synthetic 1 s utterances,
fixed F0 = 120 Hz, 44.1 kHz, 5 ms hop, default options, f0_refine = 0.
For N > 1 every rank owns its own batch of the same shape (utterances are
independent units: no data-path collective; "scaling": "weak").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--utts U]
                    [--workload fixed120|sweep|rt64|rt64pbp|l1]

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this script its own
launcher: it starts N copies of itself (one process per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set) and relays rank 0's JSON line.
Under `python -m torch.distributed.run --nproc-per-node N` the environment is already
there and the script is one rank.  Every rank asserts world size == --gpus.

Prints ONE JSON line on rank 0 (contract in the task statement) with the extra objects
  roofline      SURVEY 8(d): whole-path achieved_fp32 / achieved_hbm from F_alg / B_alg per frame,
                plus the dominant kernel (largest share of GPU time, HIP-event timed on the
                context's stream inside the timed region) priced on ALGORITHMIC work: unique
                bytes in + out for the streaming kernels, direct-formulation flops otherwise;
                `traffic` = HBM bytes per launch from the committed rocprofv3 PMC passes
  value_e2e     the same step with pinned H2D of the inputs and pinned D2H of every parameter
                row and the three waveforms inside the timed region (PCIe-inclusive; never `value`)
  cpu_baseline  the CPU oracle (a from-scratch restatement of the reference, which cannot be
                built here: ciglet absent) on a bounded sample: single-core and all-core legs,
                threads are OpenMP threads inside the C oracle
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FS = 44100.0
THOP = 0.005
NX = 44100
NFRM = 200
PEAK_FP32_TFLOPS = 157.3      # MI355X_MICROARCH.md: FP32 vector == FP32 matrix peak
PEAK_HBM_GBS = 8000.0
PEAK_PCIE_GBS = 63.0          # MI355X_MICROARCH.md: PCIe Gen5 x16 host link (spec)
NTEMPLATE_EXT = 20000 + 128


# ------------------------------------------------------------------ workload
def synth_phases_noise(u, F0):
    rng = np.random.default_rng(20260927 + u)
    K = min(int(FS / 2 / F0), 100)
    phi = rng.uniform(-np.pi, np.pi, K)
    noise = rng.standard_normal(NX).astype(np.float32)
    return K, phi, noise


def make_batch_inputs(utts, f0_of, device):
    """Synthetic utterances of BASELINE.md section 3 (utts: GLOBAL utterance indices, which
    seed the generator); harmonic sums on the GPU via torch."""
    import torch
    n_utt = len(utts)
    x = np.empty((n_utt, NX), np.float32)
    n = torch.arange(NX, device=device, dtype=torch.float64)
    for u0 in range(0, n_utt, 32):
        acc = []
        for u in utts[u0:u0 + 32]:
            F0 = f0_of(u)
            K, phi, noise = synth_phases_noise(u, F0)
            k = torch.arange(1, K + 1, device=device, dtype=torch.float64)[:, None]
            ph = torch.as_tensor(phi, device=device)[:, None]
            s = ((0.3 / k) * torch.cos(2 * np.pi * k * F0 * n[None, :] / FS + ph)).sum(0)
            acc.append((s + 0.01 * torch.as_tensor(noise, device=device, dtype=torch.float64)).float())
        x[u0:u0 + len(acc)] = torch.stack(acc).cpu().numpy()
    return x


# ------------------------------------------------------- algorithmic work
MFMA_KERNELS = ("k_harm_speech_tile", "k_synth_ola", "k_synth_ola4", "k_synth_frames", "k_l1_frame", "k_rt_hop2")
FFT_KERNELS = ("k_spgm_env_wf", "k_spgm_env", "k_psd_frames_wf", "k_psd_frames", "k_noise_filter_ola", "k_noise_filter_wf", "k_noise_filter")
def plan(f0):
    """(harmonic window, nhar) of one F0 from the product's own index plan."""
    import libllsm2_amd as llsm
    L = llsm.load()
    return (L.llsm_gpu_plan_index(6, 0, 0, f0, THOP, FS, 4.0),
            L.llsm_gpu_plan_index(7, 100, 0, f0, THOP, FS, 4.0))


def frame_alg(f0, npsd=256, nch=4, nhe=4, literal=False):
    """(F_alg flops, B_alg bytes) of ONE analysed + resynthesised frame, direct formulation.
    literal=False (the accounting every `frac` of this file uses, the same as kernel_alg): a DFT bin of a REAL input
    costs 4 flops per sample (real x complex multiply-add) and a resynthesised sample 2 flops per harmonic
    (Re(A_k z_k)).  literal=True: SURVEY 8(d)'s printed figures, which price the same sums as complex x complex
    (8 / 4 flops) -- on that count the MFMA harmonic analysis alone would sit above the machine peak, so it is
    reported only for reference."""
    hw, nh = plan(f0)
    nwin = 442
    # a transform of a REAL frame: 2.5 N log2 N (half a complex transform -- the kernels pack a frame pair per complex
    # transform); literal: SURVEY 8(d)'s 5 N log2 N.  The same real-input convention as the DFT / resynthesis counts.
    fft = lambda n: (5.0 if literal else 2.5) * n * math.log2(n)
    dft, syn1 = (8.0, 4.0) if literal else (4.0, 2.0)
    ana = dft * hw * nh + nch * dft * hw * nhe + syn1 * nh * nwin + 3 * fft(2048) + fft(1024) + 0.02e6 + 0.05e6
    syn = syn1 * nh * nwin + 0.03e6 + 2 * fft(1024) + 0.02e6
    P = 4 + 4 + 8 * nh + 4 * npsd * 2 + 4 * nch + nch * (4 + 8 * nhe)
    hop_bytes = NX / NFRM * 4.0
    return ana + syn, hop_bytes + 2 * P + 3 * hop_bytes


def kernel_alg(kernel, n_utt, f0s):
    """ALGORITHMIC work of ONE (average) launch of `kernel` over the batch: ('flop'|'byte', amount).
    Streaming kernels are priced on UNIQUE bytes in + out (not on the passes the implementation makes)."""
    F = n_utt * NFRM
    X, Y = n_utt * NX, n_utt * 44321
    nspec, npsd, nch, nhe = 513, 256, 4, 4
    if kernel in ("k_harm_speech", "k_harm_speech_tile", "k_harm_env", "k_synth_ola", "k_synth_ola4", "k_synth_frames"):
        flops = 0.0
        for f0 in f0s:
            hw, nh = plan(f0)
            if kernel in ("k_harm_speech", "k_harm_speech_tile"):
                flops += NFRM * 4.0 * hw * nh                 # real-input DFT at nhar bins
            elif kernel == "k_harm_env":
                flops += NFRM * 4.0 * hw * nch * nhe
            else:
                flops += NFRM * 2.0 * nh * 442                # Re(A_k z_k) per (sample, harmonic)
        return "flop", flops
    ffts = {"k_spgm_env_wf": [2048] * 3, "k_spgm_env": [2048] * 3, "k_psd_frames_wf": [1024], "k_psd_frames": [1024],
            "k_noise_filter_ola": [1024] * 2, "k_noise_filter_wf": [1024] * 2, "k_noise_filter": [1024] * 2}.get(kernel)
    if ffts:
        # real-input transforms: 2.5 N log2 N each (one accounting -- the DFT and resynthesis counts
        # above already price real input); `literal_factor` of the roofline object carries SURVEY 8(d)'s 5 N log2 N
        return "flop", F * sum(2.5 * m * math.log2(m) for m in ffts)
    if kernel == "k_filtfilt":
        # two launches per step.  analysis: reads x and x_res (2 planes), writes the 4 squared sub-band
        # planes; synthesis: reads 4 white templates, writes 4 band-limited ones.  Averaged per launch.
        ana = (2 + nch) * X * 4.0
        syn = 2 * nch * n_utt * NTEMPLATE_EXT * 4.0
        return "byte", (ana + syn) / 2.0
    if kernel == "k_kalman":
        return "byte", 2.0 * F * nspec * 4 + 2.0 * F * npsd * 4      # two 513-bin planes in, psd + psdres rows out
    if kernel in ("k_excite_env", "k_excite_env4"):
        return "byte", nch * n_utt * NTEMPLATE_EXT * 4.0 + F * (nch * 4 + nch * nhe * 8) + Y * 4.0
    if kernel == "k_white":
        return "byte", nch * n_utt * NTEMPLATE_EXT * 4.0
    if kernel == "k_harm_speech_rest":
        return None, None                            # the frames outside shared-F0 tiles: none in the bench workloads
    if kernel == "k_env_params":
        return "byte", F * (nch * nhe * 8) * 2.0
    return None, None


def pmc_traffic():
    """{kernel: HBM bytes per launch} from the newest committed rocprofv3 PMC passes of this command
    (profiles/*traffic.json, written by tools/gpu_prof.sh + tools/rocpd_traffic.py: separate FETCH_SIZE /
    WRITE_SIZE passes, gfx950 correction bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per the guide)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic.json")))
    if not files:
        return {}, None
    try:
        t = json.load(open(files[-1]))
        return {k: v["hbm_bytes_per_launch"] for k, v in t.items()}, os.path.basename(files[-1])
    except Exception:
        return {}, None


# ------------------------------------------------------------------ CPU baseline
def host_cpus():
    """(usable CPUs, how that was found): the smaller of the affinity mask and the cgroup CPU quota -- a container
    whose mask shows every host thread may still be throttled to a few CPUs' worth of time."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:                                                            # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:                                                        # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None and quota < aff:
        return max(1, int(math.ceil(quota))), {"affinity_mask": aff, "cgroup_cpu_quota": quota}
    return aff, {"affinity_mask": aff, "cgroup_cpu_quota": quota}


def cpu_baseline(budget_s=12.0):
    """CPU oracle (float32 build, FFT-based CZT like the reference's ciglet) on the host cores of
    this box: (i) single core, (ii) a thread-scaling sweep 1, 2, 4, ... up to the usable CPUs (one utterance per
    OpenMP thread inside the C oracle), (iii) the leg at the thread count where the sweep peaks.  `cores` is the
    thread count of the reported value; `usable_cpus` says where the ceiling came from (affinity mask vs cgroup quota)."""
    import ctypes as C
    from oracle.oracle import Oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_utterance
    o = Oracle(np.float32)
    cores, how = host_cpus()
    nd = 4
    xs = np.concatenate([make_utterance(u, 120.0) for u in range(nd)]).astype(np.float32)
    f0 = np.full(NFRM, 120.0, np.float32)
    ao = o.aoptions(f0_refine=0)
    so = o.soptions(FS)
    fn = o.lib.o_bench_anasynth
    fn.restype = C.c_double
    o.lib.o_set_czt_mode(C.c_int(1))

    def run(n_utt, threads):
        frames = C.c_longlong(0)
        dt = fn(C.byref(ao), C.byref(so), o.p(xs), C.c_int(NX), C.c_int(nd), o.f(FS), o.p(f0), C.c_int(NFRM),
                C.c_int(n_utt), C.c_int(threads), C.byref(frames))
        return frames.value / dt, dt

    r1, dt1 = run(1, 1)                                              # warm-up + rate estimate
    n1 = max(2, int(budget_s * 0.3 * r1 / NFRM))
    single, dts = run(n1, 1)
    # thread-scaling sweep: two utterances per thread at 1, 2, 4, ... threads (about 2 x 0.15 s of work per thread)
    sweep, t = [], 1
    mask = how["affinity_mask"]
    falling = 0
    while True:
        r, _ = run(2 * t, t)
        sweep.append({"threads": t, "value": r, "per_thread": r / t})
        falling = falling + 1 if r < 0.95 * max(e["value"] for e in sweep) else 0
        if t >= mask or falling >= 2:                 # two doublings past the knee: the curve is known
            break
        t = min(2 * t, mask)
    best = max(sweep, key=lambda e: e["value"])
    nall = max(best["threads"], int(budget_s * 0.5 * best["value"] / NFRM) // best["threads"] * best["threads"])
    allc, dta = run(nall, best["threads"])
    o.lib.o_set_czt_mode(C.c_int(0))
    return {"value": allc, "unit": "frames/s", "cores": best["threads"], "kind": "port",
            "usable_cpus": dict(how, used=cores),
            "thread_scaling": sweep,
            "single_core": {"value": single, "cores": 1, "sample": f"{n1} utterances x {NFRM} frames, {dts:.1f} s"},
            "sample": f"{nall} utterances x {NFRM} frames (synthetic config-2 utterances), float32 CPU restatement of "
                      f"libllsm2 layer-0 with FFT-based CZT (reference not buildable: ciglet unavailable), one utterance "
                      f"per OpenMP thread, {best['threads']} threads (the peak of the thread-scaling sweep), {dta:.1f} s"}


# ------------------------------------------------------------------ launcher
def self_launch(args, argv):
    """`--gpus N` without a torchrun environment: start N ranks of this script, relay rank 0."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    for p in procs[1:]:
        rc = p.wait() or rc
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def launcher_selftest(args, world, rank):
    """CPU-only check of the N-rank plumbing (tests/test_sharding.py): gloo group, shard plan, barrier,
    MAX / SUM reduction -- no llsm compute, no throughput claim."""
    import torch.distributed as dist
    from libllsm2_amd.sharding import reduce_timing, shard_range, sweep_f0
    if world > 1:
        # the bench's own group set-up: RCCL first (on a CPU box it fails at once), then the gloo fall-back on a store
        # of its own -- the path a node without a working RCCL takes, under torchrun and under the self-spawned ranks
        # ($LLSM_BENCH_ASSUME_DEVICES=n: pretend the node has n devices -- with n >= world a failing RCCL is fatal, as
        # in the real bench below, unless LLSM_BENCH_BACKEND=gloo asks for gloo explicitly)
        from libllsm2_amd.sharding import RcclUnavailable, init_timing_group
        try:
            used = init_timing_group(rank, world, None, log=lambda m: print("bench.py: " + m, file=sys.stderr, flush=True),
                                     strict=int(os.environ.get("LLSM_BENCH_ASSUME_DEVICES", "0")) >= world)
        except RcclUnavailable as e:
            raise SystemExit(f"bench.py: {e}")
        assert dist.get_world_size() == args.gpus and used in ("nccl", "gloo"), (dist.get_world_size(), args.gpus)
    from libllsm2_amd.sharding import gather_rank_devices, gather_rank_times, shard_strided, utt_cost
    total = args.utts * world
    mine = shard_strided(total, world, rank)            # the partition bench_layer0 uses
    dt, frames = reduce_timing(0.01 * (rank + 1), len(mine) * NFRM)
    # modelled cost of this rank's share of the 80 -> 400 Hz sweep, under the strided partition and under contiguous blocks
    cost = gather_rank_times(sum(utt_cost(sweep_f0(u, total)) for u in mine), rank, world)
    cost_block = gather_rank_times(sum(utt_cost(sweep_f0(u, total)) for u in shard_range(total, world, rank)), rank, world)
    rank_ms = gather_rank_times(10.0 * (rank + 1), rank, world)
    ranks = gather_rank_devices(rank, world, int(os.environ.get("LOCAL_RANK", "0")))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "n_gpus": world, "frames": frames, "max_dt": dt,
                          "placement": {"world_size": world, "backend": used if world > 1 else None, "ranks": ranks},
                          "backend": used if world > 1 else None,
                          "rank_ms_per_step": rank_ms, "rank_ms_spread": max(rank_ms) / min(rank_ms), "sweep_cost_strided": cost, "sweep_cost_blocks": cost_block,
                          "my_utts_head": list(mine)[:4],
                          "f0_first_last": [sweep_f0(0, args.utts * world), sweep_f0(args.utts * world - 1, args.utts * world)]}))
    return 0


# ------------------------------------------------------------------ llsmrt workload (config 4 shape)
def bench_rt(args, llsm, world, rank, local, dev, dist, placement=None, workload=None, steps=None, warmup=None, pipeline=None,
             streams=None, hops_per_feed=None):
    """BASELINE.json configs[3]: 64 lock-stepped llsmrt streams per GPU fed from analysed config-2 chunks, the
    consumer pulls 256 samples per stream per iteration.  rt64: harmonic-model path; rt64pbp: the chunk is taken to
    layer 1 (llsm_chunk_tolayer1), its harmonic models dropped and every frame marked PBPSYN, options.use_l1 = 1:
    the pulse-by-pulse path of llsmrt.c:295-420 (pulse scheduling on the host, pulses on the device)."""
    import ctypes as C
    from libllsm2_amd.sharding import reduce_timing
    L = llsm.load()
    S = args.streams if streams is None else streams
    ao = llsm.make_aoptions(f0_refine=0)
    x = make_batch_inputs([0], lambda u: 120.0, dev)[0]
    f0 = np.full(NFRM, 120.0, np.float32)
    L.llsm_analyze.restype = C.POINTER(llsm.Chunk)
    ch = L.llsm_analyze(C.byref(ao), x.ctypes.data_as(llsm.P_fp), NX, FS, f0.ctypes.data_as(llsm.P_fp), NFRM, None)
    if not ch:
        raise SystemExit("llsm_analyze failed: " + L.llsm_gpu_last_error().decode())
    workload = workload or args.workload
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    pbp = workload == "rt64pbp"
    if pbp:
        L.llsm_chunk_tolayer1(ch, 2048)
        for i in range(NFRM):
            fr = ch.contents.frames[i]
            L.llsm_container_attach_(fr, llsm.FRAME_HM, None, None, None)
            L.llsm_container_attach_(fr, llsm.FRAME_PBPSYN, C.cast(L.llsm_create_int(1), C.c_void_p),
                                     C.cast(L.llsm_delete_int, C.c_void_p), C.cast(L.llsm_copy_int, C.c_void_p))
        L.llsm_chunk_phasepropagate(ch, 1)
    so = llsm.make_soptions(FS, use_l1=1 if pbp else 0)
    if args.rt_graph >= 0:
        L.llsm_gpu_rt_graph(args.rt_graph)
    pipeline = args.rt_pipeline if pipeline is None else pipeline
    prev_pipeline = L.llsm_gpu_rt_pipeline(-1)
    if pipeline >= 0:
        L.llsm_gpu_rt_pipeline(pipeline)
    g = L.llsm_create_rtsynth_group(C.byref(so), ch.contents.conf, 8192, S)
    if not g:
        raise SystemExit("llsm_create_rtsynth_group failed: " + L.llsm_gpu_last_error().decode())
    g = C.c_void_p(g)
    # the host side of a hop is two library calls (feed every stream, pull every stream): the frame-pointer arrays are
    # built once and the pull goes through llsm_rtsynth_group_fetch_all, so that the loop measures the library and not
    # 128 ctypes calls per hop (the per-stream form of this loop ran at 0.36 ms per hop, 0.05 ms of it on the device)
    Frames = C.POINTER(llsm.Container) * S
    fr_all = ch.contents.frames
    frames_of = []
    for i in range(NFRM):
        a = Frames()
        for s in range(S):
            a[s] = fr_all[i]
        frames_of.append(a)
    bufp = np.zeros((S, 256), np.float32); bufa = np.zeros((S, 256), np.float32)
    pp, pa = bufp.ctypes.data_as(llsm.P_fp), bufa.ctypes.data_as(llsm.P_fp)
    pull_lat = []

    K = max(1, args.rt_hops if hops_per_feed is None else hops_per_feed)
    if K > 1:                                            # K hops per library call (llsm_rtsynth_group_feed_many): frames[k * S + s]
        L.llsm_rtsynth_group_feed_many.argtypes = [C.c_void_p, C.POINTER(C.POINTER(llsm.Container)), C.c_int]
        many_of = []
        for i in range(0, NFRM, K):
            a = (C.POINTER(llsm.Container) * (S * K))()
            for k in range(K):
                for s in range(S):
                    a[k * S + s] = fr_all[(i + k) % NFRM]
            many_of.append(a)

    def hop(i):
        if K > 1:
            if i % K:
                return
            L.llsm_rtsynth_group_feed_many(g, many_of[(i // K) % len(many_of)], K)
        else:
            L.llsm_rtsynth_group_feed(g, frames_of[i % NFRM])
        while L.llsm_rtsynth_group_numoutput(g, 0) >= 256:
            t = time.perf_counter()
            L.llsm_rtsynth_group_fetch_all(g, pp, pa, 256, None)
            pull_lat.append(time.perf_counter() - t)

    for i in range(warmup * 20):
        hop(i)
    if world > 1:
        dist.barrier()
    pull_lat.clear()
    t0 = time.perf_counter()
    nh = steps * 200
    for i in range(nh):
        hop(i)
    dt = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    dt, frames_all = reduce_timing(dt, nh * S, dev)
    pipelined = bool(L.llsm_gpu_rt_pipeline(-1))
    L.llsm_delete_rtsynth_group(g)
    L.llsm_delete_chunk(ch)
    L.llsm_gpu_rt_pipeline(prev_pipeline)
    if rank == 0:
        return ({
            "metric": "frames/sec (llsmrt pull loop, 44.1 kHz, 5 ms hop)", "value": frames_all / dt, "unit": "frames/s",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"llsmrt: {S} concurrent streams per GPU fed from analysed config-2 frames ("
                                   + ("layer-1 frames, pulse-by-pulse path, use_l1 = 1" if pbp else "harmonic-model path")
                                   + "), 256-sample pulls per stream, one step = 200 hops of every stream",
                       "streams_per_gpu": S, "parallelism": f"dp{world}"},
            "hop_as_graph": bool(L.llsm_gpu_rt_graph(-1)), "launches_per_hop_mode": int(L.llsm_gpu_rt_fused(-1)),
            "pinned_blocks_direct": bool(L.llsm_gpu_rt_direct(-1)),
            "hops_per_feed": K,                # > 1: llsm_rtsynth_group_feed_many, hop k + 1 packed while hop k is on the device
            "pipelined_feeds": pipelined,      # True: a feed returns once its hop is enqueued, its samples are visible one feed later
            "ms_per_hop": dt / nh * 1e3, "realtime_factor_per_stream": nh * THOP / dt,
            "max_pull_ms": max(pull_lat) * 1e3 if pull_lat else None, "placement": placement})
    return None


def bench_l1(args, llsm, world, rank, local, dev, dist, placement=None, steps=None, warmup=None, x=None):
    """SURVEY 8(f) rank 1 / BASELINE.json configs[4] shape on the batch API: the analysed config-2 batch is taken to
    layer 1 (Rd fit, vocal-tract envelope, phase residual), its harmonic models are dropped and the frames
    i % 100 > 50 marked PBPSYN (the pattern of test-layer1-anasynth.c:34-39); one step = llsm_gpu_batch_tolayer1 +
    llsm_gpu_batch_synthesize(use_l1 = 1): layer 1 -> layer 0 for the harmonic-model frames, pulse scheduling on the
    host, every pulse group on the device, cross-fade, noise path.  No effect callback (a Python callback per pulse
    would time ctypes)."""
    import torch
    from libllsm2_amd.sharding import reduce_timing
    U = args.utts
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    if x is None:                                      # (the default run hands over the headline's fixed-F0 batch)
        x = make_batch_inputs(list(range(rank * U, (rank + 1) * U)), lambda u: 120.0, dev)
    f0 = np.full(U * NFRM, 120.0, np.float32)
    ctx = llsm.Context(local)
    ao = llsm.make_aoptions(f0_refine=0)
    so = llsm.make_soptions(FS, use_l1=1)
    b = llsm.Batch(ctx, ao, FS, [NX] * U, [NFRM] * U)
    b.upload(llsm.A_X, x.reshape(-1)); b.upload(llsm.A_F0, f0)
    b.analyze()
    b.enable_layer1(2048)
    pbp = (np.arange(U * NFRM) % NFRM % 100 > 50).astype(np.int32)
    zeros = np.zeros(U * NFRM, np.int32)
    b.upload(llsm.A_PBPSYN, pbp)
    t_host = {"tolayer1": 0.0, "synthesize": 0.0}
    t_steps = []                                      # host wall time of every timed step's two calls: an outlier shows here

    def step(i):
        t0 = time.perf_counter()
        b.tolayer1(2048)
        b.upload(llsm.A_HAS_HM, zeros)                # every step rebuilds the harmonic models it needs
        t1 = time.perf_counter()
        b.synthesize(so, seed=1000 + i)
        t2 = time.perf_counter()
        t_host["tolayer1"] += t1 - t0; t_host["synthesize"] += t2 - t1
        t_steps.append((t2 - t0) * 1e3)

    def fence():
        ctx.sync(); torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    for i in range(warmup):
        step(i)
    fence()
    t_host = {k: 0.0 for k in t_host}
    del t_steps[:]
    ctx.set_profiling(True); ctx.reset_profile()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    fence()
    dt = time.perf_counter() - t0
    prof = ctx.profile(); ctx.set_profiling(False)
    dt, frames_all = reduce_timing(dt, U * NFRM * steps, dev)
    y = b.download(llsm.A_Y)[: b.y_off[1]]
    ok = bool(np.all(np.isfinite(y)) and 0.5 < np.sqrt(np.mean(y[4000:40000] ** 2)) / np.sqrt(np.mean(x[0, 4000:40000] ** 2)) < 1.5)
    if rank == 0:
        F = U * NFRM
        fft = lambda m: 2.5 * m * math.log2(m)      # transforms of REAL sequences (log spectra, cepstra, frames): the accounting of bench_layer0
        # algorithmic FLOPs per launch (direct count of the transforms + the float64 LF spectrum at ~150 flop / point)
        npulse = 0.3 * F    # 49 % of the frames are PBPSYN, 0.6 glottal pulses per 5 ms hop at 120 Hz (the scheduler reports 63 k)
        alg = {"k_l1_frame": F * (2 * fft(512) + 100 * 150.0),                    # LF removal + minimum phase (512 points)
               "k_l1_env_wf": F * (2 * fft(2048) + 1025 * 3 * 30.0),              # lobes on 1025 bins + cepstral smoothing
               "k_l1_to_l0": F * (2 * fft(512) + 100 * 150.0),
               "k_l1_rd_fit": F * 64 * 80 * 6.0,
               "k_pbp_pulse": npulse * (fft(2048) + 2 * fft(512) + 1024 * 60.0),  # minimum phase, LF spectrum (f64), inverse FFT
               "k_noise_filter_ola": F * 2 * fft(1024)}
        tot = sum(v[0] for v in prof.values())
        dom = max(prof.items(), key=lambda kv: kv[1][0])[0]

        def roof_of(name):
            ms, launches = prof[name]
            r = {"kernel": name, "avg_launch_ms": ms / launches, "launches_per_step": launches / steps,
                 "share_of_gpu_time": ms / tot, "traffic": None}
            w = alg.get(name)
            if w:
                ach = w / (ms / launches * 1e-3) / 1e12
                r.update({"bound": "mfma" if name in MFMA_KERNELS else "fp32-vector", "achieved": ach, "peak": PEAK_FP32_TFLOPS,
                          "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS, "algorithmic_gflop_per_launch": w / 1e9})
            else:
                r.update({"bound": "mfma" if name in MFMA_KERNELS else "fp32-vector", "achieved": None, "peak": PEAK_FP32_TFLOPS,
                          "unit": "TFLOP/s", "frac": None})
            return r
        res = ({
            "metric": "frames/sec (layer-1 conversion + use_l1 synthesis, 44.1 kHz, 5 ms hop)", "value": frames_all / dt,
            "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (LF model f64)", "data": "synthetic",
            "config": {"workload": f"{U} analysed synthetic 1 s utterances per GPU (F0 120 Hz): llsm_gpu_batch_tolayer1(2048) + "
                                   "use_l1 synthesis, harmonic models dropped, PBPSYN on frames i % 100 > 50, no effect callback",
                       "utterances_per_gpu": U, "frames_per_utterance": NFRM, "parallelism": f"dp{world}"},
            "roofline": roof_of(dom),
            "roofline_other_kernels": [roof_of(k) for k, _ in sorted(prof.items(), key=lambda kv: -kv[1][0]) if k != dom][:6],
            "kernels_ms_per_step": {k: v[0] / steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
            "gpu_ms_per_step": tot / steps,
            "host_ms_per_step": {k: v / steps * 1e3 for k, v in t_host.items()},
            "host_ms_of_each_step": [round(t, 3) for t in t_steps],
            "note": "host_ms_per_step = wall time of the two calls (the pulse scheduler of layer0.c:148-287 runs on the host in "
                    "float64, in the reference's order, before the pulse launch); gpu_ms_per_step = sum of kernel times",
            "sanity_ok": ok, "placement": placement})
    else:
        res = None
    b.close(); ctx.close()
    return res


# ------------------------------------------------------------------ layer-0 workloads
def measure_e2e(args, llsm, local, ao, so, x, f0, U, fence, reduce):
    """SURVEY 8(d)'s wall-clock metric: frames/s INCLUDING the page-locked upload of x / f0 and the page-locked download
    of every parameter row and the three waveforms.  The batch goes as `parts` sub-batches on their own contexts
    (streams), one host thread each -- the arrangement of the library's own fan-out (capi.cpp) --, so the transfers of
    one overlap the kernels of the others.  Reported beside `value`, never as it.

    What the figure stands on: per `parts` setting 2 warm-up steps, then `e2e_reps` repetitions of
    `e2e_steps` timed steps (min / median / max); `parts` swept over 2 / 4 / 8 once and the best median kept; the two
    directions timed ALONE over the same buffers and threads (H2D GB/s, D2H GB/s: what the link and the host give on
    this box without any kernel in the way) and the kernels alone -- so a low pcie_frac can be put on the link, on the
    overlap, or on the device; the host threads and the page-locked blocks bound to the device's NUMA node."""
    import threading
    L = llsm.load()
    numa_node = L.llsm_gpu_device_numa_node(local)
    # y = y_sin + y_noise is one float addition per sample: only the two parts cross the link, the host forms the sum
    # (llsm_gpu_sum_outputs, inside the timed step; bit-identical to the device's row -- asserted once per run below)
    ids_w = [llsm.A_YSIN, llsm.A_YNOISE]
    nsteps, nreps = max(1, args.e2e_steps), max(1, args.e2e_reps)
    affinity0 = os.sched_getaffinity(0)                                   # restored below: the CPU baseline wants every core
    bound = {"main": L.llsm_gpu_bind_thread_to_device(local)}            # page-locked blocks are allocated from this thread

    def build(nparts):
        parts = []
        cuts = [U * k // nparts for k in range(nparts + 1)]
        for u0, u1 in zip(cuts[:-1], cuts[1:]):
            if u1 <= u0:
                continue
            c2 = llsm.Context(local)
            b2 = llsm.Batch(c2, ao, FS, [NX] * (u1 - u0), [NFRM] * (u1 - u0))
            pin_in = {llsm.A_X: b2.pinned_array(llsm.A_X), llsm.A_F0: b2.pinned_array(llsm.A_F0)}
            pin_in[llsm.A_X][:] = x[u0:u1].reshape(-1); pin_in[llsm.A_F0][:] = f0[u0 * NFRM:u1 * NFRM]
            raw, views = b2.pinned_params_block()                         # the eleven rows: one block, one copy
            pin_w = {a: b2.pinned_array(a) for a in ids_w}
            y_host = np.empty_like(pin_w[llsm.A_YSIN])                    # ordinary memory: the sum is the host's own
            parts.append(dict(ctx=c2, b=b2, pin_in=pin_in, raw=raw, views=views, pin_w=pin_w, y_host=y_host))
        return parts

    def destroy(parts):
        for p_ in parts:
            for buf in list(p_["pin_in"].values()) + list(p_["pin_w"].values()) + [p_["raw"]]:
                p_["b"].free_pinned(buf)
            p_["b"].close(); p_["ctx"].close()

    def worker(p_, n, mode, t_acc):
        if "bound" not in p_:
            p_["bound"] = L.llsm_gpu_bind_thread_to_device(local)
        b2 = p_["b"]
        for i in range(n):
            t0 = time.perf_counter()
            if mode in ("all", "h2d"):
                b2.transfer_many(p_["pin_in"], to_device=True)
            t1 = time.perf_counter()
            if mode in ("all", "compute"):
                b2.analyze(); b2.synthesize(so, seed=1000 + i)
                if mode == "compute":
                    p_["ctx"].sync()
            t2 = time.perf_counter()
            if mode in ("all", "d2h"):
                b2.transfer_params_block(p_["raw"], to_device=False)
                b2.transfer_many(p_["pin_w"], to_device=False)
                if mode == "all":
                    ys_, yn_ = p_["pin_w"][llsm.A_YSIN], p_["pin_w"][llsm.A_YNOISE]
                    L.llsm_gpu_sum_outputs(p_["y_host"].ctypes.data, ys_.ctypes.data, yn_.ctypes.data, ys_.size)
            t3 = time.perf_counter()
            t_acc[0] += t1 - t0; t_acc[1] += t2 - t1; t_acc[2] += t3 - t2

    def run(parts, n, mode="all"):
        accs = [[0.0, 0.0, 0.0] for _ in parts]
        th = [threading.Thread(target=worker, args=(p_, n, mode, a_)) for p_, a_ in zip(parts, accs)]
        fence()
        t1 = time.perf_counter()
        [t.start() for t in th]; [t.join() for t in th]
        fence()
        return time.perf_counter() - t1, accs

    def nbytes_of(parts):
        up = sum(sum(v.nbytes for v in p_["pin_in"].values()) for p_ in parts)
        down = sum(p_["raw"].nbytes + sum(v.nbytes for v in p_["pin_w"].values()) for p_ in parts)
        return up, down

    sweep = {}
    best = None
    settings = [args.e2e_parts] if args.e2e_parts > 0 else [2, 4, 8]
    for nparts in settings:
        parts = build(nparts)
        up, down = nbytes_of(parts)
        run(parts, 2)                                                     # warm-up
        for p_ in parts[:1]:                                              # the host's sum IS the device's y row
            assert np.array_equal(p_["y_host"], p_["b"].download(llsm.A_Y)), "host-formed y differs from the device's"
        reps = []
        for _ in range(nreps):
            dte, accs = run(parts, nsteps)
            dte, frames_e = reduce(dte, nsteps)
            reps.append((dte / nsteps * 1e3, frames_e / dte, accs))
        ms = sorted(r[0] for r in reps)
        med = ms[len(ms) // 2]
        rec = {"parts": len(parts), "ms_per_step": {"min": ms[0], "median": med, "max": ms[-1]},
               "frames_per_s": {"min": min(r[1] for r in reps), "median": sorted(r[1] for r in reps)[len(reps) // 2],
                                "max": max(r[1] for r in reps)},
               "spread": (ms[-1] - ms[0]) / med}
        # the directions and the kernels alone, same buffers / threads / streams
        t_h2d, _ = run(parts, 4, "h2d")
        t_d2h, _ = run(parts, 4, "d2h")
        t_cmp, _ = run(parts, 4, "compute")
        rec["alone"] = {"h2d_gbs": up * 4 / t_h2d / 1e9, "d2h_gbs": down * 4 / t_d2h / 1e9,
                        "h2d_ms_per_step": t_h2d / 4 * 1e3, "d2h_ms_per_step": t_d2h / 4 * 1e3, "compute_ms_per_step": t_cmp / 4 * 1e3}
        accs = reps[len(reps) // 2][2]                                    # where a host thread's time went (one repetition)
        tot = max(sum(sum(a_) for a_ in accs), 1e-12)
        rec["thread_time_share"] = {"upload": sum(a_[0] for a_ in accs) / tot, "enqueue_kernels": sum(a_[1] for a_ in accs) / tot,
                                    "download_incl_wait_for_kernels": sum(a_[2] for a_ in accs) / tot}
        rec["threads_bound_cpus"] = [p_.get("bound", 0) for p_ in parts]
        sweep[str(nparts)] = rec
        if best is None or med < best[1]:
            best = (nparts, med, rec, up, down)
        destroy(parts)
    os.sched_setaffinity(0, affinity0)
    nparts, med, rec, up, down = best
    nbytes = up + down
    d2h_floor_ms = down / (PEAK_PCIE_GBS * 1e9) * 1e3
    return {"value": rec["frames_per_s"]["median"], "unit": "frames/s", "steps": nsteps, "repetitions": nreps, "warmup": 2,
            "ms_per_step": med, "ms_per_step_min_median_max": [rec["ms_per_step"]["min"], med, rec["ms_per_step"]["max"]],
            "spread": rec["spread"],
            "metric": "SURVEY 8(d) wall-clock metric: frames/s including H2D of the waveforms and D2H of every parameter row "
                      "and waveform (never `value`, which times HBM-resident inputs per the bench contract); median of the repetitions",
            "pcie_bytes_per_step": nbytes, "h2d_bytes_per_step": up, "d2h_bytes_per_step": down,
            "pcie_gbs": nbytes / (med * 1e-3) / 1e9, "pcie_peak_gbs": PEAK_PCIE_GBS,
            "pcie_frac": nbytes / (med * 1e-3) / 1e9 / PEAK_PCIE_GBS,
            "d2h_gbs_in_step": down / (med * 1e-3) / 1e9,
            "d2h_floor": {"ms_per_step": d2h_floor_ms, "frames_per_s": U * NFRM / (d2h_floor_ms * 1e-3),
                          "note": "the download alone at the link's 63 GB/s: no arrangement of this step can be faster",
                          "frac_of_floor": d2h_floor_ms / med},
            "directions_alone": rec["alone"], "thread_time_share": rec["thread_time_share"],
            "host_buffers": "page-locked (llsm_gpu_alloc_host); the eleven parameter rows as one block, one copy; y_sin and y_noise "
                            "downloaded, y formed on the host (llsm_gpu_sum_outputs, inside the timed step, asserted equal to the device's row)",
            "numa": {"device_node": numa_node, "main_thread_bound_cpus": bound["main"], "worker_threads_bound_cpus": rec["threads_bound_cpus"],
                     "note": "node from sysfs numa_node of the PCI device; 0 CPUs bound = node unknown / already inside it / no overlap with the cgroup's CPUs"},
            "parts": nparts, "parts_sweep": {k: {"ms_per_step_median": v["ms_per_step"]["median"], "spread": v["spread"],
                                                 "h2d_gbs_alone": v["alone"]["h2d_gbs"], "d2h_gbs_alone": v["alone"]["d2h_gbs"]}
                                             for k, v in sweep.items()},
            "note": "upload x + f0, analyse, synthesise, download every parameter row and y / y_sin / y_noise; the batch in "
                    "`parts` sub-batches, one context (stream) and host thread each, so that the PCIe link stays busy while the "
                    "others compute"}


def bench_layer0(args, llsm, world, rank, local, dev, dist, placement, workload, steps, warmup, full=True):
    """analyse + resynthesise a resident batch (BASELINE.json configs[1] `fixed120`, configs[2] `sweep`); returns
    (rank 0's result dict | None, this rank's input waveforms)."""
    from libllsm2_amd.sharding import gather_rank_times, reduce_timing, shard_strided, sweep_f0
    import torch
    total_u = args.utts * world                        # weak scaling: per-GPU work is fixed
    if workload == "fixed120":
        f0_of = lambda u: 120.0
    else:                                              # BASELINE.json configs[2]: log sweep 80 -> 400 Hz
        f0_of = lambda u: sweep_f0(u, total_u)
    # STRIDED partition: rank r owns utterances r, r + world, ... of the (F0-sorted) list, so
    # every rank sees the whole F0 range and the per-rank cost is equal to within one sweep step; a rank's own list
    # stays sorted, which is what the shared-F0 tiles and the XCD-local window reuse want.  (Contiguous blocks of the
    # sorted sweep gave rank 0 the 80-98 Hz utterances and rank 7 the 327-400 Hz ones: ~40 % more work on rank 0.)
    my_utts = shard_strided(total_u, world, rank)
    U = len(my_utts)
    f0s = [float(np.float32(f0_of(u))) for u in my_utts]
    x = make_batch_inputs(list(my_utts), f0_of, dev)
    f0 = np.repeat(np.asarray(f0s, np.float32), NFRM)

    ctx = llsm.Context(local)
    ao = llsm.make_aoptions(f0_refine=0)
    so = llsm.make_soptions(FS)
    b = llsm.Batch(ctx, ao, FS, [NX] * U, [NFRM] * U)
    b.upload(llsm.A_X, x.reshape(-1))
    b.upload(llsm.A_F0, f0)

    def step(i):
        b.analyze()
        b.synthesize(so, seed=1000 + i)

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for i in range(warmup):
        step(i)
    fence()
    # Survey pass (UNTIMED, after the warm-up): HIP events around every launch of a few steps -- which kernel is the dominant
    # one, and the per-kernel times of `roofline_other_kernels` / `kernels_ms_per_step`.
    # The survey runs every launch on ONE stream (llsm_gpu_analysis_overlap(0)): each kernel's time is its time alone.
    survey_steps = max(2, min(5, steps))
    llsm.load().llsm_gpu_analysis_overlap(0)
    ctx.set_profiling(True)
    ctx.reset_profile()
    for i in range(survey_steps):
        step(10000 + i)
    fence()
    prof = ctx.profile()
    llsm.load().llsm_gpu_analysis_overlap(1)
    dom = max(prof.items(), key=lambda kv: kv[1][0])[0]
    # Timed region: the product path as a caller runs it, with events around the launches of the DOMINANT kernel only
    # (two events per step: the roofline's `avg_launch_ms` is measured live here, on the stream that kernel is launched on).
    # Rounds 1 - 5 kept the events around EVERY launch in the timed region, which also switched the analysis' second stream
    # off: the timed step was a serialised, instrumented variant of the product (about 0.2 ms longer).
    ctx.set_profiling(True, only=dom)
    ctx.reset_profile()
    t0 = time.perf_counter()
    for i in range(steps):
        step(warmup + i)
    fence()
    dt = time.perf_counter() - t0
    prof_timed = ctx.profile()
    ctx.set_profiling(False)
    prof = {k: (v[0] * steps / survey_steps, v[1] * steps / survey_steps) for k, v in prof.items()}   # scaled to `steps` steps
    if dom in prof_timed:
        prof[dom] = prof_timed[dom]                      # the dominant kernel: as measured inside the timed region
    rank_ms = gather_rank_times(dt / steps * 1e3, rank, world)           # every rank's own ms per step (imbalance shows here)
    dt, frames_all = reduce_timing(dt, U * NFRM * steps, dev)   # MAX over ranks, SUM of frames

    # parity guard inside the bench: outputs finite and energy-preserving
    y = b.download(llsm.A_Y)[: b.y_off[1]]
    ok = bool(np.all(np.isfinite(y)) and abs(np.sqrt(np.mean(y[2000:40000] ** 2)) /
                                             np.sqrt(np.mean(x[0, 2000:40000] ** 2)) - 1) < 0.05)

    # PCIe-inclusive step (SURVEY 8d's wall-clock definition): page-locked upload of x / f0, compute, page-locked
    # download of every parameter row and the three waveforms.  Sub-batches on their own contexts (streams) driven by
    # host threads, so the transfers of one overlap the kernels of the others -- the arrangement the library's own
    # fan-out (llsm_gpu_set_fanout, csrc/capi.cpp) uses.  Reported beside `value`, never as it.
    e2e = None
    if full and not args.no_e2e:
        e2e = measure_e2e(args, llsm, local, ao, so, x, f0, U, fence, lambda dte, n: reduce_timing(dte, U * NFRM * n, dev))

    out = None
    if rank == 0:
        value = frames_all / dt
        tot_ms = sum(v[0] for v in prof.values())
        traffic, traffic_file = pmc_traffic()
        fb = [frame_alg(f) for f in f0s]
        F_alg = sum(a for a, _ in fb) / len(fb)
        B_alg = sum(bb for _, bb in fb) / len(fb)
        F_alg_literal = sum(frame_alg(f, literal=True)[0] for f in f0s) / len(f0s)
        # the same accounting summed over the kernels that are priced in flops (what the per-kernel objects use)
        kflop = 0.0
        for kname, (kms, klaunches) in prof.items():
            kind, work = kernel_alg(kname, U, f0s)
            if kind == "flop":
                kflop += work * klaunches / steps

        def roof_of(name):
            ms, launches = prof[name]
            avg_s = ms / launches * 1e-3
            kind, work = kernel_alg(name, U, f0s)
            tr = traffic.get(name)
            r = {"kernel": name, "avg_launch_ms": ms / launches, "launches_per_step": launches / steps,
                 "share_of_gpu_time": ms / tot_ms, "traffic": tr}
            if kind == "flop":
                ach = work / avg_s / 1e12
                # "mfma": the kernel's sums run on the matrix pipe (f32 MFMA); "fp32-vector": it issues no MFMA at all --
                # butterflies, filters and elementwise work on the VALU, whose fp32 peak on this part is the same 157.3 TFLOP/s
                r.update({"bound": "mfma" if name in MFMA_KERNELS else "fp32-vector", "achieved": ach, "peak": PEAK_FP32_TFLOPS,
                          "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS, "algorithmic_gflop_per_launch": work / 1e9})
                if name in FFT_KERNELS:
                    r.update({"frac_8d_literal": 2.0 * ach / PEAK_FP32_TFLOPS,
                              "note": "real-input transforms at 2.5 N log2 N (the accounting of every flop figure in this line); "
                                      "frac_8d_literal = the same at SURVEY 8(d)'s 5 N log2 N per real frame"})
                if name == "k_harm_speech_tile":
                    # The algorithmic count is the direct real-input DFT (4 flops per sample and bin); the kernel folds the
                    # window about its centre (E cos - j O sin) and so EXECUTES half of it on the MFMA, padded to whole
                    # tiles: `frac` can pass 1, `executed_frac` is what the matrix pipe really did.
                    ex = 0.0
                    for f0 in f0s:
                        hw, nh = plan(f0)
                        ex += NFRM * ((hw // 2 + 4) // 4) * 2 * ((nh + 15) // 16) / 16.0 * 2048.0
                    r.update({"executed_gflop_per_launch": ex / 1e9, "executed_frac": ex / avg_s / 1e12 / PEAK_FP32_TFLOPS,
                              "note": "frac prices the direct real-input DFT (algorithmic); the even/odd fold executes half of it, "
                                      "executed_frac = MFMA flops issued / time / peak"})
            elif kind == "byte":
                ach = work / avg_s / 1e9
                r.update({"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                          "frac": ach / PEAK_HBM_GBS, "algorithmic_bytes_per_launch": work,
                          "traffic_ratio": (tr / work) if tr else None})
            else:
                r.update({"bound": "hbm", "achieved": None, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": None})
            return r

        roof = roof_of(dom)
        roof.update({
            "timing": f"HIP events around this kernel's launches inside the timed region ({steps} steps); the other kernels' "
                      f"times (roofline_other_kernels, kernels_ms_per_step, share_of_gpu_time) from an untimed survey pass of "
                      f"{survey_steps} steps with events around every launch, scaled to {steps} steps",
            "achieved_fp32": value * F_alg / (PEAK_FP32_TFLOPS * 1e12 * world),
            "achieved_fp32_8d_literal": value * F_alg_literal / (PEAK_FP32_TFLOPS * 1e12 * world),
            "achieved_fp32_kernels": kflop / (dt / steps) / (PEAK_FP32_TFLOPS * 1e12),
            "achieved_hbm": value * B_alg / (PEAK_HBM_GBS * 1e9 * world),
            "F_alg_flop_per_frame": F_alg, "F_alg_8d_literal_flop_per_frame": F_alg_literal,
            "kernel_flop_per_step": kflop, "B_alg_bytes_per_frame": B_alg, "traffic_source": traffic_file,
            "note": "whole path per GPU: achieved_fp32 = value x F_alg / 157.3 TFLOP/s with F_alg on the accounting of the "
                    "per-kernel objects (a real-input DFT bin = 4 flops per sample, a resynthesised sample = 2 flops per "
                    "harmonic); achieved_fp32_8d_literal = the same with SURVEY 8(d)'s printed 8 / 4 flops (complex x "
                    "complex: over-counts a real-input transform 2x, kept for reference only); achieved_fp32_kernels = sum "
                    "of the flop-priced kernels' algorithmic work / step time (excludes the byte-priced streaming kernels); "
                    "achieved_hbm = value x B_alg / 8 TB/s (SURVEY 8d: the path is compute-bound, compulsory HBM traffic "
                    "cannot reach 40 % of 8 TB/s); dominant kernel priced on algorithmic work (unique bytes in + out for "
                    "streaming kernels), traffic = rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this command"})
        others = [roof_of(k) for k, _ in sorted(prof.items(), key=lambda kv: -kv[1][0]) if k != dom][:8]
        out = {"metric": "frames/sec (layer0 analyze+synth, 44.1 kHz, 5 ms hop)", "value": value,
               "unit": "frames/s", "n_gpus": world, "steps": steps, "warmup": warmup,
               "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{U} synthetic 1 s utterances per GPU, F0 "
                                      f"{'120 Hz fixed' if workload == 'fixed120' else '80-400 Hz log sweep'}"
                                      ", 44.1 kHz, 5 ms hop, layer0 analyze+synth, default options, f0_refine=0",
                          "utterances_per_gpu": U, "frames_per_utterance": NFRM, "parallelism": f"dp{world}"},
               "roofline": roof, "roofline_other_kernels": others,
               "kernels_ms_per_step": {k: v[0] / steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
               "rank_ms_per_step": rank_ms, "rank_ms_spread": (max(rank_ms) / min(rank_ms)) if min(rank_ms) > 0 else None,
               "backend": (placement or {}).get("backend"),      # "nccl" (= RCCL) | "gloo" (only on explicit request) | None at N = 1
               "partition": "strided (utterance u -> rank u mod N)",
               "value_e2e": e2e, "sanity_ok": ok, "placement": placement}
        if not full:                                   # a leg of `other_workloads`: the headline figures only
            out = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "config",
                                       "kernels_ms_per_step", "rank_ms_per_step", "rank_ms_spread", "partition", "sanity_ok")}
            out["roofline_frac_whole_path_fp32"] = roof["achieved_fp32"]
    b.close()
    ctx.close()
    return out, x


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--utts", type=int, default=1024, help="utterances per GPU")
    ap.add_argument("--workload", default="fixed120", choices=["fixed120", "sweep", "rt64", "rt64pbp", "l1"])
    ap.add_argument("--streams", type=int, default=64, help="rt64: llsmrt streams per GPU")
    ap.add_argument("--rt-hops", type=int, default=1, help="rt64*: hops per library call (llsm_rtsynth_group_feed_many when > 1)")
    ap.add_argument("--rt-pipeline", type=int, default=-1, help="rt64*: 1 / 0 = feeds return before the device has finished the hop (llsm_gpu_rt_pipeline) / synchronous feeds (default: library default = synchronous)")
    ap.add_argument("--rt-graph", type=int, default=-1, help="rt64*: 1 / 0 = one hipGraph launch per hop on / off (default: library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-other", action="store_true", help="skip the short sweep / rt64 / rt64pbp / l1 legs after the headline")
    ap.add_argument("--e2e-parts", type=int, default=0, help="value_e2e: sub-batches in flight (one context + host thread each); 0 = sweep 2 / 4 / 8 and keep the best")
    ap.add_argument("--e2e-steps", type=int, default=12, help="value_e2e: timed steps per repetition (after 2 warm-up steps)")
    ap.add_argument("--e2e-reps", type=int, default=3, help="value_e2e: repetitions (min / median / max reported)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="CPU-only check of the N-rank launch / reduction plumbing (gloo); no compute")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.launcher_selftest:
        sys.exit(launcher_selftest(args, world, rank))

    import torch
    import torch.distributed as dist
    import libllsm2_amd as llsm

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libllsm2_amd has no CPU fallback")
    if os.environ.get("LLSM_BENCH_SHARE_DEVICE") == "1":  # test hook: N ranks on the devices there are (1-GPU box, gloo)
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        # The only collectives of this bench are the barrier and the MAX / SUM of two scalars.  RCCL (backend "nccl")
        # is the default; if it cannot come up on this node ($LLSM_BENCH_BACKEND=gloo forces it) the same two
        # reductions run over gloo on host tensors -- the data path has no collective either way.
        # On a node with a device for every rank a failing RCCL is FATAL (non-zero exit): the first real N-GPU run
        # must not pass quietly with placement.backend = "gloo" (VERDICT r4 item 8).
        from libllsm2_amd.sharding import RcclUnavailable, init_timing_group
        try:
            backend_used = init_timing_group(rank, world, dev, log=lambda m: print("bench.py: " + m, file=sys.stderr, flush=True),
                                             strict=torch.cuda.device_count() >= world)
        except RcclUnavailable as e:
            raise SystemExit(f"bench.py: {e}")
        assert dist.get_world_size() == args.gpus
    else:
        backend_used = None
    os.environ["LLSM_GPU_DEVICE"] = str(local)
    # who runs where: every rank reports its device; with at least `world` devices on the node the ranks must sit on
    # DISTINCT ones (a launcher that hands two ranks one LOCAL_RANK would otherwise pass as "2 GPUs")
    from libllsm2_amd.sharding import gather_rank_devices
    rank_devices = gather_rank_devices(rank, world, local)
    if world > 1 and torch.cuda.device_count() >= world:
        ids = {(d["pci_bus_id"], d["uuid"], d["device"]) for d in rank_devices}
        if len(ids) != world:
            raise SystemExit(f"bench.py: {world} ranks but only {len(ids)} distinct devices: {rank_devices}")
    placement = {"world_size": world, "backend": backend_used, "devices_on_node": torch.cuda.device_count(),
                 "ranks": rank_devices}

    if args.workload in ("rt64", "rt64pbp"):
        res = bench_rt(args, llsm, world, rank, local, dev, dist, placement)
    elif args.workload == "l1":
        res = bench_l1(args, llsm, world, rank, local, dev, dist, placement)
    else:
        res, x = bench_layer0(args, llsm, world, rank, local, dev, dist, placement, args.workload, args.steps, args.warmup)
        # BASELINE.json configs[2-4] inside the driver-timed run: short legs after the headline's timed region (a few
        # seconds in all; `--no-other` skips them).  With several ranks only the sweep leg runs (it IS the multi-GPU
        # config); the llsmrt and layer-1 legs are single-GPU configs.
        if args.workload == "fixed120" and not args.no_other:
            others = {}
            t_legs = time.perf_counter()
            r, _ = bench_layer0(args, llsm, world, rank, local, dev, dist, placement, "sweep", 3, 1, full=False)
            others["sweep"] = r
            if world == 1:
                # llsmrt: synchronous feeds (the reference's semantics: a feed returns with its samples in the rings) and,
                # as keys of their own, pipelined feeds (one hop of extra latency, the host side beside the device)
                for wl, pipe in (("rt64", 0), ("rt64pbp", 0), ("rt64", 1), ("rt64pbp", 1)):
                    r = bench_rt(args, llsm, world, rank, local, dev, dist, None, workload=wl, steps=5, warmup=3, pipeline=pipe)   # 1000 hops (~50 ms): 400 were at the mercy of one host hiccup
                    others[wl + ("_pipelined" if pipe else "")] = {k: r[k] for k in (
                        "metric", "value", "unit", "steps", "ms_per_step", "ms_per_hop", "max_pull_ms",
                        "realtime_factor_per_stream", "pipelined_feeds", "config")}
                # llsmrt capacity: how many streams one GPU carries before the hop time moves --
                # 64 ... 1024 streams per group, synchronous and pipelined feeds, harmonic-model path, 600 hops each
                for wl in ("rt64", "rt64pbp"):           # four hops per call (feed_many): synchronous semantics, the host beside the device
                    r = bench_rt(args, llsm, world, rank, local, dev, dist, None, workload=wl, steps=5, warmup=3, pipeline=0, hops_per_feed=4)
                    others[wl + "_feed_many4"] = {k: r[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "ms_per_hop", "max_pull_ms",
                                                                    "realtime_factor_per_stream", "hops_per_feed", "config")}
                cap = {}
                for ns in (64, 128, 256, 512, 1024):
                    for pipe in (0, 1):
                        r = bench_rt(args, llsm, world, rank, local, dev, dist, None, workload="rt64", steps=3, warmup=1, pipeline=pipe, streams=ns)
                        cap[f"{ns}{'_pipelined' if pipe else ''}"] = {"frames_per_s": r["value"], "us_per_hop": r["ms_per_hop"] * 1e3,
                                                                      "max_pull_ms": r["max_pull_ms"],
                                                                      "realtime_factor_per_stream": r["realtime_factor_per_stream"]}
                others["rt_capacity"] = {"metric": "llsmrt harmonic-model path: frames/s and microseconds per hop against the number of "
                                                   "lock-stepped streams in one group on one GPU", "streams": cap}
                r = bench_l1(args, llsm, world, rank, local, dev, dist, None, steps=3, warmup=1, x=x)
                others["l1"] = {k: r[k] for k in ("metric", "value", "unit", "steps", "ms_per_step", "config", "kernels_ms_per_step",
                                                  "gpu_ms_per_step", "host_ms_per_step", "host_ms_of_each_step", "sanity_ok")}
            if rank == 0:
                res["other_workloads"] = others
                res["other_workloads_wall_s"] = time.perf_counter() - t_legs
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()

if __name__ == "__main__":
    main()
