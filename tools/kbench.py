"""Build the variants first (here, no GPU needed):  python tools/kbench.py --build --ablate HT_SCHED=1 HT_SCHED=1,HT_WPE=4
Kernel-level timing of one analysis+synthesis step for one or more builds of the library
(experiment helper: ablation builds via `-D`, selected with LLSM_AMD_LIB).

    python tools/kbench.py [--utts 512] [--kernels k_spgm_env,...] [--ablate SPGM_ABLATE=1 ...]
"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def run_one(utts, steps, thop=0.005, jitter=False, tiles=-1):
    import numpy as np
    import libllsm2_amd as llsm
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import make_utterance, FS
    ctx = llsm.Context(0)
    llsm.load().llsm_gpu_analysis_overlap(0)         # per-kernel times of kernels running ALONE (events follow the launch's stream since round 6)
    xs = [make_utterance(u, 120.0) for u in range(4)]
    x = np.concatenate([xs[u % 4] for u in range(utts)])
    nfrm = int(round(1.0 / thop))
    f0 = np.full(nfrm * utts, 120.0, np.float32)
    if jitter:                                       # every frame its own F0: nothing qualifies for the shared-F0 tiles
        f0 = (f0 * (1.0 + 1e-3 * np.random.default_rng(1).standard_normal(len(f0)))).astype(np.float32)
    if tiles >= 0:
        llsm.load().llsm_gpu_shared_f0_tiles(tiles)
    b = llsm.Batch(ctx, llsm.make_aoptions(f0_refine=0, thop=thop), FS, [44100] * utts, [nfrm] * utts)
    b.upload(llsm.A_X, x); b.upload(llsm.A_F0, f0)
    so = llsm.make_soptions(FS)
    b.analyze(); b.synthesize(so, seed=1); ctx.sync()
    ctx.set_profiling(True); ctx.reset_profile()
    for i in range(steps):
        b.analyze(); b.synthesize(so, seed=2 + i)
    ctx.sync()
    prof = ctx.profile()
    print(json.dumps({k: round(v[0] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--ablate", nargs="*", default=[])
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--build", action="store_true", help="compile the --ablate variants into exp_build/ and exit")
    ap.add_argument("--thop", type=float, default=0.005)
    ap.add_argument("--jitter", action="store_true", help="per-frame F0 jitter (no shared-F0 tiles form)")
    ap.add_argument("--tiles", type=int, default=-1, help="llsm_gpu_shared_f0_tiles(0 / 1); default: library default")
    a = ap.parse_args()
    if a.child:
        run_one(a.utts, a.steps, a.thop, a.jitter, a.tiles); sys.exit(0)
    if a.build:
        from concurrent.futures import ThreadPoolExecutor
        from libllsm2_amd import build as b
        os.makedirs(os.path.join(ROOT, "exp_build"), exist_ok=True)
        def one(d):
            # the timing experiments live in tools/kbench_experiments.h; the product sources only reach them through this define
            return b.build(defines=d.split(",") + ["LLSM_KBENCH_EXPERIMENTS"], out=os.path.join(ROOT, "exp_build", f"lib_{d.replace('=', '_').replace(',', '+')}.so"))
        with ThreadPoolExecutor(4) as ex:
            print(list(ex.map(one, a.ablate)))
        sys.exit(0)
    variants = [("base", None)] + [(d, d) for d in a.ablate]
    for name, d in variants:
        env = dict(os.environ, PYTHONPATH=ROOT)
        if d:
            env["LLSM_AMD_LIB"] = os.path.join(ROOT, "exp_build", f"lib_{d.replace('=', '_').replace(',', '+')}.so")
        r = subprocess.run([sys.executable, __file__, "--child", "--utts", str(a.utts), "--steps", str(a.steps), "--thop", str(a.thop), "--tiles", str(a.tiles)] + (["--jitter"] if a.jitter else []),
                           env=env, capture_output=True, text=True)
        print(name, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
