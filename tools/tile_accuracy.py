"""Accuracy of the shared-F0 tile kernel against the float64 oracle on fixed-F0 utterances (one process per build:
LLSM_AMD_LIB selects an experiment build of tools/kbench.py --build).
    python tools/tile_accuracy.py [--variants HT_SEG=23 ...]"""
import argparse, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def child():
    import numpy as np
    import libllsm2_amd as llsm
    from conftest import FS, make_utterance
    from gpu_common import analysis_metrics, gpu_analyze, oracle_analyze
    from oracle.oracle import Oracle
    o = Oracle(np.float64)
    ctx = llsm.Context(0)
    ao = llsm.make_aoptions(f0_refine=0)
    f0v = [80.0, 97.3, 120.0, 155.5, 199.7, 263.1, 400.0]
    xs = [make_utterance(40 + k, f, nx=22050) for k, f in enumerate(f0v)]
    f0s = [np.full(100, f, np.float32) for f in f0v]
    out = {}
    for tiles in (1, 0):
        llsm.load().llsm_gpu_shared_f0_tiles(tiles)
        b, g, xres = gpu_analyze(ctx, ao, FS, xs, f0s)
        for u, (x, f0) in enumerate(zip(xs, f0s)):
            pr, xr = oracle_analyze(o, ao, FS, x, f0)
            sl = slice(b.frm_off[u], b.frm_off[u + 1])
            m = analysis_metrics(g, sl, pr, xres[b.x_off[u]:b.x_off[u + 1]], xr)
            # where the worst relative error sits
            a_g, a_o = g[llsm.A_AMPL][sl].astype(np.float64), pr.ampl
            big = a_o > 1e-4 * a_o.max()
            rel = np.where(big, np.abs(a_g - a_o) / np.maximum(a_o, 1e-30), 0)
            fr, h = np.unravel_index(np.argmax(rel), rel.shape)
            out[f"tiles{tiles}_f0_{f0v[u]}"] = [float("%.3g" % m["ampl_rel_max"]), float("%.3g" % m["ampl_abs_over_max"]),
                                                float("%.3g" % m["phse_max_rad"]), int(fr), int(h), float("%.3g" % (a_o[fr, h] / a_o.max()))]
        b.close()
    print(json.dumps(out))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", nargs="*", default=[])
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        child(); sys.exit(0)
    for d in [None] + a.variants:
        env = dict(os.environ, PYTHONPATH=ROOT)
        if d:
            env["LLSM_AMD_LIB"] = os.path.join(ROOT, "exp_build", f"lib_{d.replace('=', '_').replace(',', '+')}.so")
        r = subprocess.run([sys.executable, __file__, "--child"], env=env, capture_output=True, text=True)
        print(d or "base", "[ampl_rel_max, ampl_abs/max, phase_max, worst frame, harmonic, its ampl/max]")
        try:
            for k, v in json.loads(r.stdout.strip().splitlines()[-1]).items():
                print("   ", k, v)
        except Exception:
            print(r.stdout[-500:], r.stderr[-800:])
