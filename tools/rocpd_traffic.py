"""Per-kernel HBM traffic per launch from two rocprofv3 PMC databases (FETCH_SIZE pass,
WRITE_SIZE pass): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction,
MI355X_MICROARCH.md section HBM).  Prints JSON {kernel: {...}}."""
import json, sqlite3, sys

def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    kn = [c for c in cols if "kernel" in c and "name" in c][0]
    cn = [c for c in cols if c in ("counter_name", "name")][0]
    vn = [c for c in cols if c in ("value", "counter_value")][0]
    out = {}
    for n, v, c in cur.execute(f"select {kn}, avg({vn}), count(*) from counters_collection where {cn}=? group by {kn}", (counter,)):
        full = n.split("(")[0].replace("void ", "").strip()
        k = full.split("<")[0].strip()
        # the FIX instantiation of the spectrogram envelope (k_spgm_env_wf<LOGN, LOGF, true>: a few dozen wavefronts redo the
        # listed pairs) is a launch of its own: keyed apart, or its 7 MB replace the 620 MB of the launch that does the work
        if k == "k_spgm_env_wf" and full.replace(" ", "").endswith(",true>"):
            k = "k_spgm_env_fix"
        out[k] = (v, c)
    return out

f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith("k_"):
        continue
    fk, wk = f.get(k, (0, 0))[0], w.get(k, (0, 0))[0]
    res[k] = {"FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
              "launches_sampled": f.get(k, (0, 0))[1]}
print(json.dumps(res, indent=1))
