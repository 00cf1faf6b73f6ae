"""Per-hop device timeline of an llsmrt run from a rocprofv3 rocpd database (kernel-trace): for the steady-state hops
(copy in, k_rt_front, k_rt_back, copy out; or k_rt_hop alone, whatever the run used) the average duration of each launch, the idle gaps between them, the span
of a hop on the device and the host-side gap from one hop's last launch to the next hop's first.

    python tools/rt_timeline.py <results.db>
"""
import sqlite3
import sys


def main(p):
    db = sqlite3.connect(p)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = cur.execute(f"select {name_col}, {start}, {end} from kernels order by {start}").fetchall()
    seq = [(n.split("(")[0].replace("void ", "").split("<")[0].replace("k_rt_hop2", "k_rt_hop"), s, e) for n, s, e in rows]
    hops = []
    i = 0
    names = [x[0] for x in seq]
    one = names.count("k_rt_hop") > names.count("k_rt_front")
    core = ["k_rt_hop"] if one else ["k_rt_front", "k_rt_back"]
    pat = ["__amd_rocclr_copyBuffer"] + core
    has_copies = sum(names[j:j + len(pat)] == pat for j in range(len(names) - len(pat))) > names.count(core[0]) // 2
    want = ["__amd_rocclr_copyBuffer"] + core + ["__amd_rocclr_copyBuffer"] if has_copies else core
    labels = ["copy in"] + core + ["copy out"] if has_copies else core
    nw = len(want)
    while i + nw <= len(seq):
        if names[i:i + nw] == want:
            hops.append(seq[i:i + nw]); i += nw
        else:
            i += 1
    if len(hops) < 10:
        print("too few steady-state hops:", len(hops)); return
    hops = hops[len(hops) // 4:]
    n = len(hops)
    dur = [sum(h[k][2] - h[k][1] for h in hops) / n / 1e3 for k in range(nw)]
    gap = [sum(h[k + 1][1] - h[k][2] for h in hops) / n / 1e3 for k in range(nw - 1)]
    span = sum(h[-1][2] - h[0][1] for h in hops) / n / 1e3
    between = [hops[j + 1][0][1] - hops[j][-1][2] for j in range(n - 1) if hops[j + 1][0][1] - hops[j][-1][2] < 1e6]
    period = [hops[j + 1][0][1] - hops[j][0][1] for j in range(n - 1) if hops[j + 1][0][1] - hops[j][0][1] < 1e6]
    print(f"steady-state hops: {n}")
    for k, nm in enumerate(labels):
        print(f"  {nm:12s} {dur[k]:7.2f} us" + (f"   then idle {gap[k]:6.2f} us" if k < nw - 1 else ""))
    print(f"  device span of a hop   {span:7.2f} us")
    print(f"  last launch of a hop -> first launch of the next (host)  {sum(between) / len(between) / 1e3:7.2f} us")
    print(f"  hop period             {sum(period) / len(period) / 1e3:7.2f} us")


if __name__ == "__main__":
    main(sys.argv[1])
