"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace [+ PMC]) as text:
per-kernel launches, total / average / min / max duration, share of GPU time,
and the per-launch average of every collected counter.

    python tools/rocpd_summary.py <results.db> [more.db ...]
"""
import sqlite3
import sys


def main(paths):
    for p in paths:
        db = sqlite3.connect(p)
        cur = db.cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
        start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
        rows = cur.execute(f"select {name_col}, count(*), sum({end}-{start}), avg({end}-{start}), "
                           f"min({end}-{start}), max({end}-{start}) from kernels group by {name_col} "
                           f"order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print(f"# {p}")
        print(f"{'kernel':48s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'share':>6s}")
        for n, c, t, a, mn, mx in rows:
            short = n.split("(")[0][:48]
            print(f"{short:48s} {c:6d} {t/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:10.2f} {mx/1e3:10.2f} {100*t/tot:5.1f}%")
        try:
            pm = cur.execute("select * from counters_collection limit 1").fetchall()
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            if pm:
                kn = [c for c in ccols if "kernel" in c and "name" in c][0] if any("kernel" in c and "name" in c for c in ccols) else None
                cn = [c for c in ccols if c in ("counter_name", "name")][0]
                vn = [c for c in ccols if c in ("value", "counter_value")][0]
                if kn:
                    print(f"{'kernel':48s} {'counter':>16s} {'avg/launch':>16s} {'launches':>9s}")
                    for n, cname, v, c in cur.execute(
                            f"select {kn}, {cn}, avg({vn}), count(*) from counters_collection group by {kn}, {cn} order by 1"):
                        print(f"{n.split('(')[0][:48]:48s} {cname:>16s} {v:16.1f} {c:9d}")
        except sqlite3.Error as e:
            print("# (no counter table:", e, ")")
        print()


if __name__ == "__main__":
    main(sys.argv[1:])
