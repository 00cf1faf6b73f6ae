"""Device timeline of the object path from a rocprofv3 kernel trace (rocpd sqlite): per kernel name calls / total / average, the
union of busy time (any kernel running), and how much of it had two or more kernels running at once.
    python tools/chunk_api_timeline.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = cur.execute(f"select {name_col}, {start}, {end} from kernels order by {start}").fetchall()
t0, t1 = rows[0][1], max(r[2] for r in rows)
agg = {}
for n, s, e in rows:
    k = n.split("(")[0].replace("void ", "")[:40]
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
print("span %.1f ms, %d launches" % ((t1 - t0) / 1e6, len(rows)))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-42s %6d %9.2f ms  %8.1f us avg" % (k, c, t, t / c * 1e3))
ev = sorted([(s, 1) for _, s, e in rows] + [(e, -1) for _, s, e in rows])
busy = multi = 0; depth = 0; last = ev[0][0]
for t, d in ev:
    if depth >= 1: busy += t - last
    if depth >= 2: multi += t - last
    depth += d; last = t
print("sum of kernel durations %.1f ms, device busy (union) %.1f ms, of which >= 2 kernels at once %.1f ms" %
      (sum(v[1] for v in agg.values()), busy / 1e6, multi / 1e6))
