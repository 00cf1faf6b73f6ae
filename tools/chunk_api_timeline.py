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

# ---- copies (rocprofv3 --memory-copy-trace beside --kernel-trace): the link's share of the same span
def union_ms(iv):
    ev2 = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
    tot = 0; d = 0; last2 = ev2[0][0] if ev2 else 0
    for t, k in ev2:
        if d >= 1: tot += t - last2
        d += k; last2 = t
    return tot / 1e6


try:
    names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table', 'view') and name like '%memory_cop%'")]
    view = "memory_copies" if "memory_copies" in names else (names[0] if names else None)
    if view:
        mc = [r[1] for r in cur.execute(f"pragma table_info({view})")]
        s_c = "start" if "start" in mc else [c for c in mc if "start" in c][0]
        e_c = "end" if "end" in mc else [c for c in mc if "end" in c][0]
        b_c = [c for c in mc if c in ("size", "bytes", "size_bytes")] or [c for c in mc if "size" in c or "byte" in c]
        n_c = [c for c in mc if c in ("name", "direction", "kind")] or [c for c in mc if "name" in c]
        cp = cur.execute(f"select {n_c[0] if n_c else 'NULL'}, {s_c}, {e_c}, {b_c[0] if b_c else '0'} from {view} order by {s_c}").fetchall()
        cp = [r for r in cp if r[1] >= t0 and r[2] <= t1 + 10_000_000]
        by = {}
        for n, s, e, sz in cp:
            a = by.setdefault(str(n), [0, 0.0, 0]); a[0] += 1; a[1] += (e - s) / 1e6; a[2] += int(sz or 0)
        print("copies inside the span (%s: %s):" % (view, ", ".join(mc)[:200]))
        for n, (c, t, sz) in sorted(by.items(), key=lambda kv: -kv[1][1]):
            print("  %-40s %6d %9.2f ms %9.1f MB  %6.1f GB/s while copying" % (n[:40], c, t, sz / 1e6, sz / 1e9 / (t / 1e3) if t else 0))
        big = [(s, e) for n, s, e, sz in cp if (sz or 0) >= (1 << 20)]
        kern = [(s, e) for _, s, e in rows]
        print("link busy with copies of >= 1 MB (union) %.1f ms; kernels or such copies (union) %.1f ms; span %.1f ms" %
              (union_ms(big), union_ms(big + kern), (t1 - t0) / 1e6))
    else:
        print("(no memory-copy table in this trace)")
except sqlite3.Error as e:
    print("(memory copies not summarised:", e, ")")
