#!/usr/bin/env python3
"""Memory copies (rocprofv3 --memory-copy-trace) interleaved with the kernels of ONE steady-state step: which copies sit
in the stream between the kernels, how long they take and what gap they leave.
usage: copies_in_step.py results.db [marker_kernel_substring] [from_us to_us]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "gip_preprocess_kernel"
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
mc = [v for v in views if "memory_cop" in v and not v.startswith("rocpd_")] or [v for v in views if "memory_cop" in v]
print("# copy tables / views:", mc)
if not mc:
    sys.exit(0)
mcv = mc[0]
cols = [r[1] for r in db.execute("pragma table_info(%s)" % mcv)]
print("# columns:", cols)
marks = [r[0] for r in db.execute("select start from kernels where name like ? order by start", ("%" + marker + "%",))]
lo, hi = marks[-2], marks[-1]
size = "size" if "size" in cols else ("bytes" if "bytes" in cols else "0")
name = "name" if "name" in cols else ("kind" if "kind" in cols else "''")
rows = [(s, e, "K", n[:80], 0) for s, e, n in db.execute("select start, end, name from kernels where start>=? and start<?", (lo, hi))]
rows += [(s, e, "C", str(n), b) for s, e, n, b in db.execute("select start, end, %s, %s from %s where start>=? and start<?" % (name, size, mcv), (lo, hi))]
rows.sort()
a = float(sys.argv[3]) if len(sys.argv) > 4 else None
b = float(sys.argv[4]) if len(sys.argv) > 4 else None
ncopy = sum(1 for r in rows if r[2] == "C")
print("# %d kernels, %d copies in the step (%.2f ms)" % (len(rows) - ncopy, ncopy, (hi - lo) / 1e6))
prev_end = lo
for s, e, k, n, by in rows:
    t = (s - lo) / 1e3
    if k == "C" or (a is not None and a <= t <= b):
        print("%9.1f %7.1f  gap %7.1f  %s %s %s" % (t, (e - s) / 1e3, (s - prev_end) / 1e3, k, n, ("%d B" % by) if k == "C" else ""))
    prev_end = max(prev_end, e)
