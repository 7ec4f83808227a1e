#!/usr/bin/env python3
"""Chronological kernel list of ONE steady-state step from a rocprofv3 rocpd database, plus a per-(kernel, grid) table.
usage: dump_step.py results.db [marker_kernel_substring] > step_trace.txt
Columns: start (us from the step's first kernel), duration (us), queue, grid x workgroup, LDS bytes, name."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "gip_preprocess_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]


def col(*names):
    for n in names:
        if n in cols:
            return n
    return "0"


grid, wg, lds, queue = col("grid_x", "grid_size_x", "grid_size"), col("workgroup_x", "workgroup_size_x", "workgroup_size"), \
    col("lds_size", "lds_block_size", "group_segment_size"), col("queue_id", "queue", "stream_id")
marks = [r[0] for r in db.execute("select start from kernels where name like ? order by start", ("%" + marker + "%",))]
lo, hi = marks[-2], marks[-1]
rows = db.execute("select start, end, %s, %s, %s, %s, name from kernels where start>=? and start<? order by start" % (queue, grid, wg, lds),
                  (lo, hi)).fetchall()
agg = {}
for s, e, q, g, w, l, n in rows:
    a = agg.setdefault((n[:70], g, w), [0, 0])
    a[0] += e - s
    a[1] += 1
print("# per (kernel, grid, workgroup): total us, launches")
for (n, g, w), (ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:120]:
    print("%9.1f %4d  grid %8s wg %5s  %s" % (ns / 1e3, c, g, w, n))
print("# chronological: start us, duration us, queue, grid, workgroup, lds, name")
for s, e, q, g, w, l, n in rows:
    print("%9.1f %8.1f q%-3s %8s %5s %6s  %s" % ((s - lo) / 1e3, (e - s) / 1e3, q, g, w, l, n[:90]))
