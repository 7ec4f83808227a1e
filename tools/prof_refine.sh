cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_refine -- python3 $GRAFT_REPO_ROOT/tools/prof_refine.py > /tmp/prof_refine.log 2>&1
f=$(ls /tmp/prof_refine/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms %.1f" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%8.1f ms %5.1f%% %6s  %s" % (float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, r["Calls"], r["Name"][:110]))
PY
