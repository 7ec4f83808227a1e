cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python3 $GRAFT_REPO_ROOT/tools/prof_vae.py > /tmp/pv.log 2>&1
f=$(ls /tmp/pv/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms per iteration %.2f" % (tot / 1e6 / 12))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%7.3f ms/it %5.1f%% %5d  %s" % (float(r["TotalDurationNs"]) / 1e6 / 12, 100 * float(r["TotalDurationNs"]) / tot, int(r["Calls"]) // 12, r["Name"][:120]))
PY
