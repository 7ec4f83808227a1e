cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ahds > /tmp/ks.log 2>&1
grep -h "gip_" /tmp/ks/*/*_kernel_stats.csv | sed 's/(.*)"/"/' | cut -c1-110
