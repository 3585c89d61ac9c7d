cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf /tmp/prc; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prc -- python tools/rcda_bench.py 10 > /dev/null 2>&1
f=$(find /tmp/prc -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rcda' in r['Name'] or 'igemm' in r['Name'] or 'wgrad' in r['Name']:
        print("  %-50s calls %4s avg %7.1f min %7.1f max %7.1f us" % (r['Name'].replace('(anonymous namespace)::','')[:50], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
