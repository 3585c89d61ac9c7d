cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; rm -rf /tmp/prm
cat > /tmp/mha_bench.py <<'PY'
import sys; sys.path.insert(0, '.')
import torch
from counting_detr_amd import ops
N, L, E, nh = 2, 300, 256, 8
qk = torch.randn(N, L, 2 * E, device="cuda"); v = torch.randn(N, L, E, device="cuda"); dO = torch.randn(N, L, E, device="cuda")
for _ in range(10):
    o, lse = ops.mha_fwd_raw(qk, v, nh)
    ops.mha_bwd_raw(qk, v, o, dO, lse, nh)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prm -- python /tmp/mha_bench.py > /dev/null 2>&1
f=$(find /tmp/prm -name "*kernel_stats.csv" | head -1); python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'flash' in r['Name']:
        print("  %-40s calls %4s avg %7.1f min %7.1f us" % (r['Name'].replace('(anonymous namespace)::','')[:40], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
