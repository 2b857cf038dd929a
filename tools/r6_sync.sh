#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r6_sync.txt
python tools/probe_sync_frame.py 300 4 2>/dev/null | tail -1 > $out
rm -rf /tmp/syncp; rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/syncp -o sp -- python3 tools/probe_sync_frame.py 300 4 > /tmp/syncp.log 2>&1
tail -1 /tmp/syncp.log >> $out
python3 - >> $out <<'P'
import csv, glob
f = glob.glob('/tmp/syncp/**/*kernel_stats.csv', recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    n = int(r['Calls']); avg = float(r['AverageNs']) / 1e3
    if n >= 300 and 'k_' in r['Name']:
        per = n / 320.0
        print('  %-70s %5.1f launches/frame x %7.1f us = %7.1f us' % (r['Name'][:70], per, avg, per * avg)); tot += per * avg
print('  kernels per frame: %.1f us' % tot)
for f in glob.glob('/tmp/syncp/**/*memory_copy_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)): print('  ', r['Name'], r['Calls'], 'avg us', float(r['AverageNs']) / 1e3)
P
cat $out
