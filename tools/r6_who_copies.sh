#!/bin/bash
# Which HIP calls are behind the few __amd_rocclr_copyBuffer / fillBuffer dispatches a bench run still shows
# (kernel trace joined with the HIP runtime trace on the correlation id).  Output: gpurun_out/r6_who_copies.txt
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf /tmp/who; mkdir -p /tmp/who
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d /tmp/who -o who -- python3 bench.py --steps 3 --warmup 1 > /tmp/who/bench.log 2>&1
python3 - <<'P' > gpurun_out/r6_who_copies.txt 2>&1
import csv, glob, collections
k = glob.glob('/tmp/who/**/*kernel_trace.csv', recursive=True)
h = glob.glob('/tmp/who/**/*hip_api_trace.csv', recursive=True)
print(k, h)
api = {}
rows = list(csv.DictReader(open(h[0])))
print('hip api columns', list(rows[0].keys()))
for r in rows:
    api[r['Correlation_Id']] = r
kr = list(csv.DictReader(open(k[0])))
print('kernel columns', list(kr[0].keys()))
kr.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(kr[0]['Start_Timestamp'])
first_trace = next(i for i, r in enumerate(kr) if 'k_trace' in r['Kernel_Name'])
print('dispatches', len(kr), 'first k_trace at index', first_trace)
for i, r in enumerate(kr):
    if 'rocclr' in r['Kernel_Name']:
        a = api.get(r['Correlation_Id'], {})
        prev = kr[i-1]['Kernel_Name'][:40] if i else ''
        print(i, r['Kernel_Name'], 'grid', r.get('Grid_Size_X', r.get('Grid_Size')), 't_ms', (int(r['Start_Timestamp'])-t0)/1e6,
              'api', a.get('Function'), 'tid', a.get('Thread_Id'), 'after', prev)
c = collections.Counter(r['Function'] for r in rows if 'emcpy' in r['Function'] or 'emset' in r['Function'])
print(c)
P
tail -2 /tmp/who/bench.log | cut -c1-300 >> gpurun_out/r6_who_copies.txt
