#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd /tmp; export TMPDIR=/tmp; mkdir -p $R/gpurun_out
for c in plain import init tensor stream; do
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/d2hpy -- python3 $R/tools/d2h_py.py $c 2>&1 | grep "^case\|^ok\|Error\|error" | head -5
  for f in $(find $R/gpurun_out/d2hpy -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do grep -v '^"Name"' $f | grep -i "copy" | cut -d, -f1-4; done
  rm -rf $R/gpurun_out/d2hpy
done
