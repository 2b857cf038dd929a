#!/bin/bash
# GPU box helper: tools/d2h_engine.hip in its three ordering modes, under the runtime's copy-engine switches, and one
# rocprofv3 trace per mode: is the copy a blit KERNEL (__amd_rocclr_copyBuffer) or an SDMA transfer (MEMORY_COPY)?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
B=$R/build/d2h_engine
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 tools/d2h_engine.hip -o $B
run() { m=$1; shift; echo "== $m $*"; env "$@" $B $m | grep -v "^check: ok"; }
for m in plain samestream event; do run $m X=0; done
run plain GPU_FORCE_BLIT_COPY_SIZE=1048576
run plain HSA_ENABLE_SDMA=0
run samestream GPU_FORCE_BLIT_COPY_SIZE=0
cd /tmp; export TMPDIR=/tmp
for m in plain samestream event; do
  rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/d2h_prof -- $B $m > /dev/null 2>&1
  echo "-- rocprofv3, mode $m"
  for f in $(find $R/gpurun_out/d2h_prof -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do grep -v '^"Name"' $f | cut -d, -f1-4; done
  rm -rf $R/gpurun_out/d2h_prof
done
