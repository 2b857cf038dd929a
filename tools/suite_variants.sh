#!/bin/bash
# GPU box helper: the GPU suite under the library's environment switches (every code path a switch selects must give the same images)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
for v in RR_LANES=1 RR_STACK_LDS=4 RR_CULL_POP=0 RR_PASS0_AZ=4 RR_GRAPHS=0 RR_TIGHT_GRID=0 RR_TIGHT_FORCE=1 RR_BVH_CHOOSE=0 RR_COPY_BLOCKS=0 RR_TRACE_CHUNK=0 RR_TRACE_CHUNK=64; do
  echo "== $v: $(env $v timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1)"
done
