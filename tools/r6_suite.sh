#!/bin/bash
# round 6: the whole GPU suite (+ the C++ host demo it builds) on the current tree
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r6_suite.log 2>&1; echo "suite rc=$? $(tail -1 gpurun_out/r6_suite.log)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
