#!/bin/bash
# GPU box helper: host delivery from a pure C++ caller (system runtime: D2H copies go over SDMA) -- the trickle (images ride on
# the next batch's trace launches) against plain copies (RR_COPY_BLOCKS=0), alternating
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
CFG=${1:-4}; FR=${2:-4800}
bash tools/cpp_bench.sh 160 8 multi $CFG > gpurun_out/d2h_cpp_build.log 2>&1
for i in 1 2 3; do for cb in 8 0; do
  echo "RR_COPY_BLOCKS=$cb: $(RR_COPY_BLOCKS=$cb /tmp/cpp_bench /tmp/c$CFG.bin $FR 8 multi 2>&1 | grep -o "rr_simulate_batch_host_async[^:]*: [0-9]* images/s\|over 1 device entry[^:]*: [0-9]* images/s" | tr '\n' ' ')"
done; done
