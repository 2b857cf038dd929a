#!/bin/bash
# GPU box helper: where the host time of an rr_multi call over 8 loopback device entries goes (RR_HOST_PROFILE=1)
# usage: tools/multi_hostprof.sh [config id = 2] [frames]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
CFG=${1:-2}; FR=${2:-8000}
bash tools/cpp_bench.sh 160 8 multi $CFG > gpurun_out/hostprof_build.log 2>&1
CPP_BENCH_NDEV=8 RR_MULTI_LOOPBACK=1 RR_MULTI_THREADS=0 /tmp/cpp_bench /tmp/c$CFG.bin $FR 8 multi 2>&1 | grep -v "^rr_simulate_batch_host_async"
CPP_BENCH_NDEV=8 RR_MULTI_LOOPBACK=1 RR_MULTI_THREADS=0 RR_HOST_PROFILE=1 /tmp/cpp_bench /tmp/c$CFG.bin $FR 8 multi 2>&1 | grep "host profile\|host CPU"
