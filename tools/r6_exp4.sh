#!/bin/bash
# round 6, GPU call 4: kernel-trace of config 2 under the blit copies and under the library's own copy kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out/r6_exp4.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp; export TMPDIR=/tmp
C2="--workload config2_100k_400x200_1pass --no-cpu-baseline --no-extras --steps 100 --warmup 5"
export RR_BENCH_LIVE_TIMING=0
prof() { local label=$1; shift
  rm -rf /tmp/kt; env "$@" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $R/bench.py $C2 > /tmp/kt_bench.json 2> /tmp/kt.log
  echo "== $label: $(grep -o '"value": [0-9.]*' /tmp/kt_bench.json | head -1)" >> $O
  f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-60,60-200 | awk -F'","' '{print $1 " calls=" $2 " avg_ns=" $4}' >> $O
}
prof "blit (hipMemcpyAsync)" RR_FLUSH_KERNEL=0
prof "own kernel xcd=3 blocks=16 inflight=2" RR_FLUSH_XCD=3 RR_FLUSH_BLOCKS=16 RR_FLUSH_INFLIGHT=2
prof "own kernel unconfined t=512 blocks=16 inflight=0" RR_FLUSH_XCD=-1 RR_FLUSH_THREADS=512 RR_FLUSH_BLOCKS=16 RR_FLUSH_INFLIGHT=0
prof "own kernel NT unconfined t=512 blocks=16" RR_FLUSH_NT=1 RR_FLUSH_XCD=-1 RR_FLUSH_THREADS=512 RR_FLUSH_BLOCKS=16 RR_FLUSH_INFLIGHT=0
cat $O
