#!/bin/bash
# GPU box helper: which engine carries D2H copies -- the system runtime (/opt/rocm) against the one bundled with the torch wheel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
B=$R/build/d2h_engine
T=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
ls -la $T/libamdhip64.so $T/libhsa-runtime64.so /opt/rocm/lib/libamdhip64.so.7* /opt/rocm/lib/libhsa-runtime64.so.1* 2>&1 | awk '{print $5, $9, $10, $11}'
echo "== system runtime"; $B plain | grep -v "^check: ok"
echo "== torch's runtime (LD_PRELOAD)"; LD_PRELOAD="$T/libhsa-runtime64.so $T/libamdhip64.so" $B plain | grep -v "^check: ok"
cd /tmp; export TMPDIR=/tmp
LD_PRELOAD="$T/libhsa-runtime64.so $T/libamdhip64.so" rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/d2hw_prof -- $B plain > gpurun_out/d2hw_rp.log 2>&1
echo "-- rocprofv3, torch's runtime"; for f in $(find $R/gpurun_out/d2hw_prof -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do grep -v '^"Name"' $f | cut -d, -f1-4; done
rm -rf $R/gpurun_out/d2hw_prof
cd $R
# the library's own host delivery from a pure C++ caller (system runtime): rr_multi over one device = rr_simulate_batch_host_async
bash tools/cpp_bench.sh 1600 8 multi 2 > gpurun_out/d2hw_cpp.log 2>&1; tail -4 gpurun_out/d2hw_cpp.log
cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/d2hw_prof_cpp -- /tmp/cpp_bench /tmp/c2.bin 1600 8 multi > /dev/null 2>&1
echo "-- C++ caller, multi"; for f in $(find $R/gpurun_out/d2hw_prof_cpp -name "*kernel_stats.csv" -o -name "*memory_copy_stats.csv"); do grep -v '^"Name"' $f | grep -i "copy\|COPY" | cut -d, -f1-4; done
rm -rf $R/gpurun_out/d2hw_prof_cpp
