#!/bin/bash
# round 6, GPU call 10: the capture / delivery conflict fixed -- tests, the three fuzz_batch runs of the campaign, benches
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out; O=gpurun_out/r6_exp10.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py tests/test_gpu_round5.py -x -q -m gpu > gpurun_out/r6_exp10_pytest.log 2>&1; echo "pytest rc=$? $(tail -1 gpurun_out/r6_exp10_pytest.log)" >> $O
RR_HOST_SDMA_VERBOSE=1 timeout 900 python tests/fuzz/fuzz_batch.py 500 55 > gpurun_out/r6_exp10_fb1.log 2>&1; echo "fuzz_batch 500 55: rc=$? $(tail -1 gpurun_out/r6_exp10_fb1.log) [rr] lines: $(grep -c '\[rr\]' gpurun_out/r6_exp10_fb1.log)" >> $O
RR_TIGHT_FORCE=2 timeout 900 python tests/fuzz/fuzz_batch.py 200 91 > gpurun_out/r6_exp10_fb2.log 2>&1; echo "RR_TIGHT_FORCE=2 fuzz_batch 200 91: rc=$? $(tail -1 gpurun_out/r6_exp10_fb2.log)" >> $O
RR_STACKLESS=1 timeout 900 python tests/fuzz/fuzz_batch.py 150 77 > gpurun_out/r6_exp10_fb3.log 2>&1; echo "RR_STACKLESS=1 fuzz_batch 150 77: rc=$? $(tail -1 gpurun_out/r6_exp10_fb3.log)" >> $O
RR_HOST_SDMA=0 timeout 900 python tests/fuzz/fuzz_batch.py 150 78 > gpurun_out/r6_exp10_fb4.log 2>&1; echo "RR_HOST_SDMA=0 fuzz_batch 150 78: rc=$? $(tail -1 gpurun_out/r6_exp10_fb4.log)" >> $O
RR_HOST_SDMA=0 RR_FOLD_MIN_BUSY=0 timeout 900 python tests/fuzz/fuzz_batch.py 150 79 > gpurun_out/r6_exp10_fb5.log 2>&1; echo "RR_HOST_SDMA=0 RR_FOLD_MIN_BUSY=0 fuzz_batch 150 79: rc=$? $(tail -1 gpurun_out/r6_exp10_fb5.log)" >> $O
cat $O
