#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_gpu_round6.py -q -x -s -k "brdf or fresnel" 2>&1 | tail -8
