#!/bin/bash
# GPU box helper: isolated kernel durations for several RR_PASS0_AZ values
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for a in ${1:-16 4 1}; do for w in config2_100k_400x200_1pass config3_1M_400x200_4pass target_10M_400x200_4pass; do
  echo "PASS0_AZ=$a $w: $(RR_PASS0_AZ=$a FPR=8 tools/ktrace1.sh $w 2>&1 | grep -E 'k_trace<true, false|value' | sed -e 's/void rr::k_trace<true, false, false>(rr::Params, int)/trace0/' | tr -s ' ' | tr '\n' ' ')"
done; done
