cd $GRAFT_REPO_ROOT
for cfg in "8 4 2" "12 4 2" "16 4 1" "8 6 2" "8 3 2" "6 4 2" "10 4 2"; do set -- $cfg; fpr=$1; sl=$2; bps=$3
  v=$(RR_LANES=$sl python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 5 --frames-per-rank $fpr --slots $sl --batches-per-step $bps 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print(d['value'], d['ms_per_step'])")
  echo "fpr=$fpr slots=$sl bps=$bps -> $v"
done
