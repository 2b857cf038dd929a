# GPU box helper: the round's evidence in one call -- rocprofv3 kernel stats + PMC passes of every workload, then the bench lines
cd $GRAFT_REPO_ROOT
tag=${1:-r05}
profiles/collect.sh $tag > gpurun_out/collect_$tag.log 2>&1
profiles/collect.sh ${tag}_c3 --workload config3_1M_400x200_4pass > gpurun_out/collect_${tag}_c3.log 2>&1
profiles/collect.sh ${tag}_c2 --workload config2_100k_400x200_1pass > gpurun_out/collect_${tag}_c2.log 2>&1
profiles/collect.sh ${tag}_c4 --workload config4_10M_400x1000_4pass > gpurun_out/collect_${tag}_c4.log 2>&1
profiles/collect.sh ${tag}_c5 --workload config5_10M_400x1000_8pass_pertri --frames-per-rank 1 --steps 40 --warmup 4 > gpurun_out/collect_${tag}_c5.log 2>&1
[ -n "$SKIP_BENCH" ] || tools/final_bench.sh $tag      # (SKIP_BENCH=1: the bench lines are taken AFTER make_counters.py has refreshed roofline_counters.json)
ls gpurun_out/profiles_${tag}*/ | head -60
