cd $GRAFT_REPO_ROOT
profiles/collect.sh r05 > gpurun_out/collect_r05.log 2>&1
profiles/collect.sh r05_c3 --workload config3_1M_400x200_4pass > gpurun_out/collect_r05_c3.log 2>&1
profiles/collect.sh r05_c2 --workload config2_100k_400x200_1pass > gpurun_out/collect_r05_c2.log 2>&1
tools/bench_all.sh r05 2>&1 | tail -12
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench_target_20.json 2>/dev/null
python bench.py --no-cpu-baseline --self-launch > gpurun_out/r05_bench_target_selflaunch.json 2>/dev/null
python bench.py --no-cpu-baseline --force-slots > gpurun_out/r05_bench_target_forceslots.json 2>/dev/null
python bench.py --no-cpu-baseline --force-slots --strong > gpurun_out/r05_bench_target_forceslots_strong.json 2>/dev/null
for f in target_20 target_selflaunch target_forceslots target_forceslots_strong; do echo "$f $(grep -o '"value": [0-9.]*' gpurun_out/r05_bench_$f.json | head -1)"; done
ls gpurun_out/profiles_r05*/ | head -40
