#!/bin/bash
# GPU box helper: builds tools/cpp_bench.cpp and runs it on a config scene
# usage: cpp_bench.sh [frames] [batch] [mode: sync|multi|graph] [config id = 2] [passes = 1 for config 2, else 4]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
CFG=${4:-2}; PASSES=${5:-$([ "$CFG" = 2 ] && echo 1 || echo 4)}
python - <<PY
import sys; sys.path.insert(0, "tests")
from test_cpp_host import write_scene
from radarays_ros_amd import params, scenes
from common import golden_beams, materials_for
s = scenes.config_scene($CFG)
write_scene("/tmp/c$CFG.bin", s, materials_for(s), golden_beams(200), scenes.default_pose(s["name"]), params.kaist_preset(n_reflections=$PASSES, ambient_noise=2))
PY
g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -Wno-unused-result -I include -I /opt/rocm/include tools/cpp_bench.cpp -o /tmp/cpp_bench \
    -L radarays_ros_amd -lradarays_mi355 -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$R/radarays_ros_amd && /tmp/cpp_bench /tmp/c$CFG.bin ${1:-4000} ${2:-4} $3
