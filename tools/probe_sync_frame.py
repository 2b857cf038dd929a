"""One synchronous rr_simulate per frame (the reference's call shape, radar_simulator.cpp:197-212) on the target workload: wall time
per call, and -- under `rocprofv3 --kernel-trace --stats` -- the kernels' own time per call, to see what the rest is.
usage: probe_sync_frame.py [frames] [workload id]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
wl = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # scene of the target = config 4's mesh, 200 rays
s = scenes.config_scene(wl)
cfg = params.kaist_preset(n_reflections=4 if wl != 2 else 1, n_samples=200, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
for k in range(20): c.simulate(poses[k % 16])
ts = []
for k in range(N):
    t0 = time.perf_counter(); c.simulate(poses[k % 16]); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("sync frames: %d, ms per call median %.4f p10 %.4f p90 %.4f; graph stats %s" % (N, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), c.graph_stats() if hasattr(c, "graph_stats") else ""))
