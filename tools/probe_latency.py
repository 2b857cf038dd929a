"""One frame at a time through rr_simulate (what a ROS node calling Radar::simulate() per frame sees): wall time per
frame and the GPU time of each kernel with the frame alone on the GPU.  usage: probe_latency.py [config id] [passes]"""
import sys, time, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = scenes.config_scene(cid)
cfg = params.kaist_preset(n_reflections=npass, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
for p in poses[:5]: c.simulate(p)
K = 100
ts = []
for k in range(K):
    t0 = time.perf_counter(); c.simulate(poses[k % 16]); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("rr_simulate: median %.3f ms  p10 %.3f  p90 %.3f per frame" % (np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
c.set_timing_mode(1)
for k in range(32): c.simulate(poses[k % 16])
tot = 0.0
for name in ("trace0", "trace", "shade", "scan", "column", "assemble"):
    ms, n = c.kernel_time(name, reset=True)
    if n: print("  %-8s %6.1f us per launch x %4.1f launches per frame = %7.1f us" % (name, 1e3 * ms / n, n / 32.0, 1e3 * ms / 32)); tot += 1e3 * ms / 32
print("  kernels together: %.1f us per frame" % tot)
c.close()
