import sys, time, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
s = scenes.config_scene(2)
cfg = params.kaist_preset(n_reflections=1, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
for p in poses[:5]: c.simulate(p)
t0 = time.time(); K = 300
for k in range(K): c.simulate(poses[k % 16])
dt = time.time() - t0
print("rr_simulate (host buffers, synchronous, D2H + host transpose): %.1f images/s, %.3f ms/frame" % (K / dt, 1e3 * dt / K))
c.close()
# the reference node's loop (radar_simulator.cpp:83-96): loadParams() + fresh noise offsets before EVERY frame
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); mats = materials_for(s)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
rs = np.random.RandomState(1)
for k in range(K + 5):
    if k == 5: t0 = time.time()
    c.set_materials(mats, s["object_materials"], 0); c.set_config(cfg)
    c.set_noise_offsets((rs.uniform(0, 1, 400) * 1000).astype(np.float32))
    c.simulate(poses[k % 16])
dt = time.time() - t0
print("node loop (loadParams + config + new noise offsets + rr_simulate per frame): %.1f images/s, %.3f ms/frame" % (K / dt, 1e3 * dt / K))
c.close()
