import sys, time, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 2
npass = int(sys.argv[2]) if len(sys.argv) > 2 else (1 if cid == 2 else 4)
noise = int(sys.argv[3]) if len(sys.argv) > 3 else 2
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 200
t0 = time.time(); s = scenes.config_scene(cid); print("scene", s["name"], len(s["faces"]), "tris", time.time() - t0)
cfg = params.kaist_preset(n_reflections=npass, ambient_noise=noise)
c = native.Context(0)
t0 = time.time(); c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); print("set_mesh", time.time() - t0, c.bvh_info())
c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg)
c.set_beam_samples(golden_beams(nb))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
img = torch.zeros((cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for p in poses[:3]: c.simulate_device(p, img.data_ptr(), st)
torch.cuda.synchronize()
c.set_stats_mode(True); c.simulate_device(poses[0], img.data_ptr(), st); print("stats", c.stats()); c.set_stats_mode(False)
K = int(sys.argv[5]) if len(sys.argv) > 5 else 64
t0 = time.time()
for k in range(K): c.simulate_device(poses[k % 16], img.data_ptr(), st)
t_enq = time.time() - t0
torch.cuda.synchronize(); dt = time.time() - t0
print("frames/s %.1f  ms/frame %.3f  (host enqueue %.1f us/frame)" % (K / dt, 1e3 * dt / K, 1e6 * t_enq / K))
c.set_timing_mode(1)
for k in range(K): c.simulate_device(poses[k % 16], img.data_ptr(), st)
torch.cuda.synchronize()
for n in ("trace", "shade", "scan", "column", "assemble"):
    ms, cnt = c.kernel_time(n, True); print(n, "avg us %.2f" % (1e3 * ms / max(cnt, 1)), "launches", cnt, "per frame us %.1f" % (1e3 * ms / K))
c.close()
del img
sys.stdout.flush()
