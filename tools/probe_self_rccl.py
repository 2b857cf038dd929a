import os, sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["RR_MULTI_SELF_RCCL"] = "1"
from radarays_ros_amd import native, params, scenes
from common import golden_beams, materials_for
s = scenes.heightfield_room(64, n_buildings=40, seed=3)
cfg = params.kaist_preset(n_reflections=3, n_samples=60, ambient_noise=2)
noise = (np.random.RandomState(5).uniform(0, 1, (4, 400)) * 1000.0).astype(np.float32)
poses = scenes.trajectory(6, s["name"])
def setup(o, nz):
    o.set_mesh(s["verts"], s["faces"], s["face_object_id"]); o.set_materials(materials_for(s), s["object_materials"], 0)
    o.set_config(cfg, 400); o.set_beam_samples(golden_beams(60)); o.set_noise_offsets(nz)
m = native.MultiContext([0]); setup(m, noise)
print("created (one-rank RCCL communicator)", flush=True)
got = m.simulate_batch(poses)
print("batch through ncclSend/ncclRecv to self done", flush=True)
c = native.Context(0); setup(c, noise[0])
ok = True
for f, p in enumerate(poses):
    c.set_noise_offsets(noise[f % 4]); ok = ok and np.array_equal(got[f], c.simulate(p)[0])
print("sync equal:", ok, flush=True)
ring = [native.HostImages((3, cfg.n_cells, 400)) for _ in range(4)]
for k in range(6):
    h = ring[k % 4]; m.wait(h.ptr); m.simulate_batch_async(poses[k % 3:k % 3 + 3], h.ptr)
m.wait(None)
print("async ok", flush=True)
m.close(); c.close()
