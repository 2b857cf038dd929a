"""Parameter-batch throughput: K material sets per call vs K x (set_materials + simulate_device)."""
import sys, time, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 2
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 1
K = int(sys.argv[3]) if len(sys.argv) > 3 else 8
s = scenes.config_scene(cid)
cfg = params.kaist_preset(n_reflections=npass, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
mats = materials_for(s)
c.set_materials(mats, s["object_materials"], 0); c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
pose = scenes.default_pose(s["name"])
base = np.asarray([m.astuple() for m in mats], np.float32)
rs = np.random.RandomState(1)
sets = np.repeat(base[None], K, axis=0); sets[:, 1:, 2] = rs.uniform(0, 1, (K, len(mats) - 1))
imgs = torch.zeros((K, cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): c.simulate_material_sets_device(pose, sets, imgs.data_ptr(), st)
torch.cuda.synchronize(); N = 40; t0 = time.time()
for _ in range(N): c.simulate_material_sets_device(pose, sets, imgs.data_ptr(), st)
torch.cuda.synchronize(); dt = time.time() - t0
print("config %d passes %d: batch of %d sets: %.0f images/s" % (cid, npass, K, N * K / dt))
ms = [[params.RadarMaterial(*[float(x) for x in sets[k, i]]) for i in range(len(mats))] for k in range(K)]
t0 = time.time()
for _ in range(5):
    for k in range(K):
        c.set_materials(ms[k], s["object_materials"], 0); c.simulate_device(pose, imgs[k].data_ptr(), st)
torch.cuda.synchronize(); dt = time.time() - t0
print("one by one (set_materials + simulate_device): %.0f images/s" % (5 * K / dt))
c.close()
