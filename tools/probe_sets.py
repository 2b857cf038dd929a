"""Parameter-batch throughput: K material sets per call vs K x (set_materials + simulate_device); and objective
evaluations/s of the optimiser's full parameter vector (beam_width, n_reflections, materials: rr_simulate_param_sets,
scores only) against the reference's shape (one set_* + simulate + D2H + numpy PSNR per evaluation).
usage: probe_sets.py [config id] [passes] [sets per call]"""
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
# the optimiser's evaluation (scripts/radaray_opti.py:170-211): K parameter vectors -> K scores, mixed n_reflections and beam widths
import math
widths = [6.0, 10.0, 14.0]
tables = [native.sample_cone_local(40 + i, math.radians(w), 200, 2, 0.8) for i, w in enumerate(widths)]
psets = [{"materials": sets[k], "beam_dirs": tables[k % 3], "n_reflections": 1 + (k % max(npass, 1))} for k in range(K)]
real, _, _ = c.simulate(pose)
for _ in range(3): c.simulate_param_sets(pose, psets, len(mats), ref_u8=real, want_images=False)
N = 30; t0 = time.time()
for _ in range(N): _, psnr = c.simulate_param_sets(pose, psets, len(mats), ref_u8=real, want_images=False)
dt = time.time() - t0
print("param sets (3 beam widths, 1..%d passes), scores only: %.0f evaluations/s (K = %d per call)" % (max(npass, 1), N * K / dt, K))
t0 = time.time()
for _ in range(3):
    for k in range(K):
        c.set_materials(ms[k], s["object_materials"], 0); c.set_beam_samples(psets[k]["beam_dirs"])
        c.set_config(cfg.copy(n_reflections=psets[k]["n_reflections"]))
        u8, _, _ = c.simulate(pose)
        err = np.mean((real.astype(np.float64) - u8.astype(np.float64)) ** 2); p1 = 10 * np.log10(255.0 ** 2 / err) if err else np.inf
dt = time.time() - t0
print("one by one (set_* + rr_simulate + host PSNR): %.0f evaluations/s" % (3 * K / dt))
c.close()
