import sys, time, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
F = int(sys.argv[1]); K = int(sys.argv[2])
s = scenes.config_scene(2)
cfg = params.kaist_preset(n_reflections=1, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
nslot = 3
blocks = [torch.zeros((F, 400, cfg.n_cells), dtype=torch.uint8, device="cuda:0") for _ in range(nslot)]
imgs = [torch.zeros((F, cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0") for _ in range(nslot)]
streams = [torch.cuda.Stream() for _ in range(nslot)]
def step(k):
    i = k % nslot
    sp = streams[i].cuda_stream
    c.simulate_batch_columns_device([poses[(k * F + f) % 16] for f in range(F)], 0, 400, blocks[i].data_ptr(), sp)
    for f in range(F):
        c.assemble_image_device(blocks[i][f].data_ptr(), imgs[i][f].data_ptr(), sp)
for k in range(20): step(k)
torch.cuda.synchronize(); t0 = time.time()
for k in range(K): step(k)
te = time.time() - t0
torch.cuda.synchronize(); dt = time.time() - t0
c.set_timing_mode(2)
for k in range(200): step(k)
torch.cuda.synchronize()
ms, n = c.kernel_time("trace", True)
print("  k_trace avg %.1f us over %d launches (%.0f rays/launch)" % (1e3 * ms / n, n, F * 80000))
print("F=%d frames/s %.1f  us/frame %.1f (host %.1f us/frame)" % (F, K * F / dt, 1e6 * dt / (K * F), 1e6 * te / (K * F)))
c.close()
