"""Does a tiny GPU write to host memory after each batch cost throughput? (mechanism of the host-delivery loss)"""
import sys, os, time
R = "/root/repo"; sys.path.insert(0, R)
import numpy as np, torch
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
s = scenes.config_scene(4)
cfg = params.kaist_preset(n_reflections=4, n_samples=200, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 16 * 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
F, NS = 8, 4
streams = [torch.cuda.Stream() for _ in range(NS)]
dimgs = [torch.zeros((F, cfg.n_cells, 400), dtype=torch.uint8, device="cuda") for _ in range(NS)]
pin = [torch.zeros(F * cfg.n_cells * 400, dtype=torch.uint8).pin_memory() for _ in range(2 * NS)]
def run(mode, K=200):
    def one(k):
        st = streams[k % NS]
        c.simulate_batch_device([poses[(k * F + f) % 16] for f in range(F)], dimgs[k % NS].data_ptr(), st.cuda_stream)
        if mode == "none": return
        n = {"w64": 64, "w64k": 65536, "w1m": 1 << 20, "full": F * cfg.n_cells * 400}[mode]
        with torch.cuda.stream(st):
            pin[k % len(pin)][:n].copy_(dimgs[k % NS].view(-1)[:n], non_blocking=True)
    for k in range(60): one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): one(k)
    torch.cuda.synchronize()
    return K * F / (time.perf_counter() - t0)
for m in ("none", "w64", "w64k", "w1m", "full", "none"):
    print(m, round(run(m), 1), round(run(m), 1), flush=True)
