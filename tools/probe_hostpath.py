"""Where does the host-image path lose time?  Target workload, 8 poses per batch, 4 streams.
modes: dev = rr_simulate_batch_device (images stay in HBM); host = rr_simulate_batch_host_async.
The variants listed in rr_api.hip (copy stream, zero-copy assemble, own copy kernel, copy folded into a trace launch)
were measured with this script; only the shipped one is left in the library."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
wl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
P = 4 if wl != 2 else 1
s = scenes.config_scene(wl)
cfg = params.kaist_preset(n_reflections=P, n_samples=200, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 16 * 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
F, NS = 8, int(os.environ.get("NSTREAMS", "4"))
streams = [torch.cuda.Stream() for _ in range(NS)]
dimgs = [torch.zeros((F, cfg.n_cells, 400), dtype=torch.uint8, device="cuda") for _ in range(NS)]
hosts = [native.HostImages((F, cfg.n_cells, 400)) for _ in range(2 * NS)]
def run(mode, K=200):
    def one(k):
        ps = [poses[(k * F + f) % 16] for f in range(F)]
        if mode == "dev": c.simulate_batch_device(ps, dimgs[k % NS].data_ptr(), streams[k % NS].cuda_stream)
        else: c.simulate_batch_host_async(ps, hosts[k % len(hosts)].ptr, streams[k % NS].cuda_stream)
    for k in range(60): one(k)
    c.wait_host(None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): one(k)
    c.wait_host(None); torch.cuda.synchronize()
    return K * F / (time.perf_counter() - t0)
for m in sys.argv[2:] or ["dev", "host"]:
    print("%s (%d streams): %.1f img/s  %.1f img/s" % (m, NS, run(m), run(m)), flush=True)
