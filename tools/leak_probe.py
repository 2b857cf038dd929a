"""Where does a process' resident memory go over many context life cycles?  Stages: 0 = create + set-up + destroy,
1 = + one synchronous frame, 2 = + a host-delivered batch.  usage: leak_probe.py STAGE [cycles]"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
stage = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
s = scenes.config_scene(2)
cfg = params.kaist_preset(n_reflections=2, n_samples=64, ambient_noise=2)
poses = scenes.trajectory(16, s["name"])
host = native.HostImages((4, cfg.n_cells, 400))
st = torch.cuda.Stream()
def rss():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS:"): return int(line.split()[1]) / 1024
out = []
for k in range(N):
    c = native.Context(0)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
    c.set_config(cfg); c.set_beam_samples(golden_beams(64))
    if stage >= 1: c.simulate(poses[k % 16])
    if stage >= 2:
        c.simulate_batch_host_async([poses[(k + f) % 16] for f in range(4)], host.ptr, st.cuda_stream); c.wait_host(None)
    c.close()
    if k % 50 == 49: out.append("%d:%.0f" % (k + 1, rss()))
free, total = torch.cuda.mem_get_info()
print("stage %d RSS MB at cycle " % stage + " ".join(out) + "  device memory in use %.0f MB" % ((total - free) / 2**20))
