"""Soak of the host delivery (rr_simulate_batch_host_async / rr_wait_host) at full size: every delivered image of every
batch is compared with the image the synchronous path gives for the same pose and noise row.
usage: soak_host.py [batches] [config id]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
K = int(sys.argv[1]) if len(sys.argv) > 1 else 400
wl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
P = 4 if wl != 2 else 1
s = scenes.config_scene(wl)
cfg = params.kaist_preset(n_reflections=P, n_samples=200, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
F, NS = 8, 4
noise = (np.random.RandomState(7).uniform(0, 1, (F, 400)) * 1000).astype(np.float32)
poses = scenes.trajectory(16, s["name"])
# reference: frame f of a batch uses noise row f
ref = {}
for f in range(F):
    c.set_noise_offsets(noise[f])
    for j in range(3):
        p = (j * 5 + f) % 16
        ref[(p, f)] = c.simulate(poses[p])[0].copy()
c.set_noise_offsets(noise)
streams = [torch.cuda.Stream() for _ in range(NS)]
hosts = [native.HostImages((F, cfg.n_cells, 400)) for _ in range(2 * NS)]
pending = []          # (batch index, host buffer index)
bad = 0; checked = 0
t0 = time.time()
for k in range(K):
    hb = k % len(hosts)
    # the buffer is reused every 2 * NS batches: check what it holds from last time first
    for (kk, h) in [x for x in pending if x[1] == hb]:
        c.wait_host(hosts[h].ptr)
        img = hosts[h].array
        for f in range(F):
            p = ((kk % 3) * 5 + f) % 16
            bad += not np.array_equal(img[f], ref[(p, f)]); checked += 1
        pending.remove((kk, h))
    # three pose sets in turn: a period coprime with the 4 lanes and the 8 host buffers, so that a lane (and a launch graph's
    # exec) meets other poses every time it comes round (advisor, round 5: with a period of 2 it never did)
    ps = [poses[((k % 3) * 5 + f) % 16] for f in range(F)]
    c.simulate_batch_host_async(ps, hosts[hb].ptr, streams[k % NS].cuda_stream)
    pending.append((k, hb))
c.wait_host(None)
for (kk, h) in pending:
    img = hosts[h].array
    for f in range(F):
        p = ((kk % 3) * 5 + f) % 16
        bad += not np.array_equal(img[f], ref[(p, f)]); checked += 1
print("host-delivery soak config %d: %d batches x %d frames in %.1f s, %d images checked, mismatching: %d" % (wl, K, F, time.time() - t0, checked, bad))
sys.exit(1 if bad else 0)
