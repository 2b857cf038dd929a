"""Frames through the GPU-built tree under builder options (env RR_LBVH_*), one context per option set.
usage: probe_lbvh.py <config id> <passes>"""
import sys, time, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
import torch
cid = int(sys.argv[1]); npass = int(sys.argv[2])
s = scenes.config_scene(cid)
cfg = params.kaist_preset(n_reflections=npass, ambient_noise=0)
poses = scenes.trajectory(16, s["name"])
img = torch.zeros((cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
ref = None
for env in ({}, {"RR_LBVH_NO_SPLIT": "1"}):
    for k in ("RR_LBVH_NO_SPLIT",):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = native.Context(0)
    c.set_materials(materials_for(s), s["object_materials"], 0); c.set_config(cfg); c.set_beam_samples(golden_beams(200))
    t0 = time.time(); c.set_mesh(s["verts"], s["faces"], s["face_object_id"], builder="gpu"); tb = time.time() - t0
    for p in poses[:3]: c.simulate_device(p, img.data_ptr(), None)
    c.synchronize(); t0 = time.time()
    for k in range(48): c.simulate_device(poses[k % 16], img.data_ptr(), None)
    c.synchronize(); dt = (time.time() - t0) / 48
    out = img.cpu().numpy().copy()
    if ref is None: ref = out
    print("%-45s build %.3f s  %s  frame %.3f ms  same image %s" % (env, tb, c.bvh_info(), 1e3 * dt, np.array_equal(out, ref)), flush=True)
    c.close()
