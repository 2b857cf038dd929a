"""Scene + ray set for tools/treeq.cpp: dumps a config scene as raw arrays and logs the rays the CPU oracle casts
(ORC_RAYLOG) for a sample of azimuths.  usage: treeq_dump.py <config id> <passes> <out dir> [n azimuths] [rays/beam]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
from radarays_ros_amd import params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
cid, P, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n_az = int(sys.argv[4]) if len(sys.argv) > 4 else 10
n_rays = int(sys.argv[5]) if len(sys.argv) > 5 else 200
os.makedirs(out, exist_ok=True)
s = scenes.config_scene(cid)
s["verts"].astype(np.float32).tofile(os.path.join(out, "verts.f32"))
s["faces"].astype(np.uint32).tofile(os.path.join(out, "faces.u32"))
log = os.path.join(out, "rays.bin")
for f in (log, os.path.join(out, "hits.ref")):
    if os.path.exists(f):
        os.remove(f)
os.environ["ORC_RAYLOG"] = log
from oracle import oracle as O
O.build()
sc = O.Scene(s["verts"], s["faces"], s["face_object_id"])
cfg = params.kaist_preset(n_reflections=P, n_samples=n_rays, ambient_noise=0)
m = [x.astuple() for x in materials_for(s)]
pose = scenes.trajectory(16, s["name"])[3]
for a in np.linspace(0, 400, n_az, endpoint=False).astype(int):
    O.simulate(sc, m, s["object_materials"], cfg, golden_beams(n_rays), pose, az_begin=int(a), az_end=int(a) + 1, n_threads=1)
print("rays:", os.path.getsize(log) // 40)
