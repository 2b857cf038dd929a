"""Per-pass traversal statistics of a config scene (stats build of k_trace): node / leaf-triangle visits per
wave-pass for n_reflections = 1..P, differenced.  usage: probe_passes.py <config id> <passes> [builder]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
cid = int(sys.argv[1]); P = int(sys.argv[2]); builder = sys.argv[3] if len(sys.argv) > 3 else "host"
s = scenes.config_scene(cid)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"], builder=builder)
print("bvh", c.bvh_info(), flush=True)
c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_beam_samples(golden_beams(200))
pose = scenes.trajectory(16, s["name"])[3]
c.set_stats_mode(True)
prev = {"wave_passes": 0, "nodes_visited": 0, "tris_tested": 0}
for p in range(1, P + 1):
    c.set_config(params.kaist_preset(n_reflections=p, ambient_noise=0))
    _, _, st = c.simulate(pose)
    d = {k: st[k] - prev[k] for k in prev}
    wp = max(d["wave_passes"], 1)
    print("pass %d: wave_passes %8d  nodes/wp %6.2f  tris/wp %6.2f  bytes/wp %7.1f" %
          (p - 1, d["wave_passes"], d["nodes_visited"] / wp, d["tris_tested"] / wp,
           (d["nodes_visited"] * 128 + d["tris_tested"] * 48) / wp + 132), flush=True)
    prev = {k: st[k] for k in prev}
wp = prev["wave_passes"]
print("all   : wave_passes %8d  nodes/wp %6.2f  tris/wp %6.2f  bytes/wp %7.1f" %
      (wp, prev["nodes_visited"] / wp, prev["tris_tested"] / wp,
       (prev["nodes_visited"] * 128 + prev["tris_tested"] * 48) / wp + 132))
c.close()
