"""One frame through a library built with -DRR_COL_EXP=4 (k_column prints, for a few columns, how many (tile, 64-signal
batch) pairs its replay scans, how many hold an overlapping signal, and the replays).  usage: probe_colstats.py [config id] [passes]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
cid = int(sys.argv[1]) if len(sys.argv) > 1 else 4
npass = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = scenes.config_scene(cid)
cfg = params.kaist_preset(n_reflections=npass, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
c.simulate(scenes.trajectory(16, s["name"])[3])
c.close()
