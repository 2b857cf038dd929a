"""Host BVH build time of a config scene (no frames): probe_hostbuild.py <config id> [repeats]"""
import sys, time, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from radarays_ros_amd import native, scenes
cid = int(sys.argv[1]); rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
s = scenes.config_scene(cid)
c = native.Context(0)
for k in range(rep):
    t0 = time.time(); c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); print("build %.3f s" % (time.time() - t0), c.bvh_info(), flush=True)
c.close()
