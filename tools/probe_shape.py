import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
for cid in (4, 3):
    s = scenes.config_scene(cid)
    c = native.Context(0)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    c.set_materials(materials_for(s), s["object_materials"], 0)
    c.set_beam_samples(golden_beams(200))
    c.set_config(params.kaist_preset(n_reflections=4, ambient_noise=0))
    c.set_stats_mode(True)
    c.simulate(scenes.trajectory(16, s["name"])[3])
    sh = c.traversal_shape()
    w = sh["waves"]
    print("config", cid, {k: round(v / w, 2) for k, v in sh.items() if k != "waves"}, "waves", w)
    c.close()
