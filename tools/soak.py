"""Soak test of the pipelined step loop: every image of every step is compared ON THE GPU with the
image the synchronous host path (rr_simulate) gives for the same pose.  usage: soak.py [steps] [config]"""
import sys, os, time, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.dist import AzimuthShard
from radarays_ros_amd.fixtures import golden_beams, materials_for
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cid = int(sys.argv[2]) if len(sys.argv) > 2 else 2
npass = 1 if cid == 2 else 4
s = scenes.config_scene(cid)
cfg = params.kaist_preset(n_reflections=npass, ambient_noise=2)
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(materials_for(s), s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(golden_beams(200))
c.set_noise_offsets((np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32))
poses = scenes.trajectory(16, s["name"])
ref = torch.stack([torch.from_numpy(c.simulate(p)[0]) for p in poses]).cuda()
dev = torch.device("cuda", 0)
shard = AzimuthShard(c, cfg.n_cells, 400, 0, 1, dev, frames_per_rank=4)
fps = shard.frames_per_step
bad = torch.zeros((), dtype=torch.int64, device=dev)
cur = torch.cuda.current_stream()
t0 = time.time()
for k in range(steps):
    # a stride of 5 poses per step: a lane (4 of them, handed out round robin) meets OTHER poses every time it comes round --
    # with a stride of fps = 4 every lane replayed its launch graph with the poses it already held (advisor, round 5)
    ids = [(k * 5 + f) % 16 for f in range(fps)]
    imgs = shard.step([poses[i] for i in ids], cur)
    shard.wait(cur)                              # the comparison below runs on `cur`, after the step
    bad += (imgs != ref[ids]).any(dim=(1, 2)).sum()
    if k % 8 == 7: torch.cuda.synchronize()      # keep at most 8 steps queued: a slot's images live for n_slots-1 steps
torch.cuda.synchronize()
print("soak config %d: %d steps x %d frames in %.1f s, mismatching images: %d" % (cid, steps, fps, time.time() - t0, int(bad.item())))
shard.close(); c.close()
sys.exit(1 if int(bad.item()) else 0)
