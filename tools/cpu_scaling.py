"""Thread-scaling curve of the CPU baseline (the oracle = line-faithful port of RadarCPU::simulate, OpenMP over azimuths
like RadarCPU.cpp:155; in-repo SAH BVH2 in Embree's place) on the host cores of this box.
usage: cpu_scaling.py [workload = target_10M_400x200_4pass] [frames per point = 6]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from radarays_ros_amd import params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
from oracle import oracle as O
from bench import WORKLOADS
wl = sys.argv[1] if len(sys.argv) > 1 else "target_10M_400x200_4pass"
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 6
scene_id, n_pass, n_rays = WORKLOADS[wl]
O.build()
s = scenes.config_scene(scene_id)
cfg = params.kaist_preset(n_reflections=n_pass, n_samples=n_rays, ambient_noise=2)
mats = [m.astuple() for m in materials_for(s)]
noise = (np.random.RandomState(7).uniform(0, 1, 400) * 1000.0).astype(np.float32)
poses = scenes.trajectory(16, s["name"])
t0 = time.time(); sc = O.Scene(s["verts"], s["faces"], s["face_object_id"]); print("BVH2 build %.1f s" % (time.time() - t0))
ncpu = os.cpu_count() or 1
print("%s on %d hardware threads" % (wl, ncpu))
base = None
for nt in [t for t in (1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256, 384, 512) if t <= ncpu]:
    n = max(2, nfr if nt >= 8 else 2)
    secs = []
    for k in range(n + 1):
        _, _, st = O.simulate(sc, mats, s["object_materials"], cfg, golden_beams(n_rays), poses[k % 16], noise_rnd=noise, want_f32=False,
                              n_threads=nt, brdf_model=1 if scene_id == 5 else 0)
        if k: secs.append(st["seconds"])
    med = float(np.median(secs)); base = base or med
    print("threads %4d: %8.3f images/s   speed-up %6.1fx   efficiency %5.1f %%" % (nt, 1.0 / med, base / med, 100.0 * base / med / nt))
