"""Soak of rr_multi's n-device path on ONE GPU (RR_MULTI_LOOPBACK: n contexts, n streams per slot, the plan's copies) with
the per-device enqueue threads on (RR_MULTI_THREADS=1): batches of random sizes issued asynchronously over a ring of host
buffers, every delivered image compared with rr_simulate's for the same pose and noise row; every 97th batch overflows on
purpose (the error must be reported once, the pipeline drained, the next batch healthy).
usage: soak_multi.py [batches] [device entries] [config id]      (RR_MULTI_THREADS=0 for the single-thread path)"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
os.environ["RR_MULTI_LOOPBACK"] = "1"
os.environ.setdefault("RR_MULTI_THREADS", "1")
import numpy as np
from radarays_ros_amd import native, params, scenes
from radarays_ros_amd.fixtures import golden_beams, materials_for
K = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ND = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = int(sys.argv[3]) if len(sys.argv) > 3 else 3
P = 4 if wl != 2 else 1
s = scenes.config_scene(wl)
cfg = params.kaist_preset(n_reflections=P, n_samples=200, ambient_noise=2)
mats = materials_for(s); beams = golden_beams(200)
F = 8
noise = (np.random.RandomState(7).uniform(0, 1, (F, 400)) * 1000).astype(np.float32)
poses = scenes.trajectory(16, s["name"])
c = native.Context(0)
c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(mats, s["object_materials"], 0)
c.set_config(cfg); c.set_beam_samples(beams)
ref = {}
for f in range(F):
    c.set_noise_offsets(noise[f])
    for p in range(16):
        ref[(p, f)] = c.simulate(poses[p])[0].copy()
c.close()
m = native.MultiContext([0] * ND)
m.set_mesh(s["verts"], s["faces"], s["face_object_id"]); m.set_materials(mats, s["object_materials"], 0)
m.set_config(cfg, 400); m.set_beam_samples(beams); m.set_noise_offsets(noise)
ring = [native.HostImages((F, cfg.n_cells, 400)) for _ in range(6)]
rs = np.random.RandomState(123)
pending = {}          # ring index -> list of pose indices
bad = checked = errors = 0
t0 = time.time()


def collect(h):
    global bad, checked
    ps = pending.pop(h)
    m.wait(ring[h].ptr)
    for f, p in enumerate(ps):
        bad += not np.array_equal(ring[h].array[f], ref[(p, f)]); checked += 1


for k in range(K):
    h = k % len(ring)
    if h in pending:
        collect(h)
    if k % 97 == 96:
        # an overflowing batch: reported by the wait, everything drained, then back to the healthy config
        for hh in sorted(pending):
            collect(hh)
        m.set_config(cfg, 400, max_waves_per_azimuth=201)
        m.simulate_batch_async([poses[0], poses[1]], ring[h].ptr)
        try:
            m.wait(None)
            if P > 1:
                bad += 1      # the error went missing
            # (one ray-cast pass: no wave ever has a child, the queue cannot overflow -- the reconfiguration alone is exercised)
        except native.RRError:
            errors += 1
        m.set_config(cfg, 400)
        continue
    n = int(rs.randint(1, F + 1))
    ps = [int(x) for x in rs.randint(0, 16, n)]
    ring[h].array[:] = 0x5A
    m.simulate_batch_async([poses[p] for p in ps], ring[h].ptr)
    pending[h] = ps
for hh in sorted(pending):
    collect(hh)
m.wait(None)
print("rr_multi soak (%d device entries in loopback, enqueue threads %s, config %d): %d batches in %.1f s, %d images checked, "
      "mismatching: %d, provoked overflows reported: %d" % (ND, os.environ["RR_MULTI_THREADS"], wl, K, time.time() - t0, checked, bad, errors))
for x in ring:
    x.close()
m.close()
sys.exit(1 if bad else 0)
