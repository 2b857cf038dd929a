"""Re-runs one seed of tests/test_gpu_parity.py::test_random_differential azimuth by azimuth and prints
where the GPU and the oracle part ways.  usage: fuzz_debug.py SEED"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from oracle import oracle
from common import golden_beams, mats_tuple
seed = int(sys.argv[1])
rs = np.random.RandomState(1000 + seed)
room = scenes.box12()
verts, faces, obj = [room["verts"]], [room["faces"]], [room["face_object_id"]]
n_obj = 1 + rs.randint(1, 4)
vb = len(room["verts"])
for o in range(1, n_obj):
    lo = rs.uniform([-8, -6, -0.9], [5, 4, 1.0]); hi = lo + rs.uniform(0.5, 3.0, 3)
    v, f = scenes._box_tris(lo, hi, vbase=vb)
    verts.append(v); faces.append(f); obj.append(np.full(12, o, np.uint32)); vb += 8
    nt = rs.randint(0, 6)
    if nt:
        tv = (rs.uniform(-7, 7, (nt, 1, 3)) * [1, 0.8, 0.1] + rs.normal(0, 0.7, (nt, 3, 3))).astype(np.float32)
        verts.append(tv.reshape(-1, 3)); faces.append((np.arange(3 * nt, dtype=np.uint32) + vb).reshape(nt, 3))
        obj.append(np.full(nt, o, np.uint32)); vb += 3 * nt
s = {"verts": np.concatenate(verts), "faces": np.concatenate(faces), "face_object_id": np.concatenate(obj)}
mats = [params.RadarMaterial(0.3, 1.0, 0.0, 1.0)]
for _ in range(3):
    mats.append(params.RadarMaterial(float(rs.choice([0.0, 0.05, 0.1, 0.2, 0.3, 0.45])), float(rs.uniform(0, 1)),
                                     float(rs.uniform(0, 1)), float(rs.choice([1.0, 5.0, 30.0, 3000.0]))))
s["object_materials"] = [int(rs.randint(1, 4)) for _ in range(n_obj)]
cfg = params.kaist_preset(
    n_reflections=int(rs.randint(1, 6)), ambient_noise=int(rs.choice([0, 0, 2, 1])),
    signal_denoising=int(rs.choice([0, 1, 1, 2, 3])), record_multi_path=bool(rs.randint(0, 2)),
    record_multi_reflection=bool(rs.randint(0, 2)), scroll_image=int(rs.randint(0, 400)),
    signal_denoising_triangular_width=int(rs.randint(1, 120)), energy_max=float(rs.uniform(0.2, 1.0)),
    signal_max=float(rs.uniform(50, 250)), resolution=float(rs.choice([0.0438, 0.0595238, 0.12])),
    n_cells=int(rs.choice([3424, 777, 2048])), multipath_threshold=float(rs.uniform(0, 0.9)))
if os.environ.get("FZ_OVERRIDE"):
    cfg = cfg.copy(**eval(os.environ["FZ_OVERRIDE"]))
beams_ = golden_beams(int(rs.randint(1, 70)))
pose = scenes.yaw_pose(float(rs.uniform(-2, 2)), float(rs.uniform(-2, 2)), float(rs.uniform(-0.5, 2.0)), float(rs.uniform(-3.1, 3.1)))
q = rs.normal(0, 1, 4); q /= np.linalg.norm(q)
if seed % 3 == 0: pose[:4] = q.astype(np.float32)
rnd = (rs.uniform(0, 1, 400) * 1000).astype(np.float32) if cfg.ambient_noise else None
a0 = int(rs.randint(0, 340))
print("cfg", cfg); print("mats", [m.astuple() for m in mats], "objmat", s["object_materials"], "beams", len(beams_), "az", a0)
c = native.Context(0); c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(mats, s["object_materials"], 0)
c.set_config(cfg, 400); c.set_beam_samples(beams_)
if rnd is not None: c.set_noise_offsets(rnd)
sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
for a in range(a0, a0 + 60):
    g8, gf, gst = c.simulate(pose, a, a + 1, want_f32=True)
    o8, of, ost = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, beams_, pose, noise_rnd=rnd, az_begin=a, az_end=a + 1)
    col = (cfg.scroll_image + a) % 400
    dg = np.abs(gf[:, col] - of[:, col])
    if gst["signals"] != ost["signals"] or gst["wave_passes"] != ost["wave_passes"] or dg.max() > 1e-3:
        bad = np.argwhere(dg > 1e-3).ravel()
        print("azimuth", a, "gpu", {k: gst[k] for k in ("wave_passes", "hits", "signals")}, "oracle", {k: ost[k] for k in ("wave_passes", "hits", "signals")},
              "max diff %.4g at bins" % dg.max(), bad[:12], "n_bad", len(bad), "gpu vals", gf[bad[:12], col], "oracle vals", of[bad[:12], col])
c.close()
if os.environ.get("FZ_DUMP"):
    a = int(os.environ["FZ_DUMP"])
    c = native.Context(0); c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(mats, s["object_materials"], 0)
    c.set_beam_samples(beams_)
    if rnd is not None: c.set_noise_offsets(rnd)
    col = (cfg.scroll_image + a) % 400
    for name, ov in (("full", {}), ("no noise", dict(ambient_noise=0)), ("no smear", dict(signal_denoising=0))):
        cf = cfg.copy(**ov); c.set_config(cf, 400)
        g8, gf, gst = c.simulate(pose, a, a + 1, want_f32=True)
        o8, of, ost = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cf, beams_, pose, noise_rnd=rnd, az_begin=a, az_end=a + 1)
        g, o = gf[:, col], of[:, col]
        i = int(np.argmax(np.abs(g - o)))
        print(name, "min/max gpu %.4f %.4f oracle %.4f %.4f  worst bin %d gpu %.5f oracle %.5f  nonfinite gpu %d oracle %d" % (
            np.nanmin(g), np.nanmax(g), np.nanmin(o), np.nanmax(o), i, g[i], o[i], (~np.isfinite(g)).sum(), (~np.isfinite(o)).sum()))
        print("   gpu   ", np.round(g[max(0, i - 4):i + 5], 4)); print("   oracle", np.round(o[max(0, i - 4):i + 5], 4))
    c.close()
