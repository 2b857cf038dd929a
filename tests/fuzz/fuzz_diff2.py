"""Second differential fuzzer against the oracle: what test_random_differential does not vary -- per-azimuth
pose tables (include_motion), azimuth counts other than 400, every denoiser with widths up to 256, all three
noise modes, tiny and huge resolutions, thresholds, range_max, GPU-built trees.  usage: fuzz_diff2.py [n] [seed]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from oracle import oracle
from common import golden_beams, image_diff, mats_tuple


def make_scene(rs):
    room = scenes.box12()
    verts, faces, obj = [room["verts"]], [room["faces"]], [room["face_object_id"]]
    n_obj = 1 + rs.randint(1, 5); vb = len(room["verts"])
    for o in range(1, n_obj):
        lo = rs.uniform([-8, -6, -0.9], [5, 4, 1.0]); hi = lo + rs.uniform(0.3, 4.0, 3)
        v, f = scenes._box_tris(lo, hi, vbase=vb)
        verts.append(v); faces.append(f); obj.append(np.full(12, o, np.uint32)); vb += 8
    s = {"verts": np.concatenate(verts), "faces": np.concatenate(faces), "face_object_id": np.concatenate(obj)}
    s["object_materials"] = [int(rs.randint(1, 4)) for _ in range(n_obj)]
    return s


def run(n=50, seed=0, verbose=True):
    bad = 0
    for it in range(n):
        rs = np.random.RandomState(100000 + 1000 * seed + it)
        s = make_scene(rs)
        mats = [params.RadarMaterial(0.3, 1.0, 0.0, 1.0)] + [
            params.RadarMaterial(float(rs.choice([0.0, 0.03, 0.1, 0.25, 0.3])), float(rs.uniform(0, 1)), float(rs.uniform(0, 1)),
                                 float(rs.choice([1.0, 2.0, 30.0, 3000.0]))) for _ in range(3)]
        n_angles = int(rs.choice([400, 100, 36, 7]))
        kind = int(rs.choice([0, 1, 2, 3]))
        cfg = params.kaist_preset(
            n_reflections=int(rs.randint(1, 7)), ambient_noise=int(rs.choice([0, 1, 2])), signal_denoising=kind,
            signal_denoising_triangular_width=int(rs.randint(1, 257)), signal_denoising_triangular_mode=float(rs.uniform(0.05, 0.9)),
            signal_denoising_gaussian_width=int(rs.randint(1, 257)), signal_denoising_gaussian_mode=float(rs.uniform(0.05, 0.9)),
            signal_denoising_mb_width=int(rs.randint(2, 257)), signal_denoising_mb_mode=float(rs.uniform(0.05, 0.9)),
            record_multi_path=bool(rs.randint(0, 2)), record_multi_reflection=bool(rs.randint(0, 2)),
            scroll_image=int(rs.randint(0, n_angles)), energy_max=float(rs.uniform(0.1, 1.5)), signal_max=float(rs.uniform(20, 255)),
            resolution=float(rs.choice([0.01, 0.0438, 0.0595238, 0.3])), n_cells=int(rs.choice([3424, 100, 1111, 4096])),
            multipath_threshold=float(rs.uniform(-0.5, 0.95)),
            ambient_noise_at_signal_0=float(rs.uniform(0, 0.5)), ambient_noise_at_signal_1=float(rs.uniform(0, 0.2)),
            ambient_noise_energy_max=float(rs.uniform(0, 0.5)), ambient_noise_energy_min=float(rs.uniform(0, 0.2)),
            ambient_noise_energy_loss=float(rs.uniform(0, 0.3)))
        b = golden_beams(int(rs.randint(1, 50)))
        rnd = (rs.uniform(0, 1, n_angles) * 1000).astype(np.float32)
        motion = bool(rs.randint(0, 2))
        base = scenes.yaw_pose(float(rs.uniform(-2, 2)), float(rs.uniform(-2, 2)), float(rs.uniform(-0.5, 2.0)), float(rs.uniform(-3.1, 3.1)))
        if motion:
            pose = np.tile(base, (n_angles, 1)).astype(np.float32)
            pose[:, 4] += np.linspace(0, float(rs.uniform(0, 1.0)), n_angles, dtype=np.float32)
            yaw = float(rs.uniform(-0.2, 0.2))
            for a in range(n_angles):
                pose[a] = scenes.yaw_pose(float(pose[a, 4]), float(base[5]), float(base[6]), float(np.arctan2(base[2], base[3]) * 2 + yaw * a / n_angles))
        else:
            pose = base
        a0 = int(rs.randint(0, max(1, n_angles - 5))); a1 = min(n_angles, a0 + int(rs.randint(1, 40)))
        c = native.Context(0)
        c.set_mesh(s["verts"], s["faces"], s["face_object_id"], builder=str(rs.choice(["host", "gpu"])))
        c.set_materials(mats, s["object_materials"], 0)
        c.set_config(cfg, n_angles)
        c.set_beam_samples(b); c.set_noise_offsets(rnd)
        if motion:
            c.set_motion_poses(pose)
        try:
            g8, gf, gst = c.simulate(pose[0] if motion else pose, a0, a1, want_f32=True)
        except native.RRError as e:
            print("GPU error at", it, e); bad += 1; c.close(); continue
        c.close()
        sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
        o8, of, ost = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, b, pose, noise_rnd=rnd, az_begin=a0, az_end=a1,
                                      n_angles=n_angles)
        cols = [(cfg.scroll_image + a) % n_angles for a in range(a0, a1)]
        d = image_diff(gf[:, cols], of[:, cols], g8[:, cols], o8[:, cols])
        finite = np.isfinite(of[:, cols]).all()
        ok = all(gst[k] == ost[k] for k in ("wave_passes", "hits", "signals")) and (
            not finite or (d["mean_dev"] <= 1e-5 and d["u8_max"] <= 1 and d["u8_mismatch_frac"] <= 2e-3))
        if not ok:
            bad += 1
            print("MISMATCH case", 100000 + 1000 * seed + it, "motion", motion, "n_angles", n_angles, "az", a0, a1,
                  {k: (gst[k], ost[k]) for k in ("wave_passes", "hits", "signals")}, d, "finite", finite, cfg)
    if verbose:
        print("diff2 fuzz: %d cases, %d mismatching" % (n, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
