"""Differential fuzz on the 100k-triangle scene of config 2 and the 1M-triangle scene of config 3 (deep
trees, spilled/long stacks, penetrable second object): random config / pose / azimuth window against the
oracle (its own SAH BVH2).  usage: fuzz_big.py [iterations] [seed] [config id]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from radarays_ros_amd import native, params, scenes
from oracle import oracle
from common import golden_beams, image_diff, materials_for, mats_tuple


def run(iters=20, seed=0, cid=2, verbose=True):
    rs = np.random.RandomState(seed)
    s = scenes.config_scene(cid)
    mats = materials_for(s)
    c = native.Context(0); c.set_mesh(s["verts"], s["faces"], s["face_object_id"]); c.set_materials(mats, s["object_materials"], 0)
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"])
    traj = scenes.trajectory(16, s["name"])
    bad = 0
    for it in range(iters):
        cfg = params.kaist_preset(
            n_reflections=int(rs.randint(1, 6)), ambient_noise=int(rs.choice([0, 2])), signal_denoising=int(rs.choice([0, 1, 1, 3])),
            record_multi_path=bool(rs.randint(0, 2)), record_multi_reflection=bool(rs.randint(0, 2)),
            scroll_image=int(rs.randint(0, 400)), multipath_threshold=float(rs.uniform(0, 0.9)))
        b = golden_beams(int(rs.randint(1, 48)))
        pose = traj[int(rs.randint(0, 16))].copy(); pose[4:6] += rs.uniform(-5, 5, 2).astype(np.float32)
        rnd = (rs.uniform(0, 1, 400) * 1000).astype(np.float32)
        a0 = int(rs.randint(0, 392)); a1 = a0 + int(rs.randint(1, 9))
        c.set_config(cfg); c.set_beam_samples(b); c.set_noise_offsets(rnd)
        g8, gf, gst = c.simulate(pose, a0, a1, want_f32=True)
        o8, of, ost = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, b, pose, noise_rnd=rnd, az_begin=a0, az_end=a1)
        d = image_diff(gf, of, g8, o8)
        ok = (gst["overflow"] == 0 and all(gst[k] == ost[k] for k in ("wave_passes", "hits", "signals")) and
              d["mean_dev"] <= 1e-5 and d["u8_max"] <= 1 and d["u8_mismatch_frac"] <= 1e-3)
        if not ok:
            bad += 1
            print("MISMATCH iteration", it, "az", a0, a1, {k: (gst[k], ost[k]) for k in ("wave_passes", "hits", "signals")}, d, cfg)
    c.close()
    if verbose:
        print("big-scene fuzz (config %d): %d iterations, %d mismatching" % (cid, iters, bad))
    return bad


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]] + [None] * 3
    sys.exit(1 if run(a[0] or 20, a[1] or 0, a[2] or 2) else 0)
