"""Fuzz of the parameter batch (rr_simulate_param_sets): random scenes, configs (every switch of fuzz_diff2: denoisers,
noise modes, multipath, odd image shapes), 1..10 sets that differ in material table, beam table (1..4 distinct tables,
some shared = pass-0 groups, some equal to the context's own) and passes (0..6, below and above the config's) --
image k must be bit-identical to the same parameters set one by one on a second context, one set per case goes to the
oracle as well, and the scores must equal numpy's PSNR.  usage: fuzz_sets.py [n] [seed]"""
import sys, os, numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from radarays_ros_amd import native, params, scenes
from oracle import oracle
from common import golden_beams, mats_tuple
from fuzz_diff2 import make_scene


def psnr_np(ref, img):
    err = np.mean((ref.astype(np.float64) - img.astype(np.float64)) ** 2)
    return np.inf if err == 0 else 10.0 * np.log10(255.0 ** 2 / err)


def run(n=20, seed=0, verbose=True):
    bad = 0
    for it in range(n):
        case = 300000 + 1000 * seed + it
        rs = np.random.RandomState(case)
        s = make_scene(rs)
        def rand_mats():
            return [params.RadarMaterial(0.3, 1.0, 0.0, 1.0)] + [
                params.RadarMaterial(float(rs.choice([0.0, 0.03, 0.1, 0.25, 0.3])), float(rs.uniform(0, 1)), float(rs.uniform(0, 1)),
                                     float(rs.choice([1.0, 2.0, 30.0, 3000.0]))) for _ in range(3)]
        mats = rand_mats()
        n_angles = int(rs.choice([400, 100, 36]))
        cfg = params.kaist_preset(
            n_reflections=int(rs.randint(0, 5)), ambient_noise=int(rs.choice([0, 1, 2])), signal_denoising=int(rs.choice([0, 1, 2, 3])),
            signal_denoising_triangular_width=int(rs.randint(1, 100)), signal_denoising_triangular_mode=float(rs.uniform(0.05, 0.9)),
            signal_denoising_gaussian_width=int(rs.randint(1, 100)), signal_denoising_gaussian_mode=float(rs.uniform(0.05, 0.9)),
            signal_denoising_mb_width=int(rs.randint(2, 100)), signal_denoising_mb_mode=float(rs.uniform(0.05, 0.9)),
            record_multi_path=bool(rs.randint(0, 2)), record_multi_reflection=bool(rs.randint(0, 2)),
            scroll_image=int(rs.randint(0, n_angles)), resolution=float(rs.choice([0.0438, 0.0595238, 0.3])),
            n_cells=int(rs.choice([3424, 100, 1111])), multipath_threshold=float(rs.uniform(-0.5, 0.95)))
        nb = int(rs.randint(1, 40))
        base = golden_beams(nb)
        tables = [None] + [native.sample_cone_local(int(rs.randint(0, 1 << 30)), float(np.radians(rs.uniform(1, 20))), nb, int(rs.randint(0, 4)), 0.8)
                           for _ in range(int(rs.randint(0, 4)))]
        rnd = (rs.uniform(0, 1, n_angles) * 1000).astype(np.float32)
        pose = scenes.yaw_pose(float(rs.uniform(-2, 2)), float(rs.uniform(-2, 2)), float(rs.uniform(-0.5, 2.0)), float(rs.uniform(-3.1, 3.1)))
        K = int(rs.randint(1, 11))
        sets = []
        for k in range(K):
            t = tables[int(rs.randint(0, len(tables)))]
            if t is not None and rs.randint(0, 2):
                t = t.copy()                                     # same bytes behind another pointer: still one group
            if t is None and rs.randint(0, 3) == 0:
                t = base.copy()                                  # the context's own samples given explicitly
            sets.append({"materials": None if rs.randint(0, 4) == 0 else np.array(mats_tuple(rand_mats()), np.float32),
                         "beam_dirs": t, "n_reflections": None if rs.randint(0, 3) == 0 else int(rs.randint(0, 7))})
        builder = str(rs.choice(["host", "gpu"]))
        def ctx():
            c = native.Context(0)
            c.set_mesh(s["verts"], s["faces"], s["face_object_id"], builder=builder)
            c.set_materials(mats, s["object_materials"], 0); c.set_config(cfg, n_angles)
            c.set_beam_samples(base); c.set_noise_offsets(rnd)
            return c
        c, one = ctx(), ctx()
        try:
            imgs, _ = c.simulate_param_sets(pose, sets, len(mats))
        except native.RRError as e:
            print("GPU error at case", case, e); bad += 1; c.close(); one.close(); continue
        refs = []
        ok = True
        for k, st in enumerate(sets):
            m = mats if st["materials"] is None else [tuple(float(x) for x in r) for r in st["materials"]]
            b = base if st["beam_dirs"] is None else st["beam_dirs"]
            npass = cfg.n_reflections if st["n_reflections"] is None else st["n_reflections"]
            one.set_materials(m, s["object_materials"], 0); one.set_beam_samples(b); one.set_config(cfg.copy(n_reflections=npass), n_angles)
            u8, _, _ = one.simulate(pose)
            refs.append(u8)
            if not np.array_equal(imgs[k], u8):
                ok = False
                print("MISMATCH case", case, "set", k, "of", K, "passes", npass, "differing pixels", int((imgs[k] != u8).sum()))
        _, psnr = c.simulate_param_sets(pose, sets, len(mats), ref_u8=refs[0], want_images=False)
        for k in range(K):
            w = psnr_np(refs[0], refs[k])
            if not ((np.isinf(w) and np.isinf(psnr[k])) or abs(psnr[k] - w) <= 1e-9):
                ok = False; print("PSNR mismatch case", case, "set", k, psnr[k], w)
        # one set against the oracle
        k = int(rs.randint(0, K)); st = sets[k]
        m = mats_tuple(mats) if st["materials"] is None else [tuple(float(x) for x in r) for r in st["materials"]]
        npass = cfg.n_reflections if st["n_reflections"] is None else st["n_reflections"]
        sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
        o8, of, _ = oracle.simulate(sc, m, s["object_materials"], cfg.copy(n_reflections=npass), base if st["beam_dirs"] is None else st["beam_dirs"],
                                    pose, noise_rnd=rnd, n_angles=n_angles)
        if np.isfinite(of).all():
            d8 = np.abs(imgs[k].astype(np.int32) - o8.astype(np.int32))
            if d8.max() > 1 or (d8 > 0).mean() > 2e-3:
                ok = False; print("ORACLE mismatch case", case, "set", k, int(d8.max()), float((d8 > 0).mean()))
        bad += 0 if ok else 1
        c.close(); one.close()
    if verbose:
        print("param-set fuzz: %d cases, %d mismatching" % (n, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
