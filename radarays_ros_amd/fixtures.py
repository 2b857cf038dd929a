"""Inputs every harness of this package shares (bench.py, tools/, tests/): the stored beam-sample
directions and the material table that goes with a synthetic scene.

The beam samples are a FIXTURE because the reference draws them with std::mt19937 +
libstdc++ distributions (radar_algorithms.cpp:263-289), which are not portable across standard
libraries: `data/beam_dirs_seed42_n1000.npy` holds 1000 directions generated once
(seed 42, KAIST preset: beam width 10 deg, sample distribution 2, p_in_cone 0.8)."""
import os

import numpy as np

from . import params

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


def golden_beams(n):
    d = np.load(os.path.join(DATA, "beam_dirs_seed42_n1000.npy"))
    assert d.shape == (1000, 3) and d.dtype == np.float32
    if n > len(d):
        raise ValueError("the fixture holds 1000 beam samples")
    return np.ascontiguousarray(d[:n])


def materials_for(scene):
    """KAIST materials (config/mulran_kaist02.yaml:8-20) + the penetrable one of the Snell/Fresnel
    configs when the scene has a second object; config 5's 8-material table for per-triangle scenes."""
    if "_pertri" in scene.get("name", ""):
        return params.config5_materials()
    m = params.kaist_materials()
    if max(scene["object_materials"]) >= 2:
        m = m + [params.PENETRABLE]
    return m


def random_room_case(seed):
    """A randomised small scene / material table / config (box12 room + random boxes and free triangles, every
    config switch, arbitrary attitude for every third seed): the differential fuzz case of the parity tests
    (tests/test_gpu_parity.py) and tests/fuzz/.  -> (scene, cfg, materials, beams, pose, noise offsets | None,
    (az_begin, az_end))"""
    from . import scenes
    rs = np.random.RandomState(1000 + seed)
    room = scenes.box12()
    verts, faces, obj = [room["verts"]], [room["faces"]], [room["face_object_id"]]
    n_obj = 1 + rs.randint(1, 4)
    vb = len(room["verts"])
    for o in range(1, n_obj):                      # random boxes and free triangles inside the room
        lo = rs.uniform([-8, -6, -0.9], [5, 4, 1.0])
        hi = lo + rs.uniform(0.5, 3.0, 3)
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
    beams_ = golden_beams(int(rs.randint(1, 70)))
    pose = scenes.yaw_pose(float(rs.uniform(-2, 2)), float(rs.uniform(-2, 2)), float(rs.uniform(-0.5, 2.0)),
                           float(rs.uniform(-3.1, 3.1)))
    q = rs.normal(0, 1, 4); q /= np.linalg.norm(q)
    if seed % 3 == 0:
        pose[:4] = q.astype(np.float32)            # arbitrary 3-D attitude, not only yaw
    rnd = (rs.uniform(0, 1, 400) * 1000).astype(np.float32) if cfg.ambient_noise else None
    a0 = int(rs.randint(0, 340))
    return s, cfg, mats, beams_, pose, rnd, (a0, a0 + 60)
