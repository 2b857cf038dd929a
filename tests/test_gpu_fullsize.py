"""BASELINE.json's full sizes (configs[2..3] scale: 10M triangles, 200 / 1000 rays, 4 passes)
through size-independent properties + an oracle spot check on a few azimuths."""
import numpy as np
import pytest

from common import golden_beams, image_diff, materials_for, mats_tuple
from radarays_ros_amd import params, scenes
from radarays_ros_amd.dist import partition

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(native_lib):
    s = scenes.config_scene(4)                     # 10,248,350 triangles
    c = native_lib.Context(0)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    c.set_materials(materials_for(s), s["object_materials"], 0)
    yield s, c
    c.close()


def test_target_config_properties(big, native_lib):
    s, c = big
    cfg = params.kaist_preset(n_reflections=4, ambient_noise=0)
    c.set_config(cfg)
    c.set_beam_samples(golden_beams(200))
    pose = scenes.default_pose(s["name"])
    info = c.bvh_info()
    assert len(s["faces"]) == 10248350
    # leaf triangle records: one per face + the parts spatial splits cut (bounded by the builder's reference budget)
    assert len(s["faces"]) <= info["n_tris"] <= 2 * len(s["faces"]) + 16
    full, _, st = c.simulate(pose)
    assert st["overflow"] == 0
    # closed scene: every pass-0 wave hits; counts are ordered
    assert st["wave_passes"] >= 80000 and st["hits"] <= st["wave_passes"] and st["signals"] <= st["hits"]
    assert st["hits"] > 0.99 * st["wave_passes"]
    # normalisation: every column peaks at round(energy_max * signal_max) = 79 (Appendix A.8)
    assert np.all(full.max(axis=0) == 79)
    # bin 0 is never written (RadarCPU.cpp:424)
    assert not full[0].any()
    # determinism (no atomics in the image path)
    again, _, st2 = c.simulate(pose)
    assert np.array_equal(full, again) and st2 == st
    # azimuth sharding as 8 ranks would do it == the full frame
    parts = np.zeros_like(full)
    wp = 0
    for r in range(8):
        b, e = partition(400, 8, r)
        p, _, sr = c.simulate(pose, b, e)
        parts[:, b:e] = p[:, b:e]
        wp += sr["wave_passes"]
    assert np.array_equal(full, parts) and wp == st["wave_passes"]
    # scroll_image only rotates columns
    c.set_config(cfg.copy(scroll_image=123))
    rolled, _, _ = c.simulate(pose)
    assert np.array_equal(np.roll(full, 123, axis=1), rolled)


def test_target_config_full_frame_vs_oracle(big, native_lib, oracle):
    s, c = big
    cfg = params.kaist_preset(n_reflections=4, ambient_noise=0)
    c.set_config(cfg)
    c.set_beam_samples(golden_beams(200))
    pose = scenes.default_pose(s["name"])
    # the WHOLE frame of the north-star target (10M triangles, 4 passes, 200 rays) against the oracle
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=1)
    g8, gf, gst = c.simulate(pose, want_f32=True)
    o8, of, ost = oracle.simulate(sc, mats_tuple(materials_for(s)), s["object_materials"], cfg, golden_beams(200), pose)
    assert gst["wave_passes"] == ost["wave_passes"] and gst["hits"] == ost["hits"] and gst["signals"] == ost["signals"]
    d = image_diff(gf, of, g8, o8)
    assert d["mean_dev"] <= 1e-5 and d["u8_max"] <= 1 and d["u8_mismatch_frac"] <= 1e-3, d


def test_bench_step_of_the_target_against_oracle(big, native_lib, oracle):
    """What `python bench.py` times by default, checked directly: the 16-pose trajectory of the north-star target
    (10M triangles, 400 x 200 rays, 4 passes, Perlin noise with one row of offsets per frame), rendered by the
    batch entry point the bench step uses (8 poses per call), delivered to host memory like the reference's
    m_polar_image; every mono8 image against the oracle's frame for that pose and that noise row."""
    s, c = big
    cfg = params.kaist_preset(n_reflections=4, n_samples=200, ambient_noise=2)
    mats = materials_for(s)
    noise = (np.random.RandomState(7).uniform(0, 1, (16, 400)) * 1000.0).astype(np.float32)
    poses = scenes.trajectory(16, s["name"])
    c.set_config(cfg)
    c.set_beam_samples(golden_beams(200))
    c.set_noise_offsets(noise)
    host = native_lib.HostImages((16, cfg.n_cells, 400))
    for k in range(2):       # a bench step = 8 poses; frame f of a batch uses noise row f % 16
        c.set_noise_offsets(np.roll(noise, -8 * k, axis=0))
        c.simulate_batch_host_async(poses[8 * k:8 * k + 8], host.ptr + 8 * k * cfg.n_cells * 400, None)
        c.wait_host(None)
    c.synchronize()
    assert c.stats()["overflow"] == 0
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=1)
    for f, p in enumerate(poses):
        o8, _, _ = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, golden_beams(200), p,
                                   noise_rnd=noise[f], want_f32=False)
        d8 = np.abs(host.array[f].astype(np.int32) - o8.astype(np.int32))
        assert d8.max() <= 1 and (d8 > 0).mean() <= 1e-3, (f, int(d8.max()), float((d8 > 0).mean()))
    host.close()


def test_config4_1000_rays(big, native_lib, oracle):
    """configs[3] on one GPU: 400 x 1000 rays, 4 passes, 10M triangles: properties of the whole frame, the frame
    sharded as 8 ranks would, and two azimuth pairs against the oracle (counts exact, mean deviation <= 1e-5)."""
    s, c = big
    cfg = params.kaist_preset(n_reflections=4, n_samples=1000, ambient_noise=2)
    c.set_config(cfg)
    c.set_beam_samples(golden_beams(1000))
    rnd = (np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32)
    c.set_noise_offsets(rnd)
    pose = scenes.default_pose(s["name"])
    img, _, st = c.simulate(pose)
    assert st["overflow"] == 0 and st["wave_passes"] >= 400000 and st["hits"] > 0.99 * st["wave_passes"]
    assert img.max() <= 255 and (img > 0).mean() > 0.5      # noise floor fills the image
    again, _, _ = c.simulate(pose)
    assert np.array_equal(img, again)
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=1)
    mats = materials_for(s)
    for az in ((11, 13), (301, 303)):
        g8, gf, gst = c.simulate(pose, az[0], az[1], want_f32=True)
        assert np.array_equal(g8[:, az[0]:az[1]], img[:, az[0]:az[1]])       # a shard == the same columns of the frame
        o8, of, ost = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, golden_beams(1000), pose,
                                      noise_rnd=rnd, az_begin=az[0], az_end=az[1])
        assert gst["wave_passes"] == ost["wave_passes"] and gst["hits"] == ost["hits"] and gst["signals"] == ost["signals"], (gst, ost)
        d = image_diff(gf[:, az[0]:az[1]], of[:, az[0]:az[1]], g8[:, az[0]:az[1]], o8[:, az[0]:az[1]])
        assert d["mean_dev"] <= 1e-5 and d["u8_max"] <= 1, d


def test_config5_per_triangle_materials_8_passes(big, native_lib, oracle):
    """configs[4] at full size: 10M triangles, one of 8 materials PER TRIANGLE (seed 5), 8 passes, Cook-Torrance
    lobe (rr_config.brdf_model = 1: the build's own specification, the reference's is on its dev/flex branch
    outside the checkout -- parity unpinned).  Properties on the whole frame + the oracle's twin on two azimuths."""
    _, c = big
    s = scenes.config_scene(5)
    mats = materials_for(s)
    assert len(mats) == 9 and len(set(s["face_object_id"].tolist())) == 8
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    c.set_materials(mats, s["object_materials"], 0)
    cfg = params.kaist_preset(n_reflections=8, ambient_noise=0)
    c.set_config(cfg, 400, brdf_model=1)
    c.set_beam_samples(golden_beams(200))
    pose = scenes.default_pose(s["name"])
    full, _, st = c.simulate(pose)
    assert st["overflow"] == 0
    # penetrable triangles split the waves: clearly more wave-passes than 8 passes of 80k unsplit rays would
    # lose to pruning, yet bounded by the doubling bound
    assert 80000 * 2 < st["wave_passes"] <= 80000 * 255
    assert np.all(full.max(axis=0) == 79) and not full[0].any()
    again, _, st2 = c.simulate(pose)
    assert np.array_equal(full, again) and st2 == st
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=1)
    for az in ((11, 13), (301, 303)):
        g8, gf, gst = c.simulate(pose, az[0], az[1], want_f32=True)
        o8, of, ost = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg,
                                      golden_beams(200), pose, az_begin=az[0], az_end=az[1], brdf_model=1)
        assert gst["wave_passes"] == ost["wave_passes"] and gst["signals"] == ost["signals"], (gst, ost)
        d = image_diff(gf[:, az[0]:az[1]], of[:, az[0]:az[1]], g8[:, az[0]:az[1]], o8[:, az[0]:az[1]])
        assert d["mean_dev"] <= 1e-5 and d["u8_max"] <= 1, d


def test_config5_bench_workload_1000_rays_8_passes(big, native_lib, oracle):
    """configs[4] exactly as `bench.py --workload config5_10M_400x1000_8pass_pertri` runs it ("as 4" = 1000 rays
    per beam, 8 passes, per-triangle materials, Cook-Torrance lobe, Perlin noise): 1000 x 2^7 exceeds the 65,536-wave
    clamp of the per-azimuth queue (radarays_mi355.h: max_waves_per_azimuth), so this is the case where the clamp could
    bite -- it must not (no overflow reported on the whole frame), and two azimuth pairs equal the oracle's twin, whose
    queues are unbounded: counts exact, mean deviation <= 1e-5."""
    _, c = big
    s = scenes.config_scene(5)
    mats = materials_for(s)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    c.set_materials(mats, s["object_materials"], 0)
    cfg = params.kaist_preset(n_reflections=8, n_samples=1000, ambient_noise=2)
    c.set_config(cfg, 400, brdf_model=1)
    c.set_beam_samples(golden_beams(1000))
    rnd = (np.random.RandomState(7).uniform(0, 1, 400) * 1000).astype(np.float32)
    c.set_noise_offsets(rnd)
    pose = scenes.default_pose(s["name"])
    full, _, st = c.simulate(pose)
    assert st["overflow"] == 0, st
    assert 400000 * 2 < st["wave_passes"] <= 400000 * 255
    # the busiest azimuth stays far below the clamp: the frame's wave-passes spread over 400 azimuths and 8 passes
    assert st["wave_passes"] / 400.0 < 8 * 65536
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=1)
    for az in ((11, 13), (301, 303)):
        g8, gf, gst = c.simulate(pose, az[0], az[1], want_f32=True)
        assert gst["overflow"] == 0
        assert np.array_equal(g8[:, az[0]:az[1]], full[:, az[0]:az[1]])
        o8, of, ost = oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, golden_beams(1000), pose,
                                      noise_rnd=rnd, az_begin=az[0], az_end=az[1], brdf_model=1)
        assert gst["wave_passes"] == ost["wave_passes"] and gst["hits"] == ost["hits"] and gst["signals"] == ost["signals"], (gst, ost)
        d = image_diff(gf[:, az[0]:az[1]], of[:, az[0]:az[1]], g8[:, az[0]:az[1]], o8[:, az[0]:az[1]])
        assert d["mean_dev"] <= 1e-5 and d["u8_max"] <= 1, d
    # restore the fixture's scene for whoever comes next
    s4 = scenes.config_scene(4)
    c.set_mesh(s4["verts"], s4["faces"], s4["face_object_id"])
    c.set_materials(materials_for(s4), s4["object_materials"], 0)
