"""GPU tests added in round 5.

  * include_motion (the .cfg default, cfg/RadarModel.cfg:85; RadarCPU.cpp:190-196) through the BATCH path: one table of
    per-azimuth poses per frame of a batch -- every image against the oracle's orc_simulate_motion, through the batch entry
    points of one context and through rr_multi (n devices in loopback).
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import gen_oracle_images as gen  # noqa: E402
from common import golden_beams, image_diff, mats_tuple
from radarays_ros_amd import params, scenes

pytestmark = pytest.mark.gpu

# mean |f32_gpu - f32_oracle| / 255 over the image; north_star allows 1e-3.  The oracle is the build's restatement of
# RadarCPU.cpp (parity with the reference's loop itself is unpinned: the reference holds no test for it)
MEAN_DEV_TOL = 1e-5


def _sweeps(n_frames, n_angles=400):
    """n_frames sweeps of the antenna while the sensor drives and turns: [n_frames][n_angles][7]"""
    out = np.zeros((n_frames, n_angles, 7), np.float32)
    for f in range(n_frames):
        for a in range(n_angles):
            t = f + a / n_angles
            out[f, a] = scenes.yaw_pose(0.4 + 0.35 * t, 1.2 - 0.2 * t, 0.2 + 0.03 * t, 0.3 + 0.11 * t)
    return out


@pytest.fixture(scope="module")
def motion_case(native_lib, oracle):
    s = gen.two_room_scene()
    cfg = params.kaist_preset(n_reflections=3, ambient_noise=0, include_motion=True)
    mats = params.kaist_materials() + [params.PENETRABLE]
    beams = golden_beams(32)
    sweeps = _sweeps(8)
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
    want = [oracle.simulate(sc, mats_tuple(mats), s["object_materials"], cfg, beams, sweeps[f]) for f in range(len(sweeps))]
    return s, cfg, mats, beams, sweeps, want


def _setup(obj, s, cfg, mats, beams):
    obj.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    obj.set_materials(mats, s["object_materials"], 0)
    obj.set_config(cfg)
    obj.set_beam_samples(beams)


def test_include_motion_through_the_batch_path(native_lib, motion_case):
    """8 frames x 400 per-azimuth poses in ONE set of launches (rr_set_motion_poses with 8 tables +
    rr_simulate_batch_device): image f equals the oracle's orc_simulate_motion of sweep f, and equals the same sweep
    rendered alone through rr_simulate byte for byte."""
    import torch
    s, cfg, mats, beams, sweeps, want = motion_case
    c = native_lib.Context(0)
    _setup(c, s, cfg, mats, beams)
    one_by_one = []
    for f in range(len(sweeps)):
        c.set_motion_poses(sweeps[f])
        g8, gf, gst = c.simulate(sweeps[f][0], want_f32=True)
        o8, of, ost = want[f]
        assert gst["wave_passes"] == ost["wave_passes"] and gst["signals"] == ost["signals"], f
        d = image_diff(gf, of, g8, o8)
        assert d["mean_dev"] <= MEAN_DEV_TOL and d["u8_max"] <= 1, (f, d)
        one_by_one.append(g8)
    assert not np.array_equal(one_by_one[0], one_by_one[5])
    c.set_motion_poses(sweeps)                                        # 8 tables
    imgs = torch.zeros((len(sweeps), cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    c.simulate_batch_device(sweeps[:, 0], imgs.data_ptr(), st)        # the pose arguments are ignored while tables are set
    c.synchronize(st)
    got = imgs.cpu().numpy()
    for f in range(len(sweeps)):
        assert np.array_equal(got[f], one_by_one[f]), f
    # fewer tables than frames: frame f takes table f % k
    c.set_motion_poses(sweeps[:3])
    c.simulate_batch_device(sweeps[:, 0], imgs.data_ptr(), st)
    c.synchronize(st)
    got = imgs.cpu().numpy()
    for f in range(len(sweeps)):
        assert np.array_equal(got[f], one_by_one[f % 3]), f
    # an azimuth block of the batch (what a rank of the sharded loop renders)
    cols = torch.zeros((len(sweeps), 100, cfg.n_cells), dtype=torch.uint8, device="cuda:0")
    c.set_motion_poses(sweeps)
    c.simulate_batch_columns_device(sweeps[:, 0], 150, 250, cols.data_ptr(), st)
    c.synchronize(st)
    blk = cols.cpu().numpy()
    for f in range(len(sweeps)):
        assert np.array_equal(blk[f].T, one_by_one[f][:, 150:250]), f
    # host delivery path
    h = native_lib.HostImages((len(sweeps), cfg.n_cells, 400))
    c.simulate_batch_host_async(sweeps[:, 0], h.ptr, st)
    c.wait_host(h.ptr)
    for f in range(len(sweeps)):
        assert np.array_equal(h.array[f], one_by_one[f]), f
    h.close()
    # ONE table and a batch of several frames: refused (every frame would be the same sweep, the call's poses ignored)
    c.set_motion_poses(sweeps[0])
    with pytest.raises(native_lib.RRError, match="ONE per-azimuth pose table"):
        c.simulate_batch_device(sweeps[:2, 0], imgs.data_ptr(), st)
    c.simulate_batch_device(sweeps[:1, 0], imgs.data_ptr(), st); c.synchronize(st)      # one frame: what the table is for
    # a table count that is not a multiple of n_angles is refused; switching off gives static frames again
    with pytest.raises(native_lib.RRError):
        c.set_motion_poses(sweeps[0][:399]); c.simulate(sweeps[0][0])
    c.set_motion_poses(None)
    s8, _, _ = c.simulate(sweeps[0][0])
    assert not np.array_equal(s8, one_by_one[0])
    c.close()


def test_python_mirror_batches_and_sweeps(native_lib, motion_case):
    """radar.py: RadarHIP.simulateBatch / simulateSweeps (the twins of the ROS-typed adapter's methods): byte-equal to the same
    frames through simulate(), one by one."""
    from radarays_ros_amd.radar import RadarHIP
    s, cfg, mats, beams, sweeps, want = motion_case
    r = RadarHIP(s["verts"], s["faces"], s["face_object_id"])
    r.loadParams(mats, s["object_materials"], 0)
    r.updateDynCfg(cfg)
    r.setBeamSamples(beams)
    imgs = r.simulateSweeps(sweeps[:3], stamp=7.0)
    assert len(imgs) == 3 and imgs[1].header.stamp == 7.0 and imgs[0].encoding == "mono8"
    for f in range(3):
        d = np.abs(imgs[f].data.astype(int) - want[f][0].astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, f
    static = r.simulateBatch(sweeps[:3, 0], stamp=8.0)
    r.updateDynCfg(cfg.copy(include_motion=False))
    for f in range(3):
        r.updateTsm(sweeps[f, 0])
        assert np.array_equal(static[f].data, r.simulate().data), f
    assert not np.array_equal(static[0].data, imgs[0].data)


@pytest.mark.parametrize("n_dev", [1, 3, 8])
def test_include_motion_through_rr_multi(native_lib, motion_case, monkeypatch, n_dev):
    """The same 8 sweeps through rr_multi_set_motion_poses + rr_multi_simulate_batch: one device (host-delivery route) and
    the n-device plan in loopback (3 = ragged blocks, 8 = equal blocks): every device renders its azimuth block of all 8
    frames with the poses of ITS azimuths."""
    s, cfg, mats, beams, sweeps, want = motion_case
    if n_dev > 1:
        monkeypatch.setenv("RR_MULTI_LOOPBACK", "1")
    m = native_lib.MultiContext([0] * n_dev)
    _setup(m, s, cfg, mats, beams)
    m.set_motion_poses(sweeps)
    got = m.simulate_batch(sweeps[:, 0])
    for f in range(len(sweeps)):
        o8 = want[f][0]
        diff = np.abs(got[f].astype(int) - o8.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3, (n_dev, f)
    # a single frame uses table 0; off -> static
    one = m.simulate(sweeps[0][0])
    assert np.array_equal(one, got[0])
    m.set_motion_poses(None)
    assert not np.array_equal(m.simulate(sweeps[0][0]), got[0])
    m.close()


# ---------------------------------------------------------------------------------------------------------------------
# The reference's OWN parameter sets as named parity cases: each preset of cfg/*.yaml and the material
# table of config/oru4_test.yaml, values as tests/test_ref_presets.py pins them to the files, rendered for all 400
# azimuths against the oracle.  Beams are drawn the way RadarCPU.cpp:136-145 asks for them (sample_cone_local with the
# preset's width / count / distribution; seeded).
# ---------------------------------------------------------------------------------------------------------------------
U8_MISMATCH_TOL = 1e-3


def _beam_of(native_lib, cfg, seed=42):
    import math
    return native_lib.sample_cone_local(seed, cfg.beam_width * math.pi / 180.0, cfg.n_samples, cfg.beam_sample_dist,
                                        cfg.beam_sample_dist_normal_p_in_cone)


def _render_and_compare(native_lib, oracle, s, cfg, mats, objmat, beams, pose, noise=None, use_bvh=1):
    c = native_lib.Context(0)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    c.set_materials(mats, objmat, 0)
    c.set_config(cfg)
    c.set_beam_samples(beams)
    if noise is not None:
        c.set_noise_offsets(noise)
    motion = np.ndim(pose) == 2
    if motion:
        c.set_motion_poses(pose)
    g8, gf, gst = c.simulate(pose[0] if motion else pose, want_f32=True)
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=use_bvh)
    o8, of, ost = oracle.simulate(sc, mats_tuple(mats), objmat, cfg, beams, pose, noise_rnd=noise)
    assert gst["overflow"] == 0
    for k in ("wave_passes", "hits", "signals"):
        assert gst[k] == ost[k], (k, gst, ost)
    d = image_diff(gf, of, g8, o8)
    assert d["mean_dev"] <= MEAN_DEV_TOL and d["u8_max"] <= 1 and d["u8_mismatch_frac"] <= U8_MISMATCH_TOL, d
    c.close()
    return g8, gst


@pytest.mark.parametrize("scene_id", ["two_rooms", "config2", "buildings"])
def test_preset_laserlike(native_lib, oracle, scene_id):
    """cfg/mulran_kaist_dyncfg_laserlike.yaml: ONE ray per azimuth (n_samples 1, beam_width 1e-4 deg, D1), one pass, no
    smear kernel (signal_denoising 0: the max branch of RadarCPU.cpp:438-446), no noise -- the one-ray-per-azimuth corner of
    the pass-0 tiling (a wave holds one beam sample of 16 azimuths: here the ONLY sample)."""
    cfg = params.laserlike_preset()
    assert (cfg.n_samples, cfg.n_reflections, cfg.signal_denoising, cfg.ambient_noise) == (1, 1, 0, 0)
    if scene_id == "two_rooms":
        s = gen.two_room_scene()
        mats, pose = params.kaist_materials() + [params.PENETRABLE], scenes.default_pose("box12")
    elif scene_id == "config2":
        s = scenes.config_scene(2)
        mats, pose = params.kaist_materials(), scenes.default_pose(s["name"])
    else:
        s = scenes.heightfield_room(160, n_buildings=900, seed=3)
        mats, pose = params.kaist_materials() + [params.PENETRABLE], scenes.default_pose(s["name"])
    beams = _beam_of(native_lib, cfg)
    assert beams.shape == (1, 3) and abs(beams[0, 0] - 1.0) < 1e-6            # a 1.7e-6 rad cone: the boresight
    g8, st = _render_and_compare(native_lib, oracle, s, cfg, mats, s["object_materials"], beams, pose)
    assert st["wave_passes"] == 400 and st["signals"] == st["hits"] == 400     # closed scenes: every ray returns one echo
    # one echo per column, unsmeared: ONE non-zero pixel, at energy_max * signal_max = 79 (RadarCPU.cpp:453,533) -- or none
    # where the echo lies beyond the last bin (3424 x 0.0595 m = 204 m; the 420 m terrain has such azimuths, the rooms none)
    lit = (g8 > 0).sum(axis=0)
    assert np.all(lit <= 1) and np.all(g8.max(axis=0)[lit == 1] == 79) and lit.sum() >= (1 if scene_id == "config2" else 100)
    if scene_id == "two_rooms":
        assert np.all(lit == 1)


def test_preset_minimal_with_its_default_motion_and_noise(native_lib, oracle):
    """cfg/mulran_kaist_dyncfg_minimal.yaml: 10 samples in a 2 degree beam, W = 23 with mode 0.1 -> int(2.3) = 2, weaker
    Perlin noise; the file sets neither include_motion nor signal_max, so the .cfg defaults hold: per-azimuth poses
    (RadarCPU.cpp:190-196) and 120."""
    cfg = params.minimal_preset()
    assert cfg.include_motion and cfg.ambient_noise == 2 and cfg.signal_max == 120.0
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    mats = params.kaist_materials() + [params.PENETRABLE]
    z = scenes.default_pose(s["name"])[6]
    sweep = np.stack([scenes.yaw_pose(1.0 + 1.5 * t, 1.5 + 0.5 * t, z, 0.3 + 0.2 * t) for t in np.linspace(0, 1, 400)])
    noise = (np.random.RandomState(11).uniform(0, 1, 400) * 1000.0).astype(np.float32)
    g8, st = _render_and_compare(native_lib, oracle, s, cfg, mats, s["object_materials"], _beam_of(native_lib, cfg), sweep, noise)
    assert st["wave_passes"] >= 4000 and (g8 > 0).mean() > 0.9                 # the noise floor fills the image


def test_preset_kaist_at_its_own_50_samples(native_lib, oracle):
    """cfg/mulran_kaist_dyncfg.yaml as it is (the paper preset): 50 samples, 10 degrees, D3 / p 0.8, 4 passes, W = 35 mode
    0.35, Perlin noise -- on the 1M-triangle scene of config 3, whole frame."""
    cfg = params.kaist_preset()
    assert cfg.n_samples == 50 and cfg.ambient_noise == 2 and not cfg.include_motion
    s = scenes.config_scene(3)
    from common import materials_for
    noise = (np.random.RandomState(7).uniform(0, 1, 400) * 1000.0).astype(np.float32)
    g8, st = _render_and_compare(native_lib, oracle, s, cfg, materials_for(s), s["object_materials"], _beam_of(native_lib, cfg),
                                 scenes.default_pose(s["name"]), noise)
    assert st["wave_passes"] > 400 * 50


@pytest.mark.parametrize("fmt", ["dae", "obj"])
def test_oru4_test_materials_on_an_18_object_scene_through_the_c_loader(native_lib, oracle, tmp_path, fmt):
    """config/oru4_test.yaml: 5 materials (glass v = 0.03 refracts; wood / metal C = 1; stone C = 3000) mapped onto 18
    objects by `object_materials`.  The scene (same 18 objects, same order; the reference's .dae is author-local) is written
    as a map FILE and comes back through rr_load_mesh_file like the node's map does (radar_simulator.cpp:149); KAIST
    preset, 4 passes."""
    from radarays_ros_amd import meshio
    s = scenes.oru4_like_scene()
    path = str(tmp_path / ("oru4_like." + fmt))
    if fmt == "dae":
        meshio.save_dae(path, s["verts"], s["faces"], s["face_object_id"], s["object_names"])
    else:
        with open(path, "w") as f:
            for oid, name in enumerate(s["object_names"]):
                f.write("o %s-mesh\n" % name)
                sel = s["faces"][s["face_object_id"] == oid]
                used = np.unique(sel)
                for v in s["verts"][used]:
                    f.write("v %r %r %r\n" % tuple(float(x) for x in v))
                # negative (relative) indices: the group's own vertices
                remap = {int(u): k - len(used) for k, u in enumerate(used)}
                for t in sel:
                    f.write("f %d %d %d\n" % tuple(remap[int(x)] for x in t))
    m = native_lib.load_mesh_file(path)
    # (a .dae geometry is  <geometry id="X-mesh" name="X">: the loader reports the name, oru4_test.yaml's comments the id)
    assert m["n_objects"] == 18 and [n.replace("-mesh", "") for n in m["object_names"]] == scenes.ORU4_OBJECT_NAMES
    assert len(m["faces"]) == len(s["faces"]) and np.array_equal(np.bincount(m["face_object_id"]), np.bincount(s["face_object_id"]))
    mats, objmat = params.oru4_test_materials(), params.ORU4_OBJECT_MATERIALS
    assert len(objmat) == m["n_objects"]
    cfg = params.kaist_preset(ambient_noise=0, n_samples=40)
    pose = scenes.default_pose(s["name"])
    loaded = {"verts": m["verts"], "faces": m["faces"], "face_object_id": m["face_object_id"]}
    g8, st = _render_and_compare(native_lib, oracle, loaded, cfg, mats, objmat, golden_beams(40), pose, use_bvh=0)
    # glass transmits: more waves than reflections alone would give (an opaque scene has <= 400 * 40 * 4)
    opaque = [params.RadarMaterial(0.0, m_.ambient, m_.diffuse, m_.specular) if i else m_ for i, m_ in enumerate(mats)]
    c = native_lib.Context(0)
    c.set_mesh(loaded["verts"], loaded["faces"], loaded["face_object_id"])
    c.set_materials(opaque, objmat, 0); c.set_config(cfg); c.set_beam_samples(golden_beams(40))
    o8, _, ost = c.simulate(pose)
    c.close()
    assert st["wave_passes"] > ost["wave_passes"] and not np.array_equal(o8, g8)


@pytest.mark.parametrize("table", ["oru4_legacy", "oru3_legacy"])
def test_legacy_material_tables_on_the_18_object_scene(native_lib, oracle, table):
    """The reference's legacy material tables (config/oru4.yaml, config/oru3.yaml; values pinned by tests/test_ref_presets.py)
    on the 18-object scene, 5 passes: velocities of 0.001 / 0.002 m/ns (n21 = 300), eleven transmitting geological materials,
    BRDF exponents of 0 (cos^0 = 1), 0.1 and 2000, ambient 0.01 -- wave, hit and signal counts exact, deviation <= 1e-5."""
    s = scenes.oru4_like_scene()
    if table == "oru4_legacy":
        mats, objmat = params.oru4_legacy_materials(), params.ORU4_OBJECT_MATERIALS
    else:
        mats = params.oru3_legacy_materials()
        objmat = [1 + (k * 5) % 12 for k in range(18)]          # every one of the 12 non-air materials is on some object
        assert set(objmat) == set(range(1, 13))
    cfg = params.kaist_preset(ambient_noise=0, n_samples=40, n_reflections=5)
    g8, st = _render_and_compare(native_lib, oracle, s, cfg, mats, objmat, golden_beams(40), scenes.default_pose(s["name"]), use_bvh=0)
    assert st["wave_passes"] > 400 * 40 * 2 and (g8 > 0).mean() > 0.01


# ---------------------------------------------------------------------------------------------------------------------
# Tight later-pass trace rows (rr_get_trace_grid): images never depend on the history
# ---------------------------------------------------------------------------------------------------------------------
def _batch(c, poses, cfg):
    import torch
    imgs = torch.zeros((len(poses), cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    c.simulate_batch_device(poses, imgs.data_ptr(), st)
    c.synchronize(st)
    return imgs.cpu().numpy()


def test_tight_trace_rows_follow_the_history_and_never_change_an_image(native_lib, monkeypatch):
    """Later-pass k_trace launches are as wide as earlier batches needed.  (1) the first batch after a change runs at the
    doubling bound, later ones tighter, same bytes as with RR_TIGHT_GRID=0; (2) a pose whose segments need MORE than the
    history (sensor moved between the buildings: more transmitted waves) overflows the tightened rows: the repair launch
    traces the rest (repaired > 0), same bytes; afterwards the history has grown and nothing is repaired any more;
    (3) RR_TIGHT_FORCE=1: one workgroup per row, nearly every ray through the repair path, same bytes."""
    s = scenes.heightfield_room(96, n_buildings=260, seed=3)
    from common import materials_for
    cfg = params.kaist_preset(n_reflections=4, n_samples=64, ambient_noise=0)
    z = scenes.default_pose(s["name"])[6]
    calm = [scenes.yaw_pose(1.0 + 0.2 * k, 1.5, z + 14.0, 0.3) for k in range(4)]          # high above the roofs: terrain hits, few children
    busy = [scenes.yaw_pose(60.0 + 3.0 * k, -40.0, z + 1.0, 0.3 + k) for k in range(4)]    # street level between buildings

    def ctx():
        c = native_lib.Context(0)
        c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
        c.set_materials(materials_for(s), s["object_materials"], 0)
        c.set_config(cfg)
        c.set_beam_samples(golden_beams(64))
        return c
    monkeypatch.delenv("RR_TIGHT_FORCE", raising=False)                    # (the suite may run under it)
    monkeypatch.delenv("RR_STACK_LDS", raising=False)                      # (... or under the spill path, whose launches keep full rows)
    monkeypatch.setenv("RR_TIGHT_GRID", "0")
    c0 = ctx()
    want_calm, want_busy = _batch(c0, calm, cfg), _batch(c0, busy, cfg)
    assert not c0.trace_grid()[0].any()
    c0.close()
    monkeypatch.delenv("RR_TIGHT_GRID")
    c = ctx()
    assert np.array_equal(_batch(c, calm, cfg), want_calm)
    rows, hist, rep = c.trace_grid()
    assert not rows.any() and rep == 0 and hist[1] > 0                     # first batch: no history yet, full rows
    assert np.array_equal(_batch(c, calm, cfg), want_calm)
    rows, hist, rep = c.trace_grid()
    full = [(min(64 << p, 64 << 3) + 15) // 16 for p in range(4)]
    assert rep == 0 and any(0 < rows[p] < full[p] for p in (2, 3)), (rows[:4], hist[:4])
    calm_hist = hist.copy()
    assert np.array_equal(_batch(c, busy, cfg), want_busy)                 # needs more than the calm poses taught
    rows2, hist2, rep2 = c.trace_grid()
    assert rep2 > 0 and (hist2[:4] >= calm_hist[:4]).all() and (hist2[:4] > calm_hist[:4]).any(), (rows2[:4], hist2[:4], rep2)
    assert np.array_equal(_batch(c, busy, cfg), want_busy)
    assert c.trace_grid()[2] == rep2                                       # the history has grown: nothing left to repair
    assert np.array_equal(_batch(c, calm, cfg), want_calm)
    # a change of the beam starts the history over
    c.set_beam_samples(golden_beams(48))
    _batch(c, calm, cfg)
    assert not c.trace_grid()[0].any() and c.trace_grid()[2] == 0
    c.close()
    monkeypatch.setenv("RR_TIGHT_FORCE", "1")
    c1 = ctx()
    assert np.array_equal(_batch(c1, busy, cfg), want_busy) and np.array_equal(_batch(c1, calm, cfg), want_calm)
    rows, _, rep = c1.trace_grid()
    assert rows[2] == 1 and rows[3] == 1 and rep > 1000
    g8, _, st = c1.simulate(busy[0])                                       # the synchronous entry point takes the same route
    assert np.array_equal(g8, want_busy[0]) and st["overflow"] == 0
    c1.close()


# ---------------------------------------------------------------------------------------------------------------------
# Launch graphs (rr_get_graph_stats): a replayed chain renders the same bytes as the chain issued kernel by kernel
# ---------------------------------------------------------------------------------------------------------------------
def test_launch_graphs_replay_identical_images_and_are_dropped_on_any_change(native_lib, monkeypatch):
    """A pose batch issued a second time with the same shape is captured in a hipGraph, later calls replay it with new
    poses (the parameters of the graph's first node): every image equals the kernel-by-kernel render (RR_GRAPHS=0).
    A change of materials / config / beam / mesh drops the graphs (captured launches hold the old tables), a different
    azimuth block or frame count is its own graph, instrumented runs are never replayed."""
    import torch
    s = gen.two_room_scene()
    cfg = params.kaist_preset(n_reflections=3, ambient_noise=2)
    mats = params.kaist_materials() + [params.PENETRABLE]
    noise = (np.random.RandomState(3).uniform(0, 1, (4, 400)) * 1000.0).astype(np.float32)
    poses = scenes.trajectory(12, "box12")
    batches = [poses[0:4], poses[4:8], poses[8:12], poses[2:6], poses[5:9]]

    def ctx():
        c = native_lib.Context(0)
        c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
        c.set_materials(mats, s["object_materials"], 0)
        c.set_config(cfg)
        c.set_beam_samples(golden_beams(32))
        c.set_noise_offsets(noise)
        return c
    monkeypatch.setenv("RR_GRAPHS", "0")
    monkeypatch.setenv("RR_LANES", "1")          # one frame lane: every batch meets the same buffers (and the same graph)
    c0 = ctx()
    want = [_batch(c0, b, cfg) for b in batches]
    assert c0.graph_stats() == (0, 0)
    monkeypatch.delenv("RR_GRAPHS")
    c = ctx()
    imgs = torch.zeros((4, cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream

    def run(b):
        c.simulate_batch_device(b, imgs.data_ptr(), st)
        c.synchronize(st)
        return imgs.cpu().numpy()
    for rep in range(3):
        for k, b in enumerate(batches):
            assert np.array_equal(run(b), want[k]), (rep, k)
    cap, rep = c.graph_stats()
    assert cap >= 1 and rep >= 10, (cap, rep)                      # (trace rows settle within the first calls: a few captures)
    # other shapes are other graphs; the first use of a shape is never a replay
    cols = torch.zeros((4, 100, cfg.n_cells), dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        c.simulate_batch_columns_device(batches[0], 100, 200, cols.data_ptr(), st)
        c.synchronize(st)
        assert np.array_equal(np.transpose(cols.cpu().numpy(), (0, 2, 1)), want[0][:, :, 100:200])
    one = torch.zeros((1, cfg.n_cells, 400), dtype=torch.uint8, device="cuda:0")
    for _ in range(3):
        c.simulate_batch_device(batches[1][:1], one.data_ptr(), st); c.synchronize(st)
        assert np.array_equal(one.cpu().numpy()[0], want[1][0])
    cap2, rep2 = c.graph_stats()
    assert cap2 >= cap + 2 and rep2 >= rep + 2
    # a change of the materials drops what was captured: same poses, other image, equal to a fresh kernel-by-kernel render
    other = [mats[0], mats[1], params.RadarMaterial(0.2, 0.5, 0.4, 5.0)]
    c.set_materials(other, s["object_materials"], 0)
    c0.set_materials(other, s["object_materials"], 0)
    want_other = _batch(c0, batches[0], cfg)
    for _ in range(3):
        assert np.array_equal(run(batches[0]), want_other)
    assert not np.array_equal(want_other, want[0]) and c.graph_stats()[0] > cap2
    # fresh noise offsets of the same shape (what a node sets before every frame, RadarCPU.cpp:461-472) keep the graphs: only
    # the contents of a buffer they already point at change -- and the images follow the new offsets
    noise2 = (np.random.RandomState(4).uniform(0, 1, (4, 400)) * 1000.0).astype(np.float32)
    c.set_noise_offsets(noise2); c0.set_noise_offsets(noise2)
    want_n2 = _batch(c0, batches[0], cfg)
    before = c.graph_stats()
    for _ in range(2):
        assert np.array_equal(run(batches[0]), want_n2)
    after = c.graph_stats()
    assert after[0] == before[0] and after[1] == before[1] + 2 and not np.array_equal(want_n2, want_other)
    c.set_noise_offsets(noise2[:1]); c0.set_noise_offsets(noise2[:1])          # another shape: one row -> captured again
    want_n1 = _batch(c0, batches[0], cfg)
    for _ in range(3):
        assert np.array_equal(run(batches[0]), want_n1)
    assert c.graph_stats()[0] == after[0] + 1
    # instrumented runs are issued kernel by kernel
    c.set_timing_mode(1)
    before = c.graph_stats()
    for _ in range(3):
        assert np.array_equal(run(batches[0]), want_n1)
    assert c.graph_stats() == before
    c.set_timing_mode(0)
    # an overflow inside a replayed chain is still reported
    c.set_config(cfg, 400, max_waves_per_azimuth=33)
    for k in range(3):
        c.simulate_batch_device(batches[0], imgs.data_ptr(), st)
        with pytest.raises(native_lib.RRError, match="capacity"):
            c.synchronize(st)
    c.close(); c0.close()


# ---------------------------------------------------------------------------------------------------------------------
# Later-pass trace grids in chunks of 16 azimuths (RR_TRACE_CHUNK): a launch shape, never a different image
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_chunked_trace_grid_is_only_a_launch_shape(native_lib, monkeypatch):
    """The later passes walk chunks of S neighbouring segments with the segment as the fast grid dimension.  Same bytes as one
    grid row per segment (RR_TRACE_CHUNK=0) for (1) a 64-frame batch of a 4-pass config -- 25,600 segments: the chunked grid of
    the late passes would exceed the 65,535 rows a grid may have, so those launches fall back to the plain layout while
    the early ones are chunked; (2) an azimuth block whose segment count is not a multiple of the chunk (7 columns x 3
    frames: the last chunk is ragged); (3) S = 64 and S = 8."""
    import torch
    s = scenes.heightfield_room(64, n_buildings=120, seed=5)
    from common import materials_for
    cfg = params.kaist_preset(n_reflections=4, n_samples=100, ambient_noise=0)
    z = scenes.default_pose(s["name"])[6]
    poses64 = [scenes.yaw_pose(-30.0 + 1.1 * k, 12.0 - 0.4 * k, z + 1.0 + 0.05 * k, 0.1 * k) for k in range(64)]

    def render(chunk):
        monkeypatch.setenv("RR_TRACE_CHUNK", str(chunk))
        c = native_lib.Context(0)
        c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
        c.set_materials(materials_for(s), s["object_materials"], 0)
        c.set_config(cfg)
        c.set_beam_samples(golden_beams(100))
        big = [_batch(c, poses64, cfg) for _ in range(2)]                   # second call: tightened rows
        cols = torch.zeros((3, 7, cfg.n_cells), dtype=torch.uint8, device="cuda:0")
        st = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            c.simulate_batch_columns_device(poses64[:3], 100, 107, cols.data_ptr(), st)
            torch.cuda.synchronize()
        c.synchronize()
        out = big, cols.cpu().numpy().copy()
        c.close()
        return out
    monkeypatch.delenv("RR_TIGHT_FORCE", raising=False)
    (w0, w1), wc = render(0)
    assert np.array_equal(w0, w1) and w0.any() and wc.any()
    for chunk in (16, 64, 8):
        (g0, g1), gc = render(chunk)
        assert np.array_equal(g0, w0) and np.array_equal(g1, w0) and np.array_equal(gc, wc), chunk
