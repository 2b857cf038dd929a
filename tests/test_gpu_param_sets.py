"""Parameter batches as the reference's optimiser uses the simulator (scripts/radaray_opti.py:36-113,170-211;
action/GenRadarImage.action): one pose, many parameter vectors {beam_width, n_reflections, material table}, one number
per evaluation (minus the PSNR against one real image).  rr_simulate_param_sets renders the sets in one set of launches
(pass 0 shared inside a beam group, sets with fewer passes stop early); rr_score_images_device scores them on the GPU."""
import math

import numpy as np
import pytest

from common import golden_beams, image_diff, materials_for, mats_tuple
from radarays_ros_amd import params, scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    s = scenes.heightfield_room(64, n_buildings=40, seed=3)
    cfg = params.kaist_preset(n_reflections=3, n_samples=60, ambient_noise=2)
    noise = (np.random.RandomState(5).uniform(0, 1, 400) * 1000.0).astype(np.float32)
    return s, cfg, materials_for(s), noise, scenes.trajectory(4, s["name"])


def _ctx(native_lib, world, beams):
    s, cfg, mats, noise, _ = world
    c = native_lib.Context(0)
    c.set_mesh(s["verts"], s["faces"], s["face_object_id"])
    c.set_materials(mats, s["object_materials"], 0)
    c.set_config(cfg, 400)
    c.set_beam_samples(beams)
    c.set_noise_offsets(noise)
    return c


def _psnr_numpy(ref, img):
    # skimage.metrics.peak_signal_noise_ratio for uint8 images: data_range 255, float64 mean squared error
    err = np.mean((ref.astype(np.float64) - img.astype(np.float64)) ** 2)
    return np.inf if err == 0 else 10.0 * np.log10((255.0 ** 2) / err)


def test_param_sets_equal_one_by_one_and_oracle(native_lib, oracle, world):
    """Sets that differ in everything the optimiser varies: three beam tables (two widths drawn by rr_sample_cone_local
    + the context's own samples; two sets share a table = one pass-0 group), 0..4 passes (more than the config's 3 as
    well), material tables.  Image k == rr_set_materials / rr_set_beam_samples / rr_set_config(set k) + rr_simulate, bit
    for bit; two of them against the oracle; the scores equal numpy's PSNR."""
    s, cfg, mats, noise, poses = world
    base = golden_beams(60)
    wide = native_lib.sample_cone_local(11, math.radians(14.0), 60, 2, 0.8)
    narrow = native_lib.sample_cone_local(12, math.radians(4.0), 60, 2, 0.8)
    m0 = np.array(mats_tuple(mats), np.float32)
    m1 = m0.copy(); m1[1:, 1] *= 0.5; m1[2:, 0] = 0.05          # ambient halved, a slower penetrable material
    m2 = m0.copy(); m2[1:, 2] = 0.4; m2[1:, 3] = 12.0          # a diffuse lobe
    sets = [
        {"materials": m0, "beam_dirs": None, "n_reflections": None},
        {"materials": m1, "beam_dirs": wide, "n_reflections": 2},
        {"materials": m2, "beam_dirs": narrow, "n_reflections": 4},
        {"materials": None, "beam_dirs": wide.copy(), "n_reflections": 1},      # same bytes as set 1's table: same group
        {"materials": m1, "beam_dirs": base.copy(), "n_reflections": 0},          # no ray cast: noise floor only
        {"materials": m2, "beam_dirs": None, "n_reflections": 3},
    ]
    c = _ctx(native_lib, world, base)
    imgs, _ = c.simulate_param_sets(poses[1], sets, len(mats))
    assert imgs.shape == (len(sets), cfg.n_cells, 400)
    ref_ctx = _ctx(native_lib, world, base)
    one = []
    for k, st in enumerate(sets):
        ref_ctx.set_materials([tuple(r) for r in (st["materials"] if st["materials"] is not None else m0)], s["object_materials"], 0)
        ref_ctx.set_beam_samples(st["beam_dirs"] if st["beam_dirs"] is not None else base)
        ref_ctx.set_config(cfg.copy(n_reflections=cfg.n_reflections if st["n_reflections"] is None else st["n_reflections"]), 400)
        u8, _, stt = ref_ctx.simulate(poses[1])
        assert stt["overflow"] == 0
        one.append(u8)
        assert np.array_equal(imgs[k], u8), k
    assert not np.array_equal(imgs[1], imgs[3]) and not np.array_equal(imgs[0], imgs[5])
    # the batch leaves the context's own state alone
    again, _, _ = c.simulate(poses[1])
    assert np.array_equal(again, one[0])
    # oracle: a wide-beam 2-pass set and the narrow 4-pass one
    sc = oracle.Scene(s["verts"], s["faces"], s["face_object_id"], use_bvh=0)
    for k in (1, 2):
        st = sets[k]
        o8, _, _ = oracle.simulate(sc, [tuple(float(x) for x in r) for r in st["materials"]], s["object_materials"],
                                   cfg.copy(n_reflections=st["n_reflections"]), st["beam_dirs"], poses[1], noise_rnd=noise, want_f32=False)
        d8 = np.abs(imgs[k].astype(np.int32) - o8.astype(np.int32))
        assert d8.max() <= 1 and (d8 > 0).mean() <= 1e-3, (k, int(d8.max()), float((d8 > 0).mean()))
    # scores: the "real" image is set 0's; no image needs to leave the GPU for them
    real = one[0]
    none, psnr = c.simulate_param_sets(poses[1], sets, len(mats), ref_u8=real, want_images=False)
    assert none is None and psnr.shape == (len(sets),)
    assert np.isinf(psnr[0]) and psnr[0] > 0
    for k in range(1, len(sets)):
        assert abs(psnr[k] - _psnr_numpy(real, one[k])) <= 1e-9, (k, psnr[k])
    both, psnr2 = c.simulate_param_sets(poses[1], sets, len(mats), ref_u8=real)
    assert np.array_equal(both, imgs) and np.array_equal(psnr, psnr2)
    c.close(); ref_ctx.close()


def test_score_images_device_is_exact(native_lib, world):
    """rr_score_images_device: the integer sum of squared differences is exact (odd image count, images that differ in a
    handful of pixels, unaligned tail handled), PSNR = 10 log10(255^2 / mse) as skimage computes it."""
    import torch
    s, cfg, mats, noise, poses = world
    c = _ctx(native_lib, world, golden_beams(60))
    rs = np.random.RandomState(3)
    ref = rs.randint(0, 256, (cfg.n_cells, 400)).astype(np.uint8)
    imgs = np.stack([ref.copy() for _ in range(5)])
    imgs[1, 7, 9] ^= 0xFF
    imgs[2] = rs.randint(0, 256, ref.shape).astype(np.uint8)
    imgs[3] = 255 - ref
    imgs[4, -1, -1] = (int(ref[-1, -1]) + 1) % 256
    d_imgs = torch.from_numpy(imgs).cuda(); d_ref = torch.from_numpy(ref).cuda()
    psnr, sse = c.score_images_device(d_imgs.data_ptr(), 5, d_ref.data_ptr(), want_sse=True)
    want = [int(((imgs[k].astype(np.int64) - ref.astype(np.int64)) ** 2).sum()) for k in range(5)]
    assert [int(x) for x in sse] == want
    for k in range(5):
        assert (np.isinf(psnr[k]) and want[k] == 0) or abs(psnr[k] - _psnr_numpy(ref, imgs[k])) <= 1e-9
    c.close()


def test_param_sets_are_validated(native_lib, world):
    s, cfg, mats, noise, poses = world
    c = _ctx(native_lib, world, golden_beams(60))
    m0 = np.array(mats_tuple(mats), np.float32)
    with pytest.raises(native_lib.RRError, match="as many materials"):
        c.simulate_param_sets(poses[0], [{"materials": m0[:2]}], 2)
    with pytest.raises(native_lib.RRError, match="n_reflections must be <= 16"):
        c.simulate_param_sets(poses[0], [{"n_reflections": 17}], len(mats))
    bad = m0.copy(); bad[1, 2] = np.nan
    with pytest.raises(native_lib.RRError, match="non-finite material"):
        c.simulate_param_sets(poses[0], [{"materials": bad}], len(mats))
    with pytest.raises(native_lib.RRError, match="n_sets must be 1..64"):
        c.simulate_param_sets(poses[0], [{}] * 65, len(mats))
    with pytest.raises(native_lib.RRError, match="go together|neither"):
        c.simulate_param_sets(poses[0], [{}], len(mats), want_images=False)
    c.close()


def test_python_twin_simulate_param_sets(native_lib, world):
    """radar.RadarHIP.simulateParamSets (the Python mirror of the C++ host class): RadarParams in, images + PSNR out;
    a set equals the same parameters applied through updateDynCfg / loadParams + simulate()."""
    import copy
    from radarays_ros_amd import radar
    from radarays_ros_amd.params import RadarModelConfig
    s, cfg, mats, noise, poses = world
    r = radar.RadarHIP(s["verts"], s["faces"], s["face_object_id"], beam_seed=9)
    r.loadParams(mats, s["object_materials"], 0)
    dyn = cfg.copy(beam_width=10.0, n_samples=60)
    r.updateDynCfg(dyn)
    r.setNoiseOffsets(noise)
    r.updateTsm(poses[2])
    cur = r.simulate(1.0)
    p0 = copy.deepcopy(r.getParams())
    p1 = copy.deepcopy(p0); p1.model.beam_width = float(np.float32(6.0 * np.pi / 180.0)); p1.model.n_reflections = 2
    p2 = copy.deepcopy(p0); p2.model.n_reflections = 1
    for m in p2.materials[1:]:
        m.ambient *= 0.5
    msgs, psnr = r.simulateParamSets([p0, p1, p2], stamp=2.0, real=cur.data)
    assert len(msgs) == 3 and np.array_equal(msgs[0].data, cur.data) and np.isinf(psnr[0])
    assert msgs[1].header.stamp == 2.0 and msgs[1].encoding == "mono8"
    assert all(np.isfinite(psnr[1:])) and not np.array_equal(msgs[1].data, msgs[2].data)
    r.updateDynCfg(dyn.copy(beam_width=6.0, n_reflections=2))
    one = r.simulate(3.0)
    assert np.array_equal(one.data, msgs[1].data)
    none, only = r.simulateParamSets([p0, p1, p2], real=cur.data, want_images=False)
    assert none is None and np.array_equal(only[1:], psnr[1:])


def test_param_batches_of_changing_depth_on_one_context(native_lib, world):
    """An optimiser's calls bring different largest numbers of passes from one call to the next: the lane's wave queues
    are sized by the largest seen and re-used for shallower batches (no re-allocation per call), ordinary frames in
    between go back to the exact layout -- every image still equals the one-by-one path."""
    s, cfg, mats, noise, poses = world
    base = golden_beams(60)
    c, ref = _ctx(native_lib, world, base), _ctx(native_lib, world, base)
    m0 = np.array(mats_tuple(mats), np.float32)

    def check(depths, pose):
        sets = [{"materials": m0, "n_reflections": d} for d in depths]
        imgs, _ = c.simulate_param_sets(pose, sets, len(mats))
        for k, d in enumerate(depths):
            ref.set_config(cfg.copy(n_reflections=d), 400)
            assert np.array_equal(imgs[k], ref.simulate(pose)[0]), (depths, k)
    for lane_round in range(2):                # 4 frame lanes: every lane sees a deep batch, then shallower ones
        for depths in ([5, 1], [2, 2, 0], [3], [4, 1, 2]):
            check(depths, poses[lane_round])
        ref.set_config(cfg, 400)
        assert np.array_equal(c.simulate(poses[3])[0], ref.simulate(poses[3])[0])
        check([1, 3], poses[2])
    c.close(); ref.close()
